// comm.cpp — the only collective on the data path: the final gather of result rows
// (and row counts) over RCCL/xGMI.  Site pairs never cross a (footprint, strand)
// block (src/giremi/mismatch.py:387-391), so ranks share nothing while computing.
//
// xGMI is a fully connected point-to-point mesh: every rank sends its rows to the root
// in ONE hop (grouped ncclSend/ncclRecv), no ring is built.  librccl is opened lazily
// with dlopen so that loading liblgmi.so never touches RCCL or the GPU.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>

#include "lgmi_internal.h"

struct lgmi_dresult;
namespace lgmi {
hipStream_t ctx_stream(lgmi_ctx* c);
int ctx_device(lgmi_ctx* c);
void** ctx_comm_slot(lgmi_ctx* c);
int* ctx_rank_slot(lgmi_ctx* c);
int* ctx_world_slot(lgmi_ctx* c);
int set_error(int code, const char* msg);
void dresult_view(const lgmi_dresult* r, DResultView* v);
lgmi_dresult* dresult_new_gathered(lgmi_ctx* c, const DResultView& v);
int pool_alloc(lgmi_ctx* c, void** out, size_t bytes);
void pool_release(lgmi_ctx* c, void* p);
}  // namespace lgmi
using namespace lgmi;

namespace {
struct Rccl {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
} g;

int load_rccl() {
    if (g.h) return LGMI_OK;
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return set_error(LGMI_E_RCCL, dlerror());
#define SYM(field, name)                                                        \
    *(void**)(&g.field) = dlsym(h, name);                                       \
    if (!g.field) { dlclose(h); return set_error(LGMI_E_RCCL, "librccl: missing symbol " name); }
    SYM(GetUniqueId, "ncclGetUniqueId") SYM(CommInitRank, "ncclCommInitRank") SYM(CommDestroy, "ncclCommDestroy")
    SYM(AllGather, "ncclAllGather") SYM(Reduce, "ncclReduce") SYM(Send, "ncclSend") SYM(Recv, "ncclRecv")
    SYM(GroupStart, "ncclGroupStart") SYM(GroupEnd, "ncclGroupEnd") SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
    g.h = h;
    return LGMI_OK;
}

int nccl_fail(ncclResult_t r, const char* what) {
    char buf[256];
    snprintf(buf, sizeof buf, "%s: %s", what, g.GetErrorString ? g.GetErrorString(r) : "rccl error");
    return set_error(LGMI_E_RCCL, buf);
}
#define NCCLCHK(expr)                                          \
    do {                                                       \
        ncclResult_t r_ = (expr);                              \
        if (r_ != ncclSuccess) return nccl_fail(r_, #expr);    \
    } while (0)
#define HIPCHK2(expr)                                                                  \
    do {                                                                               \
        hipError_t e_ = (expr);                                                        \
        if (e_ != hipSuccess) return set_error(LGMI_E_HIP, hipGetErrorString(e_));     \
    } while (0)
}  // namespace

static_assert(sizeof(ncclUniqueId) == LGMI_UNIQUE_ID_BYTES, "ncclUniqueId is 128 bytes");

extern "C" int lgmi_comm_unique_id(void* out128) {
    if (!out128) return set_error(LGMI_E_ARG, "out128 is NULL");
    int rc = load_rccl();
    if (rc) return rc;
    ncclUniqueId id;
    NCCLCHK(g.GetUniqueId(&id));
    memcpy(out128, &id, sizeof id);
    return LGMI_OK;
}

extern "C" int lgmi_comm_init(lgmi_ctx* ctx, const void* id128, int rank, int world) {
    if (!ctx || !id128 || world < 1 || rank < 0 || rank >= world) return set_error(LGMI_E_ARG, "bad comm arguments");
    if (*ctx_comm_slot(ctx)) return set_error(LGMI_E_STATE, "communicator already initialised");
    int rc = load_rccl();
    if (rc) return rc;
    HIPCHK2(hipSetDevice(ctx_device(ctx)));
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    ncclComm_t comm = nullptr;
    NCCLCHK(g.CommInitRank(&comm, world, id, rank));
    *ctx_comm_slot(ctx) = comm;
    *ctx_rank_slot(ctx) = rank;
    *ctx_world_slot(ctx) = world;
    return LGMI_OK;
}

extern "C" void lgmi_comm_destroy(lgmi_ctx* ctx) {
    if (!ctx || !*ctx_comm_slot(ctx)) return;
    (void)hipStreamSynchronize(ctx_stream(ctx));
    if (g.CommDestroy) (void)g.CommDestroy((ncclComm_t)*ctx_comm_slot(ctx));
    *ctx_comm_slot(ctx) = nullptr;
}

extern "C" int lgmi_comm_allgather_u64v(lgmi_ctx* ctx, const uint64_t* mine, uint32_t n, uint64_t* out_world) {
    if (!ctx || !mine || !out_world || n == 0) return set_error(LGMI_E_ARG, "NULL argument");
    ncclComm_t comm = (ncclComm_t)*ctx_comm_slot(ctx);
    if (!comm) return set_error(LGMI_E_STATE, "lgmi_comm_init has not been called");
    const int world = *ctx_world_slot(ctx);
    hipStream_t st = ctx_stream(ctx);
    HIPCHK2(hipSetDevice(ctx_device(ctx)));
    uint64_t* d = nullptr;
    int rc = pool_alloc(ctx, (void**)&d, sizeof(uint64_t) * n * (size_t)(world + 1));
    if (rc) return rc;
    hipError_t e = hipMemcpyAsync(d + (size_t)world * n, mine, 8ull * n, hipMemcpyHostToDevice, st);
    ncclResult_t r = ncclSuccess;
    if (e == hipSuccess) r = g.AllGather(d + (size_t)world * n, d, n, ncclUint64, comm, st);
    if (e == hipSuccess && r == ncclSuccess) e = hipMemcpyAsync(out_world, d, 8ull * n * world, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    pool_release(ctx, d);
    if (r != ncclSuccess) return nccl_fail(r, "ncclAllGather");
    if (e != hipSuccess) return set_error(LGMI_E_HIP, hipGetErrorString(e));
    return LGMI_OK;
}

extern "C" int lgmi_comm_allgather_u64(lgmi_ctx* ctx, uint64_t mine, uint64_t* out_world) {
    return lgmi_comm_allgather_u64v(ctx, &mine, 1, out_world);
}

namespace {
__global__ void k_add_base(uint32_t* __restrict__ a, uint32_t* __restrict__ b, uint64_t n, uint32_t base) {
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) { a[k] += base; b[k] += base; }
}
__global__ void k_mean_from_sums(uint32_t n, const unsigned long long* __restrict__ sum, const uint32_t* __restrict__ cnt,
                                 double* __restrict__ mean) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const uint32_t c = cnt[s];
    mean[s] = c ? ((double)sum[s] / MEAN_SCALE) / (double)c : __longlong_as_double(0x7ff8000000000000ll);
}
// p of a Monte-Carlo row is a function of its exceed count: the gather moves 4 bytes instead of 12 and the root
// recomputes p with the expression the permutation kernels use (bit-equal)
__global__ void k_p_from_exceed(double* __restrict__ p, const uint32_t* __restrict__ exceed, uint64_t n, uint32_t n_shuffles) {
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) p[k] = (1.0 + (double)exceed[k]) / ((double)n_shuffles + 1.0);
}
__global__ void k_fill_nan(double* __restrict__ mean, uint32_t* __restrict__ cnt, unsigned long long* __restrict__ sum, uint64_t n) {
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) { mean[k] = __longlong_as_double(0x7ff8000000000000ll); cnt[k] = 0u; sum[k] = 0ull; }
}
enum { M_ROWS = 0, M_FLAGS, M_BASE, M_SITES, M_EXAMINED, M_GENERAL, M_SHUFFLES, M_N };
}  // namespace

extern "C" int lgmi_dresult_fetch(lgmi_dresult* r, lgmi_result* out);
extern "C" void lgmi_dresult_free(lgmi_dresult* r);

extern "C" int lgmi_comm_gather(lgmi_ctx* ctx, const lgmi_dresult* mine, int root, const lgmi_gather_opts* opts,
                                lgmi_dresult** out, uint64_t* rank_row_begin) {
    if (!ctx || !mine || !out) return set_error(LGMI_E_ARG, "NULL argument");
    *out = nullptr;
    ncclComm_t comm = (ncclComm_t)*ctx_comm_slot(ctx);
    if (!comm) return set_error(LGMI_E_STATE, "lgmi_comm_init has not been called");
    const int world = *ctx_world_slot(ctx), rank = *ctx_rank_slot(ctx);
    if (root < 0 || root >= world) return set_error(LGMI_E_ARG, "root out of range");
    lgmi_gather_opts o = {};
    if (opts) o = *opts;
    if (o.reserved[0] || o.reserved[1] || o.reserved[2] || o.same_batch > 1) return set_error(LGMI_E_ARG, "bad gather options");
    hipStream_t st = ctx_stream(ctx);
    HIPCHK2(hipSetDevice(ctx_device(ctx)));
    DResultView v;
    dresult_view(mine, &v);

    // ---- 1. everybody learns everybody's sizes and flags
    uint64_t my_meta[M_N];
    my_meta[M_ROWS] = v.n_rows;
    my_meta[M_FLAGS] = (v.p ? 1u : 0u) | (v.counts ? 2u : 0u) | (o.same_batch ? 4u : 0u) | (v.p && v.p_from_exceed ? 8u : 0u);
    my_meta[M_SHUFFLES] = v.n_shuffles;
    my_meta[M_BASE] = o.site_base;
    my_meta[M_SITES] = v.n_sites;
    my_meta[M_EXAMINED] = v.info.n_examined;
    my_meta[M_GENERAL] = v.info.n_general_rows;
    std::vector<uint64_t> meta((size_t)world * M_N);
    int rc = lgmi_comm_allgather_u64v(ctx, my_meta, M_N, meta.data());
    if (rc) return rc;
    auto M = [&](int r, int k) { return meta[(size_t)r * M_N + k]; };
    uint64_t total = 0, total_sites = 0, examined = 0, general = 0;
    bool consistent = true;
    for (int r = 0; r < world; ++r) {
        if (M(r, M_FLAGS) != M(0, M_FLAGS) || M(r, M_SHUFFLES) != M(0, M_SHUFFLES)) consistent = false;
        if (o.same_batch && (M(r, M_BASE) != 0 || M(r, M_SITES) != M(0, M_SITES))) consistent = false;
        if (M(r, M_BASE) + M(r, M_SITES) >= 0xFFFFFFF0ull) consistent = false;
        total += M(r, M_ROWS);
        total_sites = std::max<uint64_t>(total_sites, M(r, M_BASE) + M(r, M_SITES));
        examined += M(r, M_EXAMINED);
        general += M(r, M_GENERAL);
    }
    if (rank_row_begin) {
        rank_row_begin[0] = 0;
        for (int r = 0; r < world; ++r) rank_row_begin[r + 1] = rank_row_begin[r] + M(r, M_ROWS);
    }
    // every rank holds the same meta table, so every rank takes this exit together
    if (!consistent)
        return set_error(LGMI_E_ARG, "lgmi_comm_gather: ranks disagree (p / counts / same_batch flags, or same_batch with "
                                     "different site counts or a non-zero site_base)");
    const bool has_p = M(0, M_FLAGS) & 1u, has_counts = M(0, M_FLAGS) & 2u, same_batch = M(0, M_FLAGS) & 4u;
    const bool derive_p = M(0, M_FLAGS) & 8u;          // p travels as its exceed count
    const uint32_t n_shuffles = (uint32_t)M(0, M_SHUFFLES);

    // ---- 2. the root allocates; the outcome is agreed on before anything is posted
    uint32_t *gi = nullptr, *gj = nullptr, *gexc = nullptr, *gcnt = nullptr, *gnp = nullptr;
    double *gmi = nullptr, *gp = nullptr, *gmean = nullptr;
    unsigned long long* gsum = nullptr;
    auto release_all = [&]() {
        pool_release(ctx, gi); pool_release(ctx, gj); pool_release(ctx, gmi); pool_release(ctx, gp); pool_release(ctx, gexc);
        pool_release(ctx, gcnt); pool_release(ctx, gmean); pool_release(ctx, gnp); pool_release(ctx, gsum);
    };
    uint64_t my_status = 0;
    if (rank == root) {
        const size_t tn = (size_t)std::max<uint64_t>(total, 1), tsn = (size_t)std::max<uint64_t>(total_sites, 1);
        int e = pool_alloc(ctx, (void**)&gi, tn * 4);
        if (!e) e = pool_alloc(ctx, (void**)&gj, tn * 4);
        if (!e) e = pool_alloc(ctx, (void**)&gmi, tn * 8);
        if (!e && has_p) e = pool_alloc(ctx, (void**)&gp, tn * 8);
        if (!e && has_p) e = pool_alloc(ctx, (void**)&gexc, tn * 4);
        if (!e && has_counts) e = pool_alloc(ctx, (void**)&gcnt, tn * 36);
        if (!e) e = pool_alloc(ctx, (void**)&gmean, tsn * 8);
        if (!e) e = pool_alloc(ctx, (void**)&gnp, tsn * 4);
        if (!e) e = pool_alloc(ctx, (void**)&gsum, tsn * 8);
        if (e) my_status = 1;
    } else if (same_batch) {
        // ncclReduce only writes recvbuff on the root, but every rank hands RCCL a valid device pointer
        const size_t tsn = (size_t)std::max<uint64_t>(v.n_sites, 1);
        int e = pool_alloc(ctx, (void**)&gnp, tsn * 4);
        if (!e) e = pool_alloc(ctx, (void**)&gsum, tsn * 8);
        if (e) my_status = 1;
    }
    std::vector<uint64_t> status(world);
    rc = lgmi_comm_allgather_u64v(ctx, &my_status, 1, status.data());
    if (rc) { release_all(); return rc; }
    for (int r = 0; r < world; ++r)
        if (status[r]) { release_all(); return set_error(LGMI_E_OOM, "lgmi_comm_gather: a rank could not allocate its buffers"); }

    // different batches: sites no rank reports (gaps between the ranks' site ranges) read NaN / 0 pairs; queued on
    // the stream before the receives that fill the ranks' own ranges
    if (rank == root && !same_batch && total_sites)
        hipLaunchKernelGGL(k_fill_nan, dim3((uint32_t)((total_sites + 255) / 256)), dim3(256), 0, st, gmean, gnp, gsum, total_sites);

    // ---- 3. rows: one hop each over the xGMI mesh, straight into their place on the root
    ncclResult_t nr = ncclSuccess;
#define NC(expr) do { if (nr == ncclSuccess) nr = (expr); } while (0)
    NC(g.GroupStart());
    if (rank != root) {
        const uint64_t n = v.n_rows;
        if (n) {
            NC(g.Send(v.i, n, ncclUint32, root, comm, st));
            NC(g.Send(v.j, n, ncclUint32, root, comm, st));
            NC(g.Send(v.mi, n, ncclFloat64, root, comm, st));
            if (has_p) { if (!derive_p) NC(g.Send(v.p, n, ncclFloat64, root, comm, st)); NC(g.Send(v.exceed, n, ncclUint32, root, comm, st)); }
            if (has_counts) NC(g.Send(v.counts, n * 9, ncclUint32, root, comm, st));
        }
        if (!same_batch && v.n_sites) {
            NC(g.Send(v.mean, v.n_sites, ncclFloat64, root, comm, st));
            NC(g.Send(v.npairs, v.n_sites, ncclUint32, root, comm, st));
        }
    } else {
        uint64_t off = 0;
        for (int r = 0; r < world; ++r) {
            const uint64_t c = M(r, M_ROWS), sb = M(r, M_BASE), sn = M(r, M_SITES);
            if (r != root) {
                if (c) {
                    NC(g.Recv(gi + off, c, ncclUint32, r, comm, st));
                    NC(g.Recv(gj + off, c, ncclUint32, r, comm, st));
                    NC(g.Recv(gmi + off, c, ncclFloat64, r, comm, st));
                    if (has_p) { if (!derive_p) NC(g.Recv(gp + off, c, ncclFloat64, r, comm, st)); NC(g.Recv(gexc + off, c, ncclUint32, r, comm, st)); }
                    if (has_counts) NC(g.Recv(gcnt + 9 * off, c * 9, ncclUint32, r, comm, st));
                }
                if (!same_batch && sn) {
                    NC(g.Recv(gmean + sb, sn, ncclFloat64, r, comm, st));
                    NC(g.Recv(gnp + sb, sn, ncclUint32, r, comm, st));
                }
            }
            off += c;
        }
    }
    {   // the group is always closed, error or not: a rank must never be left inside an open group
        const ncclResult_t ge = g.GroupEnd();
        if (nr == ncclSuccess) nr = ge;
    }
    // same batch: the shards' per-site integer sums and counts add up on the root (exact, any order)
    if (nr == ncclSuccess && same_batch && v.n_sites) {
        NC(g.Reduce(v.sum, gsum, v.n_sites, ncclUint64, ncclSum, root, comm, st));
        NC(g.Reduce(v.npairs, gnp, v.n_sites, ncclUint32, ncclSum, root, comm, st));
    }
#undef NC
    if (nr != ncclSuccess) { (void)hipStreamSynchronize(st); release_all(); return nccl_fail(nr, "lgmi_comm_gather"); }
    if (rank != root) {
        const hipError_t es = hipStreamSynchronize(st);
        release_all();
        if (es != hipSuccess) return set_error(LGMI_E_HIP, hipGetErrorString(es));
        return LGMI_OK;
    }
    // ---- 4. root: its own rows, the site bases, the per-site means
    hipError_t e = hipSuccess;
    uint64_t off = 0;
    for (int r = 0; r < root; ++r) off += M(r, M_ROWS);
    const uint64_t n = v.n_rows;
    auto d2d = [&](void* dst, const void* src, size_t bytes) {
        if (e == hipSuccess && bytes) e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st);
    };
    d2d(gi + off, v.i, n * 4); d2d(gj + off, v.j, n * 4); d2d(gmi + off, v.mi, n * 8);
    if (has_p) { if (!derive_p) d2d(gp + off, v.p, n * 8); d2d(gexc + off, v.exceed, n * 4); }
    if (has_counts) d2d(gcnt + 9 * off, v.counts, n * 36);
    if (has_p && derive_p && total)                        // every row, the root's own included
        hipLaunchKernelGGL(k_p_from_exceed, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, st, gp, gexc, total, n_shuffles);
    if (same_batch && v.n_sites)
        hipLaunchKernelGGL(k_mean_from_sums, dim3((uint32_t)((v.n_sites + 255) / 256)), dim3(256), 0, st,
                           (uint32_t)v.n_sites, gsum, gnp, gmean);
    off = 0;
    for (int r = 0; r < world; ++r) {
        const uint64_t c = M(r, M_ROWS), sb = M(r, M_BASE);
        if (c && sb) hipLaunchKernelGGL(k_add_base, dim3((uint32_t)((c + 255) / 256)), dim3(256), 0, st, gi + off, gj + off, c, (uint32_t)sb);
        off += c;
    }
    if (!same_batch) { d2d(gmean + M(root, M_BASE), v.mean, v.n_sites * 8); d2d(gnp + M(root, M_BASE), v.npairs, v.n_sites * 4); }
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { release_all(); return set_error(LGMI_E_HIP, hipGetErrorString(e)); }
    DResultView gv;
    gv.n_rows = total; gv.n_sites = total_sites;
    gv.i = gi; gv.j = gj; gv.mi = gmi; gv.p = gp; gv.exceed = gexc; gv.counts = gcnt;
    gv.mean = gmean; gv.npairs = gnp; gv.sum = gsum;
    gv.n_shuffles = n_shuffles; gv.p_from_exceed = derive_p;
    gv.info = v.info;                       // stage times stay the root's own
    gv.info.n_rows = total; gv.info.n_examined = examined; gv.info.n_general_rows = general;
    *out = dresult_new_gathered(ctx, gv);
    return LGMI_OK;
}

extern "C" int lgmi_comm_gather_rows(lgmi_ctx* ctx, const lgmi_dresult* mine, int root, lgmi_result* out) {
    if (!out) return set_error(LGMI_E_ARG, "NULL argument");
    memset(out, 0, sizeof *out);
    lgmi_dresult* gathered = nullptr;
    int rc = lgmi_comm_gather(ctx, mine, root, nullptr, &gathered, nullptr);
    if (rc || !gathered) return rc;
    rc = lgmi_dresult_fetch(gathered, out);
    lgmi_dresult_free(gathered);
    return rc;
}
