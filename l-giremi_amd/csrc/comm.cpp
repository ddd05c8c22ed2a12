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
#include <cstdlib>
#include <cstring>
#include <vector>

#include "lgmi_internal.h"

struct lgmi_dresult;
namespace lgmi {
hipStream_t ctx_stream(lgmi_ctx* c);
hipStream_t ctx_comm_stream(lgmi_ctx* c);
hipEvent_t ctx_comm_event(lgmi_ctx* c);
int ctx_device(lgmi_ctx* c);
void** ctx_comm_slot(lgmi_ctx* c);
uint64_t* ctx_pinned_words(lgmi_ctx* c, size_t* n_words);
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
    // optional (lgmi_comm_info): a library without them still serves the gather
    ncclResult_t (*GetVersion)(int*) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    bool stand_in = false;
    char path[512] = {0};
} g;

int load_rccl() {
    if (g.h) return LGMI_OK;
    // LGMI_RCCL_LIB: another library with the same ten entry points — the tests' file-based stand-in that lets two
    // ranks share the one GPU of a test box (tests/helpers/fake_rccl.cpp).  A product run must never pick one up by
    // accident (a stale environment variable would turn a "scaling" measurement into file copies): it is refused unless
    // LGMI_ALLOW_RCCL_STANDIN=1 says the caller knows, and lgmi_comm_info() reports stand_in = 1 for it.
    const char* alt = getenv("LGMI_RCCL_LIB");
    if (alt && *alt) {
        const char* ok = getenv("LGMI_ALLOW_RCCL_STANDIN");
        if (!ok || strcmp(ok, "1") != 0)
            return set_error(LGMI_E_RCCL, "LGMI_RCCL_LIB is set (an RCCL stand-in) but LGMI_ALLOW_RCCL_STANDIN=1 is not: refusing to "
                                          "run the gather over anything but librccl");
    }
    void* h = (alt && *alt) ? dlopen(alt, RTLD_NOW | RTLD_GLOBAL) : dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h && !(alt && *alt)) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return set_error(LGMI_E_RCCL, dlerror());
#define SYM(field, name)                                                        \
    *(void**)(&g.field) = dlsym(h, name);                                       \
    if (!g.field) { dlclose(h); return set_error(LGMI_E_RCCL, "librccl: missing symbol " name); }
    SYM(GetUniqueId, "ncclGetUniqueId") SYM(CommInitRank, "ncclCommInitRank") SYM(CommDestroy, "ncclCommDestroy")
    SYM(AllGather, "ncclAllGather") SYM(Reduce, "ncclReduce") SYM(Send, "ncclSend") SYM(Recv, "ncclRecv")
    SYM(GroupStart, "ncclGroupStart") SYM(GroupEnd, "ncclGroupEnd") SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
    *(void**)(&g.GetVersion) = dlsym(h, "ncclGetVersion");
    *(void**)(&g.CommCount) = dlsym(h, "ncclCommCount");
    *(void**)(&g.CommUserRank) = dlsym(h, "ncclCommUserRank");
    g.stand_in = alt && *alt;
    Dl_info di;
    if (dladdr((void*)g.CommInitRank, &di) && di.dli_fname) snprintf(g.path, sizeof g.path, "%s", di.dli_fname);
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
    // every collective of this file runs on the communication stream: nothing may still be queued there (a _begin
    // without its _finish, an error path) when the communicator goes
    (void)hipStreamSynchronize(ctx_comm_stream(ctx));
    (void)hipStreamSynchronize(ctx_stream(ctx));
    if (g.CommDestroy) (void)g.CommDestroy((ncclComm_t)*ctx_comm_slot(ctx));
    *ctx_comm_slot(ctx) = nullptr;
}

// what a SCALE line needs to prove that RCCL saw N ranks: the library that was bound, its version, the communicator's
// own idea of its size and of this rank, and whether it is the tests' stand-in
extern "C" int lgmi_comm_info(lgmi_ctx* ctx, lgmi_comm_info_t* out) {
    if (!ctx || !out) return set_error(LGMI_E_ARG, "NULL argument");
    memset(out, 0, sizeof *out);
    out->nranks = -1; out->rank = -1; out->rccl_version = -1;
    if (!g.h) return set_error(LGMI_E_STATE, "no RCCL library is loaded (lgmi_comm_unique_id / lgmi_comm_init come first)");
    out->stand_in = g.stand_in ? 1 : 0;
    snprintf(out->lib_path, sizeof out->lib_path, "%s", g.path);
    int v = -1;
    if (g.GetVersion && g.GetVersion(&v) == ncclSuccess) out->rccl_version = v;
    ncclComm_t comm = (ncclComm_t)*ctx_comm_slot(ctx);
    out->initialised = comm ? 1 : 0;
    if (comm) {
        int n = -1, r = -1;
        if (g.CommCount && g.CommCount(comm, &n) == ncclSuccess) out->nranks = n;
        if (g.CommUserRank && g.CommUserRank(comm, &r) == ncclSuccess) out->rank = r;
        out->world_given = *ctx_world_slot(ctx);
        out->rank_given = *ctx_rank_slot(ctx);
    }
    return LGMI_OK;
}

namespace {
// order the communication stream after everything queued so far on the main stream (the kernels that made the rows a
// gather is about to send): an event, not a host-side wait in somebody else's function
int comm_after_main(lgmi_ctx* ctx) {
    hipStream_t ms = ctx_stream(ctx), cs = ctx_comm_stream(ctx);
    if (ms == cs) return LGMI_OK;
    hipEvent_t ev = ctx_comm_event(ctx);
    if (!ev) return set_error(LGMI_E_HIP, "no event for the communication stream");
    HIPCHK2(hipEventRecord(ev, ms));
    HIPCHK2(hipStreamWaitEvent(cs, ev, 0));
    return LGMI_OK;
}
}  // namespace

extern "C" int lgmi_comm_allgather_u64v(lgmi_ctx* ctx, const uint64_t* mine, uint32_t n, uint64_t* out_world) {
    if (!ctx || !mine || !out_world || n == 0) return set_error(LGMI_E_ARG, "NULL argument");
    ncclComm_t comm = (ncclComm_t)*ctx_comm_slot(ctx);
    if (!comm) return set_error(LGMI_E_STATE, "lgmi_comm_init has not been called");
    const int world = *ctx_world_slot(ctx);
    HIPCHK2(hipSetDevice(ctx_device(ctx)));
    hipStream_t st = ctx_comm_stream(ctx);
    uint64_t* d = nullptr;
    int rc = pool_alloc(ctx, (void**)&d, sizeof(uint64_t) * n * (size_t)(world + 1));
    if (rc) return rc;
    // through the context's pinned words when they fit (they do for everything the gather exchanges): copies to and
    // from pageable memory are staged by the runtime and were seen to block the host for tens of milliseconds
    size_t pw = 0;
    uint64_t* pin = ctx_pinned_words(ctx, &pw);
    const bool pinned = (size_t)n * (size_t)(world + 1) <= pw;
    const uint64_t* src = mine;
    uint64_t* dst = out_world;
    if (pinned) { memcpy(pin, mine, 8ull * n); src = pin; dst = pin + n; }
    hipError_t e = hipMemcpyAsync(d + (size_t)world * n, src, 8ull * n, hipMemcpyHostToDevice, st);
    ncclResult_t r = ncclSuccess;
    if (e == hipSuccess) r = g.AllGather(d + (size_t)world * n, d, n, ncclUint64, comm, st);
    if (e == hipSuccess && r == ncclSuccess) e = hipMemcpyAsync(dst, d, 8ull * n * world, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e == hipSuccess && r == ncclSuccess && pinned) memcpy(out_world, dst, 8ull * n * world);
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
// ---- the compact gather (shards of ONE batch): row_i never travels, row_j only from a rank that skipped a pair, the
// permutation counts in 16 bits when they fit.  The root rebuilds row_i / row_j of rank r's rows from the rank's per-site
// row counts: rows are in reference order (mutual_information.py:10-12), a site's rows are consecutive, and a rank that
// emitted every pair it examined holds, for each site, a run of consecutive candidates starting at c0 — 0, or where the
// rank's first work item starts inside its first site's row.
// one wave per site
__global__ __launch_bounds__(256) void k_expand_rows(uint32_t n_sites, const uint32_t* __restrict__ nfirst, const uint64_t* __restrict__ row_off,
                                                     uint64_t base, int write_j, uint32_t first_site, uint32_t first_c0,
                                                     const uint8_t* __restrict__ isx, const uint32_t* __restrict__ cand0,
                                                     const uint32_t* __restrict__ xs, uint32_t* __restrict__ gi, uint32_t* __restrict__ gj)
{
    const uint32_t s = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (s >= n_sites) return;
    const uint32_t n = nfirst[s];
    if (!n) return;
    const uint64_t o = base + row_off[s];
    const uint32_t c0 = cand0[s] + (s == first_site ? first_c0 : 0u);
    const bool x = isx[s] != 0;
    for (uint32_t k = lane; k < n; k += 64u) {
        gi[o + k] = s;
        if (write_j) gj[o + k] = x ? c0 + k : xs[c0 + k];
    }
}
__global__ void k_widen_u16(uint64_t n, const uint16_t* __restrict__ in, uint32_t* __restrict__ out) {
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) out[k] = in[k];
}
enum { M_ROWS = 0, M_FLAGS, M_BASE, M_SITES, M_EXAMINED, M_GENERAL, M_SHUFFLES, M_ALL, M_FSITE, M_FSEG, M_N };
}  // namespace

extern "C" int lgmi_dresult_fetch(lgmi_dresult* r, lgmi_result* out);
extern "C" void lgmi_dresult_free(lgmi_dresult* r);

// the state of a gather between its two halves
struct lgmi_gather {
    lgmi_ctx* ctx = nullptr;
    const lgmi_dresult* mine = nullptr;
    int root = 0, world = 1, rank = 0;
    std::vector<uint64_t> meta;
    uint64_t total = 0, total_sites = 0, examined = 0;
    bool has_p = false, has_counts = false, same_batch = false, derive_p = false, no_p_array = false;
    uint32_t n_shuffles = 0;
    uint32_t *gi = nullptr, *gj = nullptr, *gexc = nullptr, *gcnt = nullptr, *gnp = nullptr;
    double *gmi = nullptr, *gp = nullptr, *gmean = nullptr;
    unsigned long long* gsum = nullptr;
    // compact form
    bool compact = false, narrow = false;
    uint32_t* gnfirst_all = nullptr;       // root: [world][n_sites] the ranks' rows by first site
    uint32_t* gnfirst = nullptr;           // root: their sum (the gathered result's own)
    uint32_t* gncand = nullptr;
    uint16_t* g16 = nullptr;               // root: the ranks' 16-bit counts as they arrive; elsewhere: this rank's narrowed counts
    uint64_t M(int r, int k) const { return meta[(size_t)r * M_N + k]; }
    void release_tmp() { pool_release(ctx, gnfirst_all); pool_release(ctx, g16); gnfirst_all = nullptr; g16 = nullptr; }
    void release_all() {
        pool_release(ctx, gi); pool_release(ctx, gj); pool_release(ctx, gmi); pool_release(ctx, gp); pool_release(ctx, gexc);
        pool_release(ctx, gcnt); pool_release(ctx, gmean); pool_release(ctx, gnp); pool_release(ctx, gsum);
        pool_release(ctx, gnfirst); pool_release(ctx, gncand);
        gi = gj = gexc = gcnt = gnp = nullptr; gmi = gp = gmean = nullptr; gsum = nullptr; gnfirst = gncand = nullptr;
        release_tmp();
    }
};

extern "C" int lgmi_comm_gather_begin(lgmi_ctx* ctx, const lgmi_dresult* mine, int root, const lgmi_gather_opts* opts,
                                      lgmi_gather** handle) {
    if (!ctx || !mine || !handle) return set_error(LGMI_E_ARG, "NULL argument");
    *handle = nullptr;
    ncclComm_t comm = (ncclComm_t)*ctx_comm_slot(ctx);
    if (!comm) return set_error(LGMI_E_STATE, "lgmi_comm_init has not been called");
    const int world = *ctx_world_slot(ctx), rank = *ctx_rank_slot(ctx);
    if (root < 0 || root >= world) return set_error(LGMI_E_ARG, "root out of range");
    lgmi_gather_opts o = {};
    if (opts) o = *opts;
    if (o.reserved[0] || o.reserved[1] || o.reserved[2] || o.same_batch > 1) return set_error(LGMI_E_ARG, "bad gather options");
    HIPCHK2(hipSetDevice(ctx_device(ctx)));
    hipStream_t st = ctx_comm_stream(ctx);
    {   // the rows of `mine` were made on the main stream: the communication stream waits for them on the device
        const int rc0 = comm_after_main(ctx);
        if (rc0) return rc0;
    }
    DResultView v;
    dresult_view(mine, &v);

    // ---- 1. everybody learns everybody's sizes and flags
    uint64_t my_meta[M_N];
    my_meta[M_ROWS] = v.n_rows;
    my_meta[M_FLAGS] = (v.exceed ? 1u : 0u) | (v.counts ? 2u : 0u) | (o.same_batch ? 4u : 0u) | (v.exceed && v.p_from_exceed ? 8u : 0u) |
                       (v.exceed && !v.p ? 16u : 0u);          // 16: no p array at all (lgmi_params.no_row_p)
    my_meta[M_BASE] = o.site_base;
    my_meta[M_SITES] = v.n_sites;
    my_meta[M_EXAMINED] = v.info.n_examined;
    my_meta[M_GENERAL] = 0;
    my_meta[M_SHUFFLES] = v.n_shuffles;
    // compact gather: this rank can take part (bit 32 of the flags: every rank must), and emitted every pair it examined
    static const bool legacy = getenv("LGMI_GATHER_LEGACY") != nullptr;
    if (o.same_batch && !legacy && v.nfirst && v.ncand && v.site_type && v.block_site_begin) my_meta[M_FLAGS] |= 1ull << 32;
    my_meta[M_ALL] = v.n_rows == v.info.n_examined ? 1u : 0u;
    my_meta[M_FSITE] = v.first_site; my_meta[M_FSEG] = v.first_seg;
    lgmi_gather* h = new lgmi_gather();
    struct Drop { lgmi_gather* p; ~Drop() { if (p) { p->release_all(); delete p; } } } drop{h};
    h->ctx = ctx; h->mine = mine; h->root = root; h->world = world; h->rank = rank;
    h->meta.resize((size_t)world * M_N);
    int rc = lgmi_comm_allgather_u64v(ctx, my_meta, M_N, h->meta.data());
    if (rc) return rc;
    bool consistent = true;
    for (int r = 0; r < world; ++r) {
        if (h->M(r, M_FLAGS) != h->M(0, M_FLAGS) || h->M(r, M_SHUFFLES) != h->M(0, M_SHUFFLES)) consistent = false;
        if (o.same_batch && (h->M(r, M_BASE) != 0 || h->M(r, M_SITES) != h->M(0, M_SITES))) consistent = false;
        if (h->M(r, M_BASE) + h->M(r, M_SITES) >= 0xFFFFFFF0ull) consistent = false;
        h->total += h->M(r, M_ROWS);
        h->total_sites = std::max<uint64_t>(h->total_sites, h->M(r, M_BASE) + h->M(r, M_SITES));
        h->examined += h->M(r, M_EXAMINED);
    }
    // every rank holds the same meta table, so every rank takes this exit together
    if (!consistent)
        return set_error(LGMI_E_ARG, "lgmi_comm_gather: ranks disagree (p / counts / same_batch flags, shuffles, or same_batch "
                                     "with different site counts or a non-zero site_base)");
    h->has_p = h->M(0, M_FLAGS) & 1u; h->has_counts = h->M(0, M_FLAGS) & 2u; h->same_batch = h->M(0, M_FLAGS) & 4u;
    h->derive_p = h->M(0, M_FLAGS) & 8u;               // p travels as its exceed count
    h->no_p_array = h->M(0, M_FLAGS) & 16u;            // and is not materialised on the root either
    h->n_shuffles = (uint32_t)h->M(0, M_SHUFFLES);
    h->compact = h->same_batch && (h->M(0, M_FLAGS) >> 32 & 1u);       // (the flags are equal on all ranks: checked above)
    h->narrow = h->compact && h->has_p && h->derive_p && h->n_shuffles <= 65535u;
    const uint64_t total = h->total, total_sites = h->total_sites;

    // ---- 2. the root allocates; the outcome is agreed on before anything is posted
    uint64_t my_status = 0;
    if (rank == root) {
        const size_t tn = (size_t)std::max<uint64_t>(total, 1), tsn = (size_t)std::max<uint64_t>(total_sites, 1);
        int e = pool_alloc(ctx, (void**)&h->gi, tn * 4);
        if (!e) e = pool_alloc(ctx, (void**)&h->gj, tn * 4);
        if (!e) e = pool_alloc(ctx, (void**)&h->gmi, tn * 8);
        if (!e && h->has_p && !h->no_p_array) e = pool_alloc(ctx, (void**)&h->gp, tn * 8);
        if (!e && h->has_p) e = pool_alloc(ctx, (void**)&h->gexc, tn * 4);
        if (!e && h->has_counts) e = pool_alloc(ctx, (void**)&h->gcnt, tn * 36);
        if (!e) e = pool_alloc(ctx, (void**)&h->gmean, tsn * 8);
        if (!e) e = pool_alloc(ctx, (void**)&h->gnp, tsn * 4);
        if (!e) e = pool_alloc(ctx, (void**)&h->gsum, tsn * 8);
        if (!e && h->compact) e = pool_alloc(ctx, (void**)&h->gnfirst_all, tsn * 4 * (size_t)world);
        if (!e && h->compact) e = pool_alloc(ctx, (void**)&h->gnfirst, tsn * 4);
        if (!e && h->compact) e = pool_alloc(ctx, (void**)&h->gncand, tsn * 4);
        if (!e && h->narrow) e = pool_alloc(ctx, (void**)&h->g16, tn * 2 + 4);
        if (e) my_status = 1;
    } else if (h->same_batch) {
        // ncclReduce only writes recvbuff on the root, but every rank hands RCCL a valid device pointer
        const size_t tsn = (size_t)std::max<uint64_t>(v.n_sites, 1);
        int e = pool_alloc(ctx, (void**)&h->gnp, tsn * 4);
        if (!e) e = pool_alloc(ctx, (void**)&h->gsum, tsn * 8);
        if (!e && h->narrow) e = pool_alloc(ctx, (void**)&h->g16, (size_t)std::max<uint64_t>(v.n_rows, 1) * 2 + 4);
        if (e) my_status = 1;
    }
    std::vector<uint64_t> status(world);
    rc = lgmi_comm_allgather_u64v(ctx, &my_status, 1, status.data());
    if (rc) return rc;
    for (int r = 0; r < world; ++r)
        if (status[r]) return set_error(LGMI_E_OOM, "lgmi_comm_gather: a rank could not allocate its buffers");

    // different batches: sites no rank reports (gaps between the ranks' site ranges) read NaN / 0 pairs; queued on
    // the stream before the receives that fill the ranks' own ranges
    if (rank == root && !h->same_batch && total_sites)
        hipLaunchKernelGGL(k_fill_nan, dim3((uint32_t)((total_sites + 255) / 256)), dim3(256), 0, st, h->gmean, h->gnp, h->gsum, total_sites);

    // ---- 3. what is final already — (i, j, mi) and the tables — one hop each over the xGMI mesh, straight into place on
    //         the root.  Not waited for: the caller may run the permutation stage on its main stream meanwhile.
    ncclResult_t nr = ncclSuccess;
#define NC(expr) do { if (nr == ncclSuccess) nr = (expr); } while (0)
    NC(g.GroupStart());
    if (rank != root) {
        const uint64_t n = v.n_rows;
        if (n) {
            // compact: row_i is a run-length code of the per-site counts below; row_j only if this rank skipped a pair
            if (!h->compact) NC(g.Send(v.i, n, ncclUint32, root, comm, st));
            if (!h->compact || !h->M(rank, M_ALL)) NC(g.Send(v.j, n, ncclUint32, root, comm, st));
            NC(g.Send(v.mi, n, ncclFloat64, root, comm, st));
            if (h->has_counts) NC(g.Send(v.counts, n * 9, ncclUint32, root, comm, st));
        }
        if (h->compact && v.n_sites) NC(g.Send(v.nfirst, v.n_sites, ncclUint32, root, comm, st));
    } else {
        uint64_t off = 0;
        for (int r = 0; r < world; ++r) {
            const uint64_t c = h->M(r, M_ROWS);
            if (r != root && c) {
                if (!h->compact) NC(g.Recv(h->gi + off, c, ncclUint32, r, comm, st));
                if (!h->compact || !h->M(r, M_ALL)) NC(g.Recv(h->gj + off, c, ncclUint32, r, comm, st));
                NC(g.Recv(h->gmi + off, c, ncclFloat64, r, comm, st));
                if (h->has_counts) NC(g.Recv(h->gcnt + 9 * off, c * 9, ncclUint32, r, comm, st));
            }
            if (r != root && h->compact && v.n_sites) NC(g.Recv(h->gnfirst_all + (size_t)r * v.n_sites, v.n_sites, ncclUint32, r, comm, st));
            off += c;
        }
    }
    {   // the group is always closed, error or not: a rank must never be left inside an open group
        const ncclResult_t ge = g.GroupEnd();
        if (nr == ncclSuccess) nr = ge;
    }
    if (nr != ncclSuccess) { (void)hipStreamSynchronize(st); return nccl_fail(nr, "lgmi_comm_gather_begin"); }
    if (rank == root) {             // the root's own rows
        hipError_t e = hipSuccess;
        uint64_t off = 0;
        for (int r = 0; r < root; ++r) off += h->M(r, M_ROWS);
        const uint64_t n = v.n_rows;
        auto d2d = [&](void* dst, const void* src, size_t bytes) {
            if (e == hipSuccess && bytes) e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st);
        };
        d2d(h->gi + off, v.i, n * 4); d2d(h->gj + off, v.j, n * 4); d2d(h->gmi + off, v.mi, n * 8);
        if (h->has_counts) d2d(h->gcnt + 9 * off, v.counts, n * 36);
        if (h->compact) { d2d(h->gnfirst_all + (size_t)root * v.n_sites, v.nfirst, v.n_sites * 4); d2d(h->gncand, v.ncand, v.n_sites * 4); }
        if (e != hipSuccess) { (void)hipStreamSynchronize(st); return set_error(LGMI_E_HIP, hipGetErrorString(e)); }
    }
#undef NC
    drop.p = nullptr;
    *handle = h;
    return LGMI_OK;
}

extern "C" int lgmi_comm_gather_finish(lgmi_gather* h, lgmi_dresult** out, uint64_t* rank_row_begin) {
    if (!h || !out) { if (h) { h->release_all(); delete h; } return set_error(LGMI_E_ARG, "NULL argument"); }
    *out = nullptr;
    struct Drop { lgmi_gather* p; bool keep; ~Drop() { if (!keep) p->release_all(); delete p; } } drop{h, false};
    lgmi_ctx* ctx = h->ctx;
    ncclComm_t comm = (ncclComm_t)*ctx_comm_slot(ctx);
    if (!comm) return set_error(LGMI_E_STATE, "the communicator is gone");
    const int world = h->world, rank = h->rank, root = h->root;
    if (hipSetDevice(ctx_device(ctx)) != hipSuccess) return set_error(LGMI_E_HIP, "hipSetDevice");
    hipStream_t st = ctx_comm_stream(ctx);
    {   // the permutation stage, if it ran in between, ran on the main stream: wait for it on the device
        const int rc0 = comm_after_main(ctx);
        if (rc0) return rc0;
    }
    DResultView v;
    dresult_view(h->mine, &v);
    if (rank_row_begin) {
        rank_row_begin[0] = 0;
        for (int r = 0; r < world; ++r) rank_row_begin[r + 1] = rank_row_begin[r] + h->M(r, M_ROWS);
    }
    // the larger-than-2x2 row counts are only known now
    uint64_t my_general = v.info.n_general_rows;
    std::vector<uint64_t> gen(world);
    int rc = lgmi_comm_allgather_u64v(ctx, &my_general, 1, gen.data());
    if (rc) { (void)hipStreamSynchronize(st); return rc; }
    uint64_t general = 0;
    for (int r = 0; r < world; ++r) general += gen[r];

    // ---- 4. what the permutation stage made, then the per-site figures
    // (every count is at most n_shuffles <= 65535: 2 bytes instead of 4 — narrowed here, outside the group of sends)
    if (rank != root && h->narrow && v.n_rows && h->has_p) {
        launch_narrow_u16(st, v.n_rows, v.exceed, h->g16);
        if (hipGetLastError() != hipSuccess) { (void)hipStreamSynchronize(st); return set_error(LGMI_E_HIP, "narrowing kernel"); }
    }
    ncclResult_t nr = ncclSuccess;
#define NC(expr) do { if (nr == ncclSuccess) nr = (expr); } while (0)
    NC(g.GroupStart());
    if (rank != root) {
        const uint64_t n = v.n_rows;
        if (n && h->has_p) {
            if (!h->derive_p) NC(g.Send(v.p, n, ncclFloat64, root, comm, st));
            if (h->narrow) NC(g.Send(h->g16, 2 * n, ncclUint8, root, comm, st));    // (narrowed before the group was opened)
            else NC(g.Send(v.exceed, n, ncclUint32, root, comm, st));
        }
        if (!h->same_batch && v.n_sites) {
            NC(g.Send(v.mean, v.n_sites, ncclFloat64, root, comm, st));
            NC(g.Send(v.npairs, v.n_sites, ncclUint32, root, comm, st));
        }
    } else {
        uint64_t off = 0;
        for (int r = 0; r < world; ++r) {
            const uint64_t c = h->M(r, M_ROWS), sb = h->M(r, M_BASE), sn = h->M(r, M_SITES);
            if (r != root) {
                if (c && h->has_p) {
                    if (!h->derive_p) NC(g.Recv(h->gp + off, c, ncclFloat64, r, comm, st));
                    if (h->narrow) NC(g.Recv(h->g16 + off, 2 * c, ncclUint8, r, comm, st));
                    else NC(g.Recv(h->gexc + off, c, ncclUint32, r, comm, st));
                }
                if (!h->same_batch && sn) {
                    NC(g.Recv(h->gmean + sb, sn, ncclFloat64, r, comm, st));
                    NC(g.Recv(h->gnp + sb, sn, ncclUint32, r, comm, st));
                }
            }
            off += c;
        }
    }
    {
        const ncclResult_t ge = g.GroupEnd();
        if (nr == ncclSuccess) nr = ge;
    }
    // same batch: the shards' per-site integer sums and counts add up on the root (exact, any order)
    if (nr == ncclSuccess && h->same_batch && v.n_sites) {
        NC(g.Reduce(v.sum, h->gsum, v.n_sites, ncclUint64, ncclSum, root, comm, st));
        NC(g.Reduce(v.npairs, h->gnp, v.n_sites, ncclUint32, ncclSum, root, comm, st));
    }
#undef NC
    if (nr != ncclSuccess) { (void)hipStreamSynchronize(st); return nccl_fail(nr, "lgmi_comm_gather_finish"); }
    if (rank != root) {
        const hipError_t es = hipStreamSynchronize(st);
        if (es != hipSuccess) return set_error(LGMI_E_HIP, hipGetErrorString(es));
        return LGMI_OK;
    }
    if (h->compact && h->total) {
        // ---- 4b. root, compact form: row_i (and the row_j of the ranks that emitted every pair) from the ranks' per-site row
        //          counts; the 16-bit counts widened into place.  Host tables of the batch's candidates: per site its first
        //          candidate (an x site: the next site; another site: its place in the block's x-site list), the x sites by rank
        const uint32_t ns = (uint32_t)v.n_sites;
        const std::vector<uint64_t>& bsb = *v.block_site_begin;
        const std::vector<uint8_t>& typ = *v.site_type;
        std::vector<uint8_t> isx(ns);
        std::vector<uint32_t> cand0(ns), xs;
        xs.reserve(ns);
        for (size_t b = 0; b + 1 < bsb.size(); ++b) {
            const uint32_t x0 = (uint32_t)xs.size();
            for (uint64_t s = bsb[b]; s < bsb[b + 1]; ++s) if (!v.het_only || typ[s] == LGMI_TYPE_HET_SNP) xs.push_back((uint32_t)s);
            uint32_t xnext = 0;
            for (uint64_t s = bsb[b]; s < bsb[b + 1]; ++s) {
                isx[s] = (!v.het_only || typ[s] == LGMI_TYPE_HET_SNP) ? 1 : 0;
                if (isx[s]) { ++xnext; cand0[s] = (uint32_t)s + 1u; } else cand0[s] = x0 + xnext;
            }
        }
        uint8_t* d_isx = nullptr; uint32_t* d_cand0 = nullptr; uint32_t* d_xs = nullptr; uint64_t* d_off = nullptr; uint64_t* d_tmp = nullptr;
        struct Tmp { lgmi_ctx* c; hipStream_t st; void** p[5]; ~Tmp() { (void)hipStreamSynchronize(st); for (auto q : p) pool_release(c, *q); } }
            tmp{ctx, st, {(void**)&d_isx, (void**)&d_cand0, (void**)&d_xs, (void**)&d_off, (void**)&d_tmp}};
        int e2 = pool_alloc(ctx, (void**)&d_isx, std::max<size_t>(ns, 1));
        if (!e2) e2 = pool_alloc(ctx, (void**)&d_cand0, std::max<size_t>(ns, 1) * 4);
        if (!e2) e2 = pool_alloc(ctx, (void**)&d_xs, std::max<size_t>(xs.size(), 1) * 4);
        if (!e2) e2 = pool_alloc(ctx, (void**)&d_off, ((size_t)ns + 1) * 8);
        if (!e2) e2 = pool_alloc(ctx, (void**)&d_tmp, scan_tmp_words(ns) * 8);
        if (e2) return e2;
        HIPCHK2(hipMemcpyAsync(d_isx, isx.data(), ns, hipMemcpyHostToDevice, st));
        HIPCHK2(hipMemcpyAsync(d_cand0, cand0.data(), (size_t)ns * 4, hipMemcpyHostToDevice, st));
        if (!xs.empty()) HIPCHK2(hipMemcpyAsync(d_xs, xs.data(), xs.size() * 4, hipMemcpyHostToDevice, st));
        HIPCHK2(hipMemsetAsync(h->gnfirst, 0, (size_t)ns * 4, st));
        uint64_t roff = 0;
        for (int r = 0; r < world; ++r) {
            const uint64_t c = h->M(r, M_ROWS);
            const uint32_t* nf = h->gnfirst_all + (size_t)r * ns;
            if (c) {
                launch_scan(st, nf, d_off, ns, d_tmp);
                // the rank's first work item may start inside its first site's row: segment g of an x site starts at partner
                // g * LGMI_EMIT_SEG, of another site at g * LGMI_EMIT_SEG_Q
                const uint32_t fs = (uint32_t)h->M(r, M_FSITE);
                const uint32_t c0 = (uint32_t)h->M(r, M_FSEG) * ((fs < ns && isx[fs]) ? LGMI_EMIT_SEG : LGMI_EMIT_SEG_Q);
                hipLaunchKernelGGL(k_expand_rows, dim3((ns + 3u) / 4u), dim3(256), 0, st, ns, nf, d_off, roff, h->M(r, M_ALL) ? 1 : 0,
                                   fs, c0, d_isx, d_cand0, d_xs, h->gi, h->gj);
                if (h->narrow && r != root)
                    hipLaunchKernelGGL(k_widen_u16, dim3((uint32_t)((c + 255) / 256)), dim3(256), 0, st, c, h->g16 + roff, h->gexc + roff);
            }
            launch_add_u32(st, ns, h->gnfirst, nf);
            roff += c;
        }
        HIPCHK2(hipGetLastError());
        HIPCHK2(hipStreamSynchronize(st));            // (the host vectors above go out of scope)
    }
    // ---- 5. root: its own p / exceed, the site bases, the per-site means
    hipError_t e = hipSuccess;
    uint64_t off = 0;
    for (int r = 0; r < root; ++r) off += h->M(r, M_ROWS);
    const uint64_t n = v.n_rows, total = h->total;
    auto d2d = [&](void* dst, const void* src, size_t bytes) {
        if (e == hipSuccess && bytes) e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st);
    };
    if (h->has_p) { if (!h->derive_p) d2d(h->gp + off, v.p, n * 8); d2d(h->gexc + off, v.exceed, n * 4); }
    if (h->has_p && h->derive_p && !h->no_p_array && total)   // every row, the root's own included
        hipLaunchKernelGGL(k_p_from_exceed, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, st, h->gp, h->gexc, total, h->n_shuffles);
    if (h->same_batch && v.n_sites)
        hipLaunchKernelGGL(k_mean_from_sums, dim3((uint32_t)((v.n_sites + 255) / 256)), dim3(256), 0, st,
                           (uint32_t)v.n_sites, h->gsum, h->gnp, h->gmean);
    off = 0;
    for (int r = 0; r < world; ++r) {
        const uint64_t c = h->M(r, M_ROWS), sb = h->M(r, M_BASE);
        if (c && sb) hipLaunchKernelGGL(k_add_base, dim3((uint32_t)((c + 255) / 256)), dim3(256), 0, st, h->gi + off, h->gj + off, c, (uint32_t)sb);
        off += c;
    }
    if (!h->same_batch) { d2d(h->gmean + h->M(root, M_BASE), v.mean, v.n_sites * 8); d2d(h->gnp + h->M(root, M_BASE), v.npairs, v.n_sites * 4); }
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return set_error(LGMI_E_HIP, hipGetErrorString(e));
    DResultView gv;
    gv.n_rows = total; gv.n_sites = h->total_sites;
    gv.i = h->gi; gv.j = h->gj; gv.mi = h->gmi; gv.p = h->gp; gv.exceed = h->gexc; gv.counts = h->gcnt;
    gv.mean = h->gmean; gv.npairs = h->gnp; gv.sum = h->gsum;
    gv.n_shuffles = h->n_shuffles; gv.p_from_exceed = h->derive_p;
    if (h->compact) {                       // the gathered result keeps its compact form within reach
        gv.nfirst = h->gnfirst; gv.ncand = h->gncand; gv.table_owner = v.table_owner; gv.het_only = v.het_only;
        gv.first_site = (uint32_t)h->M(0, M_FSITE); gv.first_seg = (uint32_t)h->M(0, M_FSEG);
    }
    h->release_tmp();
    gv.info = v.info;                       // stage times stay the root's own
    gv.info.n_rows = total; gv.info.n_examined = h->examined; gv.info.n_general_rows = general;
    *out = dresult_new_gathered(ctx, gv);
    drop.keep = true;                       // the buffers belong to the gathered result now
    return LGMI_OK;
}

extern "C" int lgmi_comm_gather(lgmi_ctx* ctx, const lgmi_dresult* mine, int root, const lgmi_gather_opts* opts,
                                lgmi_dresult** out, uint64_t* rank_row_begin) {
    if (!out) return set_error(LGMI_E_ARG, "NULL argument");
    *out = nullptr;
    lgmi_gather* h = nullptr;
    const int rc = lgmi_comm_gather_begin(ctx, mine, root, opts, &h);
    if (rc) return rc;
    return lgmi_comm_gather_finish(h, out, rank_row_begin);
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
