// comm.cpp — the only collective on the data path: the final gather of result rows
// (and row counts) over RCCL/xGMI.  Site pairs never cross a (footprint, strand)
// block (src/giremi/mismatch.py:387-391), so ranks share nothing while computing.
//
// xGMI is a fully connected point-to-point mesh: every rank sends its rows to the root
// in ONE hop (grouped ncclSend/ncclRecv), no ring is built.  librccl is opened lazily
// with dlopen so that loading liblgmi.so never touches RCCL or the GPU.
#include <dlfcn.h>
#include <rccl/rccl.h>

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
void dresult_rows(const lgmi_dresult* r, uint64_t* n, const uint32_t** i, const uint32_t** j, const double** mi,
                  const double** p);
}  // namespace lgmi
using namespace lgmi;

namespace {
struct Rccl {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
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
    SYM(AllGather, "ncclAllGather") SYM(Send, "ncclSend") SYM(Recv, "ncclRecv")
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

extern "C" int lgmi_comm_allgather_u64(lgmi_ctx* ctx, uint64_t mine, uint64_t* out_world) {
    if (!ctx || !out_world) return set_error(LGMI_E_ARG, "NULL argument");
    ncclComm_t comm = (ncclComm_t)*ctx_comm_slot(ctx);
    if (!comm) return set_error(LGMI_E_STATE, "lgmi_comm_init has not been called");
    const int world = *ctx_world_slot(ctx);
    hipStream_t st = ctx_stream(ctx);
    HIPCHK2(hipSetDevice(ctx_device(ctx)));
    uint64_t* d = nullptr;
    HIPCHK2(hipMalloc((void**)&d, sizeof(uint64_t) * (world + 1)));
    hipError_t e = hipMemcpyAsync(d + world, &mine, 8, hipMemcpyHostToDevice, st);
    ncclResult_t r = ncclSuccess;
    if (e == hipSuccess) r = g.AllGather(d + world, d, 1, ncclUint64, comm, st);
    if (e == hipSuccess && r == ncclSuccess) e = hipMemcpyAsync(out_world, d, 8 * world, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(d);
    if (r != ncclSuccess) return nccl_fail(r, "ncclAllGather");
    if (e != hipSuccess) return set_error(LGMI_E_HIP, hipGetErrorString(e));
    return LGMI_OK;
}

namespace {
struct GatherOwner : lgmi::ResultOwner { std::vector<uint32_t> i, j; std::vector<double> mi, p; };
}

extern "C" int lgmi_comm_gather_rows(lgmi_ctx* ctx, const lgmi_dresult* mine, int root, lgmi_result* out) {
    if (!ctx || !mine || !out) return set_error(LGMI_E_ARG, "NULL argument");
    memset(out, 0, sizeof *out);
    ncclComm_t comm = (ncclComm_t)*ctx_comm_slot(ctx);
    if (!comm) return set_error(LGMI_E_STATE, "lgmi_comm_init has not been called");
    const int world = *ctx_world_slot(ctx), rank = *ctx_rank_slot(ctx);
    if (root < 0 || root >= world) return set_error(LGMI_E_ARG, "root out of range");
    hipStream_t st = ctx_stream(ctx);
    HIPCHK2(hipSetDevice(ctx_device(ctx)));
    uint64_t n = 0;
    const uint32_t *di, *dj;
    const double *dmi, *dp;
    dresult_rows(mine, &n, &di, &dj, &dmi, &dp);
    // every rank must agree on whether p is carried: encode it in the count exchange
    std::vector<uint64_t> counts(world);
    int rc = lgmi_comm_allgather_u64(ctx, (n << 1) | (dp ? 1u : 0u), counts.data());
    if (rc) return rc;
    bool has_p = true;
    uint64_t total = 0;
    for (int r = 0; r < world; ++r) { has_p = has_p && (counts[r] & 1u); counts[r] >>= 1; total += counts[r]; }
    if (rank != root) {
        if (n) {
            NCCLCHK(g.GroupStart());
            NCCLCHK(g.Send(di, n, ncclUint32, root, comm, st));
            NCCLCHK(g.Send(dj, n, ncclUint32, root, comm, st));
            NCCLCHK(g.Send(dmi, n, ncclFloat64, root, comm, st));
            if (has_p) NCCLCHK(g.Send(dp, n, ncclFloat64, root, comm, st));
            NCCLCHK(g.GroupEnd());
        }
        HIPCHK2(hipStreamSynchronize(st));
        return LGMI_OK;
    }
    // root: receive straight into one device buffer per column, rank order
    const size_t tn = (size_t)(total ? total : 1);
    uint32_t *gi = nullptr, *gj = nullptr;
    double *gmi = nullptr, *gp = nullptr;
    struct Free { uint32_t** a; uint32_t** b; double** c; double** d;
                  ~Free() { (void)hipFree(*a); (void)hipFree(*b); (void)hipFree(*c); (void)hipFree(*d); } } fr{&gi, &gj, &gmi, &gp};
    HIPCHK2(hipMalloc((void**)&gi, tn * 4));
    HIPCHK2(hipMalloc((void**)&gj, tn * 4));
    HIPCHK2(hipMalloc((void**)&gmi, tn * 8));
    if (has_p) HIPCHK2(hipMalloc((void**)&gp, tn * 8));
    NCCLCHK(g.GroupStart());
    uint64_t off = 0;
    for (int r = 0; r < world; ++r) {
        const uint64_t c = counts[r];
        if (c && r != root) {
            NCCLCHK(g.Recv(gi + off, c, ncclUint32, r, comm, st));
            NCCLCHK(g.Recv(gj + off, c, ncclUint32, r, comm, st));
            NCCLCHK(g.Recv(gmi + off, c, ncclFloat64, r, comm, st));
            if (has_p) NCCLCHK(g.Recv(gp + off, c, ncclFloat64, r, comm, st));
        }
        off += c;
    }
    NCCLCHK(g.GroupEnd());
    off = 0;
    for (int r = 0; r < root; ++r) off += counts[r];
    if (n) {
        HIPCHK2(hipMemcpyAsync(gi + off, di, n * 4, hipMemcpyDeviceToDevice, st));
        HIPCHK2(hipMemcpyAsync(gj + off, dj, n * 4, hipMemcpyDeviceToDevice, st));
        HIPCHK2(hipMemcpyAsync(gmi + off, dmi, n * 8, hipMemcpyDeviceToDevice, st));
        if (has_p) HIPCHK2(hipMemcpyAsync(gp + off, dp, n * 8, hipMemcpyDeviceToDevice, st));
    }
    GatherOwner* h = new GatherOwner();
    h->i.resize(total); h->j.resize(total); h->mi.resize(total);
    if (has_p) h->p.resize(total);
    hipError_t e = hipSuccess;
    if (total) {
        e = hipMemcpyAsync(h->i.data(), gi, total * 4, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipMemcpyAsync(h->j.data(), gj, total * 4, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipMemcpyAsync(h->mi.data(), gmi, total * 8, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess && has_p) e = hipMemcpyAsync(h->p.data(), gp, total * 8, hipMemcpyDeviceToHost, st);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { delete h; return set_error(LGMI_E_HIP, hipGetErrorString(e)); }
    out->n_rows = total;
    out->row_i = h->i.data(); out->row_j = h->j.data(); out->row_mi = h->mi.data();
    out->row_p = has_p ? h->p.data() : nullptr;
    out->owner_ = static_cast<lgmi::ResultOwner*>(h);   // released by lgmi_result_free()
    return LGMI_OK;
}
