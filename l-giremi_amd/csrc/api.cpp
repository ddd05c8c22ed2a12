// api.cpp — the C ABI of liblgmi.so (include/lgmi.h): contexts, batch upload,
// per-run planning on the host, kernel sequencing on one HIP stream, results.
//
// Reference interface replaced: the two Python functions of
// src/giremi/mutual_information.py (:6-45, :48-60) as called from
// src/giremi/mismatch.py:384-404 — see include/lgmi.h for the mapping.
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "lgmi_internal.h"
#include "philox.h"

using namespace lgmi;

// ---------------------------------------------------------------- errors
static thread_local std::string g_err;

static int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIPCHK(expr)                                                                        \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess)                                                               \
            return fail(e_ == hipErrorOutOfMemory ? LGMI_E_OOM : LGMI_E_HIP, "%s: %s (%s:%d)", #expr, \
                        hipGetErrorString(e_), __FILE__, __LINE__);                         \
    } while (0)

// ---------------------------------------------------------------- device memory pool
// Grow-only cache of device allocations so that repeated runs (bench steps) do not
// pay hipMalloc/hipFree inside the timed region.
struct Pool {
    std::multimap<size_t, void*> free_;
    std::map<void*, size_t> live_;
    int alloc(void** out, size_t bytes) {
        size_t n = std::max<size_t>(256, (bytes + 255) & ~size_t(255));
        auto it = free_.lower_bound(n);
        if (it != free_.end() && it->first <= 2 * n + (1u << 20)) {
            *out = it->second;
            live_[it->second] = it->first;
            free_.erase(it);
            return LGMI_OK;
        }
        void* p = nullptr;
        hipError_t e = hipMalloc(&p, n);
        if (e != hipSuccess) {  // give cached blocks back and retry once
            trim();
            e = hipMalloc(&p, n);
            if (e != hipSuccess) return fail(LGMI_E_OOM, "hipMalloc(%zu bytes): %s", n, hipGetErrorString(e));
        }
        live_[p] = n;
        *out = p;
        return LGMI_OK;
    }
    void release(void* p) {
        if (!p) return;
        auto it = live_.find(p);
        if (it == live_.end()) return;
        free_.emplace(it->second, p);
        live_.erase(it);
    }
    void trim() {
        for (auto& kv : free_) (void)hipFree(kv.second);
        free_.clear();
    }
    void destroy() {
        trim();
        for (auto& kv : live_) (void)hipFree(kv.first);
        live_.clear();
    }
};

// ---------------------------------------------------------------- objects
struct lgmi_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev[8] = {};
    Pool pool;
    long long* d_G = nullptr;   // round(n ln n * 2^28): permutation statistic (perm.hip)
    double* d_LF = nullptr;     // ln n!
    uint32_t tables_len = 0;
    void* comm = nullptr;       // ncclComm_t (comm.cpp)
    int rank = 0, world = 1;
};

struct lgmi_dbatch {
    lgmi_ctx* ctx = nullptr;
    DevBatch d;
    // host copies of the site metadata (planning happens on the host)
    std::vector<uint64_t> block_site_begin;
    std::vector<uint32_t> block_n_reads;
    std::vector<int64_t> pos;
    std::vector<uint8_t> type, tri;
    std::vector<Col> cols;               // [n_cols]
    std::vector<uint32_t> pseudo_site;   // site of each pseudo column
    std::vector<uint32_t> pseudo_of_site;// column id of the site's pseudo column or NONE
    uint32_t max_reads = 0;
    // download buffers
    std::vector<uint32_t> dl_word_off, dl_n_words;
    std::vector<uint64_t> dl_plane_off, dl_planes;
};

struct lgmi_dresult {
    lgmi_ctx* ctx = nullptr;
    lgmi_run_info info = {};
    uint64_t n_rows = 0, n_sites = 0;
    uint32_t* d_i = nullptr; uint32_t* d_j = nullptr;
    double* d_mi = nullptr; double* d_p = nullptr;
    uint32_t* d_exceed = nullptr; uint32_t* d_counts = nullptr;
    double* d_mean = nullptr; uint32_t* d_npairs = nullptr;
    bool has_p = false, has_counts = false;
};

struct HostResult : ResultOwner {  // owner_ of a host lgmi_result
    std::vector<uint32_t> i, j, exceed, counts, npairs;
    std::vector<double> mi, p, mean;
};

// ---------------------------------------------------------------- basics
extern "C" int lgmi_abi_version(void) { return LGMI_ABI_VERSION; }
extern "C" const char* lgmi_last_error(void) { return g_err.c_str(); }

extern "C" int lgmi_device_count(int* out_count) {
    if (!out_count) return fail(LGMI_E_ARG, "out_count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *out_count = 0; return fail(LGMI_E_NODEV, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    *out_count = n;
    return LGMI_OK;
}

extern "C" int lgmi_ctx_create(int device_id, lgmi_ctx** out) {
    if (!out) return fail(LGMI_E_ARG, "out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(LGMI_E_NODEV, "no HIP device available (%s); liblgmi has no CPU fallback",
                    e != hipSuccess ? hipGetErrorString(e) : "device count 0");
    if (device_id < 0 || device_id >= n) return fail(LGMI_E_ARG, "device_id %d out of range [0,%d)", device_id, n);
    HIPCHK(hipSetDevice(device_id));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device_id));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(LGMI_E_NODEV, "device %d is %s; liblgmi is built for gfx950 (MI355X) only", device_id, prop.gcnArchName);
    lgmi_ctx* c = new lgmi_ctx();
    c->device = device_id;
    HIPCHK(hipStreamCreate(&c->stream));
    for (auto& ev : c->ev) HIPCHK(hipEventCreate(&ev));
    *out = c;
    return LGMI_OK;
}

extern "C" void lgmi_comm_destroy(lgmi_ctx* ctx);

extern "C" void lgmi_ctx_destroy(lgmi_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    lgmi_comm_destroy(ctx);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->d_G) (void)hipFree(ctx->d_G);
    if (ctx->d_LF) (void)hipFree(ctx->d_LF);
    ctx->pool.destroy();
    for (auto& ev : ctx->ev) if (ev) (void)hipEventDestroy(ev);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

// accessors for comm.cpp (keeps lgmi_ctx private to this file)
namespace lgmi {
hipStream_t ctx_stream(lgmi_ctx* c) { return c->stream; }
int ctx_device(lgmi_ctx* c) { return c->device; }
void** ctx_comm_slot(lgmi_ctx* c) { return &c->comm; }
int* ctx_rank_slot(lgmi_ctx* c) { return &c->rank; }
int* ctx_world_slot(lgmi_ctx* c) { return &c->world; }
int set_error(int code, const char* msg) { return fail(code, "%s", msg); }
void dresult_rows(const lgmi_dresult* r, uint64_t* n, const uint32_t** i, const uint32_t** j, const double** mi,
                  const double** p) {
    *n = r->n_rows; *i = r->d_i; *j = r->d_j; *mi = r->d_mi; *p = r->has_p ? r->d_p : nullptr;
}
}  // namespace lgmi

// ---------------------------------------------------------------- upload
static void free_dbatch_device(lgmi_dbatch* db) {
    if (!db) return;
    (void)hipFree(db->d.d_pos); (void)hipFree(db->d.d_type); (void)hipFree(db->d.d_tri);
    (void)hipFree(db->d.d_cols); (void)hipFree(db->d.d_cplanes);
    db->d = DevBatch();
}

extern "C" void lgmi_dbatch_free(lgmi_dbatch* db) {
    if (!db) return;
    (void)hipSetDevice(db->ctx->device);
    (void)hipStreamSynchronize(db->ctx->stream);
    free_dbatch_device(db);
    delete db;
}

static int validate_batch(const lgmi_batch* b) {
    if (!b) return fail(LGMI_E_ARG, "batch is NULL");
    if (b->n_blocks && (!b->block_site_begin || !b->block_n_reads)) return fail(LGMI_E_ARG, "block arrays are NULL");
    if (!b->block_site_begin && b->n_sites) return fail(LGMI_E_ARG, "block_site_begin is NULL");
    if (b->n_sites >= 0xFFFFFFF0ull) return fail(LGMI_E_ARG, "too many sites (%llu)", (unsigned long long)b->n_sites);
    if (b->n_sites && (!b->site_pos || !b->site_type || !b->site_word_off || !b->site_n_words || !b->site_plane_off))
        return fail(LGMI_E_ARG, "site arrays are NULL");
    if (b->n_plane_words && !b->planes) return fail(LGMI_E_ARG, "planes is NULL");
    if (b->n_blocks == 0) {
        if (b->n_sites) return fail(LGMI_E_ARG, "n_sites %llu with 0 blocks", (unsigned long long)b->n_sites);
        return LGMI_OK;
    }
    if (b->block_site_begin[0] != 0 || b->block_site_begin[b->n_blocks] != b->n_sites)
        return fail(LGMI_E_ARG, "block_site_begin must start at 0 and end at n_sites");
    for (uint64_t k = 0; k < b->n_blocks; ++k) {
        uint64_t sb = b->block_site_begin[k], se = b->block_site_begin[k + 1];
        if (se < sb) return fail(LGMI_E_ARG, "block_site_begin not monotone at block %llu", (unsigned long long)k);
        uint64_t W = ((uint64_t)b->block_n_reads[k] + 63) / 64;
        for (uint64_t s = sb; s < se; ++s) {
            if (b->site_type[s] > 2) return fail(LGMI_E_ARG, "site %llu: type %u", (unsigned long long)s, b->site_type[s]);
            if ((uint64_t)b->site_word_off[s] + b->site_n_words[s] > W)
                return fail(LGMI_E_ARG, "site %llu: band [%u,+%u) exceeds %llu words of block %llu",
                            (unsigned long long)s, b->site_word_off[s], b->site_n_words[s], (unsigned long long)W,
                            (unsigned long long)k);
            if (b->site_plane_off[s] + 2ull * b->site_n_words[s] > b->n_plane_words)
                return fail(LGMI_E_ARG, "site %llu: planes exceed n_plane_words", (unsigned long long)s);
            if (s > sb && b->site_pos[s] <= b->site_pos[s - 1])
                return fail(LGMI_E_ARG, "site %llu: positions must increase strictly inside a block", (unsigned long long)s);
        }
    }
    return LGMI_OK;
}

template <class T>
static int dev_copy_new(T** dptr, const T* h, size_t n, hipStream_t st) {
    *dptr = nullptr;
    if (!n) return LGMI_OK;
    HIPCHK(hipMalloc((void**)dptr, n * sizeof(T)));
    HIPCHK(hipMemcpyAsync(*dptr, h, n * sizeof(T), hipMemcpyHostToDevice, st));
    return LGMI_OK;
}

extern "C" int lgmi_batch_upload(lgmi_ctx* ctx, const lgmi_batch* b, lgmi_dbatch** out) {
    if (!ctx || !out) return fail(LGMI_E_ARG, "ctx/out is NULL");
    *out = nullptr;
    int rc = validate_batch(b);
    if (rc) return rc;
    HIPCHK(hipSetDevice(ctx->device));
    lgmi_dbatch* db = new lgmi_dbatch();
    db->ctx = ctx;
    struct Guard { lgmi_dbatch* p; ~Guard() { if (p) { free_dbatch_device(p); delete p; } } } guard{db};
    const uint64_t ns = b->n_sites;
    db->d.n_blocks = b->n_blocks;
    db->d.n_sites = ns;
    db->block_site_begin.assign(b->block_site_begin, b->block_site_begin + b->n_blocks + (b->block_site_begin ? 1 : 0));
    if (db->block_site_begin.empty()) db->block_site_begin.push_back(0);
    db->block_n_reads.assign(b->block_n_reads, b->block_n_reads + b->n_blocks);
    db->pos.assign(b->site_pos, b->site_pos + ns);
    db->type.assign(b->site_type, b->site_type + ns);
    for (uint64_t k = 0; k < b->n_blocks; ++k) db->max_reads = std::max(db->max_reads, b->block_n_reads[k]);

    hipStream_t st = ctx->stream;
    uint64_t* d_planes = nullptr; uint64_t* d_poff = nullptr; uint32_t* d_nw = nullptr; uint32_t* d_pseudo = nullptr;
    struct Tmp { uint64_t** a; uint64_t** b; uint32_t** c; uint32_t** d;
                 ~Tmp() { (void)hipFree(*a); (void)hipFree(*b); (void)hipFree(*c); (void)hipFree(*d); } } tmp{&d_planes, &d_poff, &d_nw, &d_pseudo};
    if ((rc = dev_copy_new(&d_planes, b->planes, b->n_plane_words, st))) return rc;
    if ((rc = dev_copy_new(&d_poff, b->site_plane_off, ns, st))) return rc;
    if ((rc = dev_copy_new(&d_nw, b->site_n_words, ns, st))) return rc;
    if ((rc = dev_copy_new(&db->d.d_pos, b->site_pos, ns, st))) return rc;
    if ((rc = dev_copy_new(&db->d.d_type, b->site_type, ns, st))) return rc;
    db->tri.assign(ns, 0);
    if (ns) {
        HIPCHK(hipMalloc((void**)&db->d.d_tri, ns));
        HIPCHK(hipMemsetAsync(db->d.d_tri, 0, ns, st));
        launch_tri_flags(st, (uint32_t)ns, d_nw, d_poff, d_planes, db->d.d_tri);
        HIPCHK(hipMemcpyAsync(db->tri.data(), db->d.d_tri, ns, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
    }
    // column table: real sites, then one pseudo column per tri site
    db->pseudo_of_site.assign(ns, NONE);
    uint64_t off = 0;
    db->cols.resize(ns);
    for (uint64_t s = 0; s < ns; ++s) {
        db->cols[s] = Col{off, b->site_word_off[s], b->site_n_words[s]};
        off += b->site_n_words[s];
    }
    for (uint64_t s = 0; s < ns; ++s) {
        if (db->tri[s]) {
            db->pseudo_of_site[s] = (uint32_t)(ns + db->pseudo_site.size());
            db->pseudo_site.push_back((uint32_t)s);
            db->cols.push_back(Col{off, b->site_word_off[s], b->site_n_words[s]});
            off += b->site_n_words[s];
        }
    }
    if (db->cols.size() >= 0xFFFFFFF0ull) return fail(LGMI_E_ARG, "too many columns");
    db->d.n_cols = db->cols.size();
    db->d.n_pairs16 = off;
    if ((rc = dev_copy_new(&db->d.d_cols, db->cols.data(), db->cols.size(), st))) return rc;
    if ((rc = dev_copy_new(&d_pseudo, db->pseudo_site.data(), db->pseudo_site.size(), st))) return rc;
    // one all-zero entry after the last column: k_count_mfma reads it for words outside a column's band
    HIPCHK(hipMalloc((void**)&db->d.d_cplanes, (off + 1) * sizeof(ulonglong2)));
    HIPCHK(hipMemsetAsync(db->d.d_cplanes + off, 0, sizeof(ulonglong2), st));
    launch_prep_cols(st, (uint32_t)db->d.n_cols, (uint32_t)ns, db->d.d_cols, d_pseudo, d_poff, d_planes, db->d.d_cplanes);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));
    guard.p = nullptr;
    *out = db;
    return LGMI_OK;
}

// ---------------------------------------------------------------- synthetic dense chromosome
extern "C" int lgmi_synth_dense(lgmi_ctx* ctx, const lgmi_synth_spec* sp, lgmi_dbatch** out) {
    if (!ctx || !sp || !out) return fail(LGMI_E_ARG, "NULL argument");
    *out = nullptr;
    if (sp->n_sites == 0 || sp->n_reads == 0) return fail(LGMI_E_ARG, "n_sites and n_reads must be > 0");
    if (sp->dropout_u16 > 65536 || sp->het_noise_u16 > 65536 || sp->tri_frac_u16 > 65536)
        return fail(LGMI_E_ARG, "u16 probabilities must be <= 65536");
    HIPCHK(hipSetDevice(ctx->device));
    lgmi_dbatch* db = new lgmi_dbatch();
    db->ctx = ctx;
    struct Guard { lgmi_dbatch* p; ~Guard() { if (p) { free_dbatch_device(p); delete p; } } } guard{db};
    const uint32_t ns = sp->n_sites, W = (sp->n_reads + 63) / 64;
    db->d.n_blocks = 1;
    db->d.n_sites = ns;
    db->block_site_begin = {0, ns};
    db->block_n_reads = {sp->n_reads};
    db->max_reads = sp->n_reads;
    db->pos.resize(ns); db->type.resize(ns); db->tri.resize(ns);
    db->pseudo_of_site.assign(ns, NONE);
    for (uint32_t s = 0; s < ns; ++s) {
        SynthSite ss = synth_site(*sp, s);
        db->pos[s] = 10000 + 37ll * s;
        db->type[s] = ss.het ? LGMI_TYPE_HET_SNP : (ss.snp ? LGMI_TYPE_SNP : LGMI_TYPE_MISMATCH);
        db->tri[s] = ss.tri ? 1 : 0;
    }
    db->cols.resize(ns);
    for (uint32_t s = 0; s < ns; ++s) db->cols[s] = Col{(uint64_t)s * W, 0u, W};
    for (uint32_t s = 0; s < ns; ++s) {
        if (db->tri[s]) {
            uint32_t c = (uint32_t)db->cols.size();
            db->pseudo_of_site[s] = c;
            db->pseudo_site.push_back(s);
            db->cols.push_back(Col{(uint64_t)c * W, 0u, W});
        }
    }
    db->d.n_cols = db->cols.size();
    db->d.n_pairs16 = (uint64_t)db->cols.size() * W;
    hipStream_t st = ctx->stream;
    int rc;
    if ((rc = dev_copy_new(&db->d.d_pos, db->pos.data(), ns, st))) return rc;
    if ((rc = dev_copy_new(&db->d.d_type, db->type.data(), ns, st))) return rc;
    if ((rc = dev_copy_new(&db->d.d_tri, db->tri.data(), ns, st))) return rc;
    if ((rc = dev_copy_new(&db->d.d_cols, db->cols.data(), db->cols.size(), st))) return rc;
    HIPCHK(hipMalloc((void**)&db->d.d_cplanes, (db->d.n_pairs16 + 1) * sizeof(ulonglong2)));
    HIPCHK(hipMemsetAsync(db->d.d_cplanes + db->d.n_pairs16, 0, sizeof(ulonglong2), st));
    uint32_t* d_depth = nullptr; uint32_t* d_pos_ = nullptr;
    struct Tmp { uint32_t** a; uint32_t** b; ~Tmp() { (void)hipFree(*a); (void)hipFree(*b); } } tmp{&d_depth, &d_pos_};
    HIPCHK(hipMalloc((void**)&d_depth, (size_t)ns * 3 * sizeof(uint32_t)));
    HIPCHK(hipMemsetAsync(d_depth, 0, (size_t)ns * 3 * sizeof(uint32_t), st));
    if ((rc = dev_copy_new(&d_pos_, db->pseudo_of_site.data(), ns, st))) return rc;
    launch_synth_depth(st, *sp, W, d_depth);
    launch_synth_write(st, *sp, W, d_depth, d_pos_, db->d.d_cplanes);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));
    guard.p = nullptr;
    *out = db;
    return LGMI_OK;
}

// ---------------------------------------------------------------- download (HBM -> ABI lo/hi planes)
extern "C" int lgmi_dbatch_download(lgmi_dbatch* db, lgmi_batch* out) {
    if (!db || !out) return fail(LGMI_E_ARG, "NULL argument");
    HIPCHK(hipSetDevice(db->ctx->device));
    const uint64_t ns = db->d.n_sites;
    std::vector<ulonglong2> cp(db->d.n_pairs16);
    if (!cp.empty()) HIPCHK(hipMemcpy(cp.data(), db->d.d_cplanes, cp.size() * sizeof(ulonglong2), hipMemcpyDeviceToHost));
    db->dl_word_off.resize(ns); db->dl_n_words.resize(ns); db->dl_plane_off.resize(ns);
    uint64_t total = 0;
    for (uint64_t s = 0; s < ns; ++s) {
        db->dl_word_off[s] = db->cols[s].w0;
        db->dl_n_words[s] = db->cols[s].nw;
        db->dl_plane_off[s] = total;
        total += 2ull * db->cols[s].nw;
    }
    db->dl_planes.assign(total, 0);
    for (uint64_t s = 0; s < ns; ++s) {
        const Col& c = db->cols[s];
        uint64_t* lo = db->dl_planes.data() + db->dl_plane_off[s];
        uint64_t* hi = lo + c.nw;
        const uint32_t pc = db->pseudo_of_site[s];
        for (uint32_t k = 0; k < c.nw; ++k) {
            const uint64_t C = cp[c.off + k].x, M2 = cp[c.off + k].y;
            const uint64_t M1 = (pc != NONE) ? cp[db->cols[pc].off + k].y : (C & ~M2);
            const uint64_t c0 = C & ~M2 & ~M1;
            lo[k] = M1 | c0;
            hi[k] = M2 | c0;
        }
    }
    out->n_blocks = db->d.n_blocks;
    out->n_sites = ns;
    out->n_plane_words = total;
    out->block_site_begin = db->block_site_begin.data();
    out->block_n_reads = db->block_n_reads.data();
    out->site_pos = db->pos.data();
    out->site_type = db->type.data();
    out->site_word_off = db->dl_word_off.data();
    out->site_n_words = db->dl_n_words.data();
    out->site_plane_off = db->dl_plane_off.data();
    out->planes = db->dl_planes.data();
    return LGMI_OK;
}

// ---------------------------------------------------------------- planning (host, every run)
struct Plan {
    std::vector<BlockPlan> plans;
    std::vector<uint32_t> xlist, ylist;
    std::vector<SiteMap> smap;
    bool mfma_fp4 = true;           // every matrix-core block has fewer than 2^24 reads: f32 accumulation is exact
    std::vector<Tile> tiles;        // 64 x 64 tiles for k_count (VALU popcount)
    std::vector<Tile> mtiles;       // 128 x 128 tiles for the matrix-core count kernels
    std::vector<uint2> items;       // emit work items: (site, segment of EMIT_SEG partners), in row order
    uint64_t total_slots = 0, n_examined = 0, bytes_in = 0;
};

static void build_plan(const lgmi_dbatch* db, bool het_only, Plan& pl) {
    int count_kernel_choice = 0;   // 0 auto, 1 VALU popcount only, 2 matrix cores only, 3 matrix cores with int8 operands only
    if (const char* e = getenv("LGMI_COUNT_KERNEL")) {
        if (!strcmp(e, "valu")) count_kernel_choice = 1;
        else if (!strcmp(e, "mfma")) count_kernel_choice = 2;
        else if (!strcmp(e, "mfma_i8")) count_kernel_choice = 3;
    }
    if (count_kernel_choice == 3) pl.mfma_fp4 = false;
    const uint64_t ns = db->d.n_sites;
    pl.smap.assign(ns, SiteMap{NONE, NONE, NONE, NONE, 0, 0});
    pl.plans.resize(db->d.n_blocks);
    std::vector<uint32_t> xmin, xmax, ymin, ymax;
    for (uint64_t b = 0; b < db->d.n_blocks; ++b) {
        const uint32_t sb = (uint32_t)db->block_site_begin[b], se = (uint32_t)db->block_site_begin[b + 1];
        const uint32_t P = se - sb;
        BlockPlan bp{};
        bp.slot_base = pl.total_slots;
        bp.xl_off = (uint32_t)pl.xlist.size();
        bp.yl_off = (uint32_t)pl.ylist.size();
        bp.site_begin = sb;
        bp.site_end = se;
        // x list: x sites in position order, then pseudo rows of the tri x sites
        uint32_t nxs = 0;
        for (uint32_t s = sb; s < se; ++s) {
            const bool in_x = !het_only || db->type[s] == LGMI_TYPE_HET_SNP;
            if (in_x) { pl.smap[s].xrow = nxs++; pl.xlist.push_back(s); }
            pl.smap[s].xnext = nxs;
            pl.smap[s].block = (uint32_t)b;
        }
        uint32_t nx = nxs;
        for (uint32_t s = sb; s < se; ++s)
            if (pl.smap[s].xrow != NONE && db->tri[s]) { pl.smap[s].prow = nx++; pl.xlist.push_back(db->pseudo_of_site[s]); }
        // y list: non-x sites, x sites (same order as the x list), pseudo cols of every tri site
        uint32_t ny = 0;
        for (uint32_t s = sb; s < se; ++s) if (pl.smap[s].xrow == NONE) { pl.smap[s].ycol = ny++; pl.ylist.push_back(s); }
        const uint32_t y_xpart = ny;
        for (uint32_t s = sb; s < se; ++s) if (pl.smap[s].xrow != NONE) { pl.smap[s].ycol = ny++; pl.ylist.push_back(s); }
        for (uint32_t s = sb; s < se; ++s) if (db->tri[s]) { pl.smap[s].pcol = ny++; pl.ylist.push_back(db->pseudo_of_site[s]); }
        bp.nx = nx; bp.ny = ny; bp.nxs = nxs;
        bp.ny_pad = (ny + 3u) & ~3u;
        if (nxs == 0 || P < 2) { bp.nx = 0; }
        pl.total_slots += (uint64_t)bp.nx * bp.ny_pad;
        pl.plans[b] = bp;
        // examined pairs: pairs a wave of the emit kernel will look at
        for (uint32_t s = sb; s < se; ++s) {
            const uint32_t ncand = (pl.smap[s].xrow != NONE) ? (se - 1 - s) : (nxs - pl.smap[s].xnext);
            pl.n_examined += ncand;
            for (uint32_t g = 0; g * EMIT_SEG < ncand; ++g) pl.items.push_back(make_uint2(s, g));
        }
        if (bp.nx == 0) continue;
        // which count kernel: the matrix-core kernel pays off on blocks with many columns and many reads
        // (its 128 x 128 tile has a 256-store epilogue per lane); small or shallow blocks keep the
        // VALU popcount kernel.  LGMI_COUNT_KERNEL=valu|mfma forces one of them (tests, A/B runs).
        const uint32_t block_words = (db->block_n_reads[b] + 63u) / 64u;
        bool use_mfma = bp.nx >= 96 && bp.ny >= 96 && block_words >= 32;
        if (count_kernel_choice == 1) use_mfma = false;
        if (count_kernel_choice >= 2) use_mfma = true;
        if (db->block_n_reads[b] >= (1u << 26)) use_mfma = false;   // the int8 kernel's accumulators hold 64 * count in 32 bits
        if (use_mfma && db->block_n_reads[b] >= (1u << 24)) pl.mfma_fp4 = false;
        const uint32_t edge = use_mfma ? 128u : (uint32_t)TILE;
        std::vector<Tile>& out_tiles = use_mfma ? pl.mtiles : pl.tiles;
        // tiles: union band per `edge`-column group of each list
        const uint32_t ntx = (bp.nx + edge - 1) / edge, nty = (bp.ny + edge - 1) / edge;
        xmin.assign(ntx, 0xFFFFFFFFu); xmax.assign(ntx, 0); ymin.assign(nty, 0xFFFFFFFFu); ymax.assign(nty, 0);
        for (uint32_t r = 0; r < bp.nx; ++r) {
            const Col& c = db->cols[pl.xlist[bp.xl_off + r]];
            if (!c.nw) continue;
            xmin[r / edge] = std::min(xmin[r / edge], c.w0); xmax[r / edge] = std::max(xmax[r / edge], c.w0 + c.nw);
        }
        for (uint32_t q = 0; q < bp.ny; ++q) {
            const Col& c = db->cols[pl.ylist[bp.yl_off + q]];
            if (!c.nw) continue;
            ymin[q / edge] = std::min(ymin[q / edge], c.w0); ymax[q / edge] = std::max(ymax[q / edge], c.w0 + c.nw);
        }
        // tile order: groups of XG x-tile rows sweep the y tiles together, so that the XG tiles that
        // run side by side on an XCD (xcd_remap in the kernels) share one y tile in L2 and every y column is
        // fetched from HBM once per group instead of once per x-tile row
        uint32_t XG = use_mfma ? 4 : 8;
        if (const char* e = getenv("LGMI_XG")) XG = (uint32_t)std::max(1, atoi(e));
        for (uint32_t tg = 0; tg < ntx; tg += XG) {
            for (uint32_t ty = 0; ty < nty; ++ty) {
                if (ymin[ty] >= ymax[ty]) continue;
                const uint32_t y0 = ty * edge, y1 = std::min(y0 + edge, bp.ny);
                for (uint32_t tx = tg; tx < std::min(tg + XG, ntx); ++tx) {
                    if (xmin[tx] >= xmax[tx]) continue;
                    const uint32_t x0 = tx * edge, x1 = std::min(x0 + edge, bp.nx);
                    // x site rows against x site cols: only row rank < col rank is ever read
                    if (x1 <= nxs && y0 >= y_xpart && y1 <= y_xpart + nxs && x0 >= (y1 - 1 - y_xpart)) continue;
                    const uint32_t k0 = std::max(xmin[tx], ymin[ty]), k1 = std::min(xmax[tx], ymax[ty]);
                    if (k0 >= k1) continue;
                    out_tiles.push_back(Tile{(uint32_t)b, x0, y0, k0, k1});
                }
            }
        }
    }
    for (uint64_t s = 0; s < ns; ++s) pl.bytes_in += 16ull * db->cols[s].nw + 17ull;
}

// ---------------------------------------------------------------- the run
// host-built tables of the permutation test: libm is only ever called
// here, on the host, so the device results do not depend on the device math library
static int ensure_perm_tables(lgmi_ctx* ctx, uint32_t max_n) {
    if (ctx->tables_len > max_n) return LGMI_OK;
    if (ctx->d_G) { (void)hipFree(ctx->d_G); ctx->d_G = nullptr; }
    if (ctx->d_LF) { (void)hipFree(ctx->d_LF); ctx->d_LF = nullptr; }
    ctx->tables_len = 0;
    std::vector<long long> g((size_t)max_n + 1);
    std::vector<double> lf((size_t)max_n + 1);
    g[0] = 0;
    for (uint32_t n = 1; n <= max_n; ++n) g[n] = llrint((double)n * std::log((double)n) * 268435456.0);
    for (uint32_t n = 0; n <= max_n; ++n) lf[n] = lgamma((double)n + 1.0);
    HIPCHK(hipMalloc((void**)&ctx->d_G, g.size() * sizeof(long long)));
    HIPCHK(hipMalloc((void**)&ctx->d_LF, lf.size() * sizeof(double)));
    HIPCHK(hipMemcpy(ctx->d_G, g.data(), g.size() * sizeof(long long), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(ctx->d_LF, lf.data(), lf.size() * sizeof(double), hipMemcpyHostToDevice));
    ctx->tables_len = max_n + 1;
    return LGMI_OK;
}

extern "C" void lgmi_dresult_free(lgmi_dresult* r) {
    if (!r) return;
    Pool& p = r->ctx->pool;
    p.release(r->d_i); p.release(r->d_j); p.release(r->d_mi); p.release(r->d_p);
    p.release(r->d_exceed); p.release(r->d_counts); p.release(r->d_mean); p.release(r->d_npairs);
    delete r;
}

extern "C" int lgmi_run_device(lgmi_ctx* ctx, const lgmi_dbatch* db, const lgmi_params* prm, lgmi_dresult** out) {
    if (!ctx || !db || !prm || !out) return fail(LGMI_E_ARG, "NULL argument");
    *out = nullptr;
    if (db->ctx != ctx) return fail(LGMI_E_ARG, "batch belongs to another context");
    for (uint8_t r : prm->reserved) if (r) return fail(LGMI_E_ARG, "reserved params bytes must be 0");
    if (prm->n_shuffles > (1u << 24)) return fail(LGMI_E_ARG, "n_shuffles must be <= 2^24");
    HIPCHK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    Pool& pool = ctx->pool;
    const uint32_t ns = (uint32_t)db->d.n_sites;
    const bool want_p = prm->n_shuffles > 0;
    const bool want_counts = prm->emit_counts != 0;
    int rc;

    // the log-factorial table also serves the binomial draw of the 2 x 2 path: LF[0 .. n_shuffles]
    if (want_p && (rc = ensure_perm_tables(ctx, std::max(db->max_reads, prm->n_shuffles)))) return rc;   // first use only
    HIPCHK(hipEventRecord(ctx->ev[0], st));
    Plan pl;
    build_plan(db, prm->het_only != 0, pl);
    if (pl.tiles.size() >= 0x7FFFFFFFull || pl.mtiles.size() >= 0x7FFFFFFFull || pl.items.size() >= 0x7FFFFFFFull)
        return fail(LGMI_E_ARG, "too many tiles");

    lgmi_dresult* res = new lgmi_dresult();
    res->ctx = ctx;
    res->n_sites = ns;
    res->has_p = want_p;
    res->has_counts = want_counts;
    // scratch (returned to the pool at the end of the call) and the result
    std::vector<void*> scratch;
    struct Guard {
        Pool& p; std::vector<void*>& s; lgmi_dresult* r;
        ~Guard() { for (void* q : s) p.release(q); if (r) lgmi_dresult_free(r); }
    } guard{pool, scratch, res};
    auto salloc = [&](void** p, size_t bytes) { int e = pool.alloc(p, bytes); if (!e) scratch.push_back(*p); return e; };

    BlockPlan* d_plans; uint32_t* d_xlist; uint32_t* d_ylist; SiteMap* d_smap; Tile* d_tiles; Tile* d_mtiles; uint2* d_items;
    uint32_t *sN, *sR, *sC, *sA, *d_rowcnt; uint64_t* d_rowstart; unsigned long long* d_sum; uint32_t* d_cnt;
    int* d_err; unsigned long long* d_wordpairs;
    if ((rc = salloc((void**)&d_plans, pl.plans.size() * sizeof(BlockPlan)))) return rc;
    if ((rc = salloc((void**)&d_xlist, pl.xlist.size() * 4))) return rc;
    if ((rc = salloc((void**)&d_items, std::max<size_t>(pl.items.size(), 1) * sizeof(uint2)))) return rc;
    if ((rc = salloc((void**)&d_ylist, pl.ylist.size() * 4))) return rc;
    if ((rc = salloc((void**)&d_smap, pl.smap.size() * sizeof(SiteMap)))) return rc;
    if ((rc = salloc((void**)&d_tiles, pl.tiles.size() * sizeof(Tile)))) return rc;
    if ((rc = salloc((void**)&d_mtiles, pl.mtiles.size() * sizeof(Tile)))) return rc;
    if ((rc = salloc((void**)&sN, pl.total_slots * 4))) return rc;
    if ((rc = salloc((void**)&sR, pl.total_slots * 4))) return rc;
    if ((rc = salloc((void**)&sC, pl.total_slots * 4))) return rc;
    if ((rc = salloc((void**)&sA, pl.total_slots * 4))) return rc;
    const size_t n_items = pl.items.size();
    if ((rc = salloc((void**)&d_rowcnt, std::max<size_t>(n_items, 1) * 4))) return rc;
    if ((rc = salloc((void**)&d_rowstart, (n_items + 1) * 8))) return rc;
    if ((rc = salloc((void**)&d_sum, (size_t)ns * 8))) return rc;
    if ((rc = salloc((void**)&d_cnt, (size_t)ns * 4 + 16))) return rc;
    d_err = (int*)(d_cnt + ns);
    d_wordpairs = nullptr;
    if ((rc = salloc((void**)&d_wordpairs, 8))) return rc;
    if ((rc = pool.alloc((void**)&res->d_mean, (size_t)ns * 8))) return rc;
    if ((rc = pool.alloc((void**)&res->d_npairs, (size_t)ns * 4))) return rc;

    auto h2d = [&](void* d, const void* h, size_t n) -> hipError_t {
        return n ? hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, st) : hipSuccess;
    };
    HIPCHK(h2d(d_plans, pl.plans.data(), pl.plans.size() * sizeof(BlockPlan)));
    HIPCHK(h2d(d_xlist, pl.xlist.data(), pl.xlist.size() * 4));
    HIPCHK(h2d(d_items, pl.items.data(), pl.items.size() * sizeof(uint2)));
    HIPCHK(h2d(d_ylist, pl.ylist.data(), pl.ylist.size() * 4));
    HIPCHK(h2d(d_smap, pl.smap.data(), pl.smap.size() * sizeof(SiteMap)));
    HIPCHK(h2d(d_tiles, pl.tiles.data(), pl.tiles.size() * sizeof(Tile)));
    HIPCHK(h2d(d_mtiles, pl.mtiles.data(), pl.mtiles.size() * sizeof(Tile)));
    HIPCHK(hipMemsetAsync(d_sum, 0, (size_t)ns * 8, st));
    HIPCHK(hipMemsetAsync(d_cnt, 0, (size_t)ns * 4 + 16, st));
    HIPCHK(hipMemsetAsync(d_wordpairs, 0, 8, st));

    HIPCHK(hipEventRecord(ctx->ev[1], st));
    (pl.mfma_fp4 ? launch_count_mfma_fp4 : launch_count_mfma)(
        st, (uint32_t)pl.mtiles.size(), d_mtiles, d_plans, d_xlist, d_ylist, db->d.d_cols, db->d.d_cplanes,
        db->d.d_cplanes + db->d.n_pairs16, sN, sR, sC, sA);
    launch_count(st, (uint32_t)pl.tiles.size(), d_tiles, d_plans, d_xlist, d_ylist, db->d.d_cols, db->d.d_cplanes,
                 sN, sR, sC, sA);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(ctx->ev[2], st));

    EmitArgs ea{};
    ea.n_sites = ns; ea.min_common = prm->min_common; ea.het_only = prm->het_only != 0;
    ea.plans = d_plans; ea.smap = d_smap; ea.xlist = d_xlist; ea.cols = db->d.d_cols;
    ea.type = db->d.d_type; ea.tri = db->d.d_tri;
    ea.sN = sN; ea.sR = sR; ea.sC = sC; ea.sA = sA;
    ea.n_items = (uint32_t)n_items; ea.items = d_items;
    ea.row_cnt = d_rowcnt; ea.row_start = d_rowstart;
    ea.site_sum = d_sum; ea.site_cnt = d_cnt; ea.err_flag = d_err; ea.word_pairs = d_wordpairs;
    launch_emit_count(st, ea);
    launch_scan(st, d_rowcnt, d_rowstart, (uint32_t)n_items);
    HIPCHK(hipGetLastError());
    uint64_t n_rows = 0;
    HIPCHK(hipMemcpyAsync(&n_rows, d_rowstart + n_items, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    res->n_rows = n_rows;
    const size_t nr = (size_t)std::max<uint64_t>(n_rows, 1);
    if ((rc = pool.alloc((void**)&res->d_i, nr * 4))) return rc;
    if ((rc = pool.alloc((void**)&res->d_j, nr * 4))) return rc;
    if ((rc = pool.alloc((void**)&res->d_mi, nr * 8))) return rc;
    if (want_counts || want_p) if ((rc = pool.alloc((void**)&res->d_counts, nr * 36))) return rc;
    if (want_p) {
        if ((rc = pool.alloc((void**)&res->d_p, nr * 8))) return rc;
        if ((rc = pool.alloc((void**)&res->d_exceed, nr * 4))) return rc;
    }
    ea.out_i = res->d_i; ea.out_j = res->d_j; ea.out_mi = res->d_mi; ea.out_counts = res->d_counts;
    launch_emit_write(st, ea);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(ctx->ev[3], st));
    if (want_p && n_rows) {
        uint32_t* d_genlist; unsigned int* d_gencount;
        if ((rc = salloc((void**)&d_genlist, (size_t)n_rows * 4))) return rc;
        if ((rc = salloc((void**)&d_gencount, 4))) return rc;
        HIPCHK(hipMemsetAsync(d_gencount, 0, 4, st));
        launch_perm(st, n_rows, res->d_i, res->d_j, res->d_counts, ctx->d_G, ctx->d_LF, prm->n_shuffles,
                    prm->seed, res->d_p, res->d_exceed, d_genlist, d_gencount);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipEventRecord(ctx->ev[4], st));
    launch_site_mean(st, ns, d_sum, d_cnt, res->d_mean);
    if (ns) HIPCHK(hipMemcpyAsync(res->d_npairs, d_cnt, (size_t)ns * 4, hipMemcpyDeviceToDevice, st));
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(ctx->ev[5], st));
    int err = 0; unsigned long long wp = 0;
    HIPCHK(hipMemcpyAsync(&err, d_err, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(&wp, d_wordpairs, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (err) return fail(LGMI_E_DOMAIN, "math domain error: a pair with 0 common reads reached the MI (min_common == 0)");

    lgmi_run_info& inf = res->info;
    inf.n_rows = n_rows;
    inf.n_examined = pl.n_examined;
    inf.n_tile_pairs = (uint64_t)pl.tiles.size() * TILE * TILE + (uint64_t)pl.mtiles.size() * 128 * 128;
    inf.word_pairs = wp;
    inf.bytes_in = pl.bytes_in;
    inf.bytes_out = n_rows * (16ull + (want_p ? 8ull : 0ull) + (want_counts ? 36ull : 0ull));
    inf.n_count_launches = (pl.tiles.empty() ? 0 : 1) + (pl.mtiles.empty() ? 0 : 1);
    inf.n_mfma_tiles = (uint32_t)std::min<size_t>(pl.mtiles.size(), 0xFFFFFFFFu);
    inf.mfma_dtype = pl.mtiles.empty() ? 0u : (pl.mfma_fp4 ? 2u : 1u);
    inf.reserved = 0;
    HIPCHK(hipEventElapsedTime(&inf.ms_prep, ctx->ev[0], ctx->ev[1]));
    HIPCHK(hipEventElapsedTime(&inf.ms_count, ctx->ev[1], ctx->ev[2]));
    HIPCHK(hipEventElapsedTime(&inf.ms_emit, ctx->ev[2], ctx->ev[3]));
    HIPCHK(hipEventElapsedTime(&inf.ms_perm, ctx->ev[3], ctx->ev[4]));
    HIPCHK(hipEventElapsedTime(&inf.ms_mean, ctx->ev[4], ctx->ev[5]));
    HIPCHK(hipEventElapsedTime(&inf.ms_total, ctx->ev[0], ctx->ev[5]));
    if (!want_counts && res->d_counts) { pool.release(res->d_counts); res->d_counts = nullptr; }
    guard.r = nullptr;
    *out = res;
    return LGMI_OK;
}

extern "C" int lgmi_dresult_info(const lgmi_dresult* r, lgmi_run_info* out) {
    if (!r || !out) return fail(LGMI_E_ARG, "NULL argument");
    *out = r->info;
    return LGMI_OK;
}

extern "C" int lgmi_dresult_device_ptrs(const lgmi_dresult* r, lgmi_result* v) {
    if (!r || !v) return fail(LGMI_E_ARG, "NULL argument");
    memset(v, 0, sizeof *v);
    v->n_rows = r->n_rows; v->n_sites = r->n_sites;
    v->row_i = r->d_i; v->row_j = r->d_j; v->row_mi = r->d_mi;
    v->row_p = r->has_p ? r->d_p : nullptr;
    v->row_exceed = r->has_p ? r->d_exceed : nullptr;
    v->row_counts = r->has_counts ? r->d_counts : nullptr;
    v->site_mean_mi = r->d_mean; v->site_n_pairs = r->d_npairs;
    return LGMI_OK;
}

extern "C" void lgmi_result_free(lgmi_result* res) {
    if (!res) return;
    delete static_cast<ResultOwner*>(res->owner_);
    memset(res, 0, sizeof *res);
}

extern "C" int lgmi_dresult_fetch(lgmi_dresult* r, lgmi_result* out) {
    if (!r || !out) return fail(LGMI_E_ARG, "NULL argument");
    memset(out, 0, sizeof *out);
    HIPCHK(hipSetDevice(r->ctx->device));
    HostResult* h = new HostResult();
    struct Guard { HostResult* p; ~Guard() { delete p; } } guard{h};
    const size_t n = (size_t)r->n_rows, ns = (size_t)r->n_sites;
    hipStream_t st = r->ctx->stream;
    auto d2h = [&](void* dst, const void* src, size_t bytes) -> hipError_t {
        return bytes ? hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, st) : hipSuccess;
    };
    h->i.resize(n); h->j.resize(n); h->mi.resize(n); h->mean.resize(ns); h->npairs.resize(ns);
    HIPCHK(d2h(h->i.data(), r->d_i, n * 4));
    HIPCHK(d2h(h->j.data(), r->d_j, n * 4));
    HIPCHK(d2h(h->mi.data(), r->d_mi, n * 8));
    HIPCHK(d2h(h->mean.data(), r->d_mean, ns * 8));
    HIPCHK(d2h(h->npairs.data(), r->d_npairs, ns * 4));
    if (r->has_p) {
        h->p.resize(n); h->exceed.resize(n);
        HIPCHK(d2h(h->p.data(), r->d_p, n * 8));
        HIPCHK(d2h(h->exceed.data(), r->d_exceed, n * 4));
    }
    if (r->has_counts) {
        h->counts.resize(n * 9);
        HIPCHK(d2h(h->counts.data(), r->d_counts, n * 36));
    }
    HIPCHK(hipStreamSynchronize(st));
    out->n_rows = n; out->n_sites = ns;
    out->row_i = h->i.data(); out->row_j = h->j.data(); out->row_mi = h->mi.data();
    out->row_p = r->has_p ? h->p.data() : nullptr;
    out->row_exceed = r->has_p ? h->exceed.data() : nullptr;
    out->row_counts = r->has_counts ? h->counts.data() : nullptr;
    out->site_mean_mi = h->mean.data(); out->site_n_pairs = h->npairs.data();
    out->owner_ = static_cast<ResultOwner*>(h);
    guard.p = nullptr;
    return LGMI_OK;
}

extern "C" int lgmi_run(lgmi_ctx* ctx, const lgmi_batch* batch, const lgmi_params* prm, lgmi_result* out,
                        lgmi_run_info* info) {
    if (!out) return fail(LGMI_E_ARG, "out is NULL");
    memset(out, 0, sizeof *out);
    lgmi_dbatch* db = nullptr;
    int rc = lgmi_batch_upload(ctx, batch, &db);
    if (rc) return rc;
    lgmi_dresult* dr = nullptr;
    rc = lgmi_run_device(ctx, db, prm, &dr);
    if (!rc) {
        if (info) *info = dr->info;
        rc = lgmi_dresult_fetch(dr, out);
    }
    lgmi_dresult_free(dr);
    lgmi_dbatch_free(db);
    return rc;
}

// ---------------------------------------------------------------- mean of caller rows
extern "C" int lgmi_site_mean(lgmi_ctx* ctx, uint64_t n_rows, const uint32_t* row_i, const uint32_t* row_j,
                              const double* row_mi, uint64_t n_sites, double* mean_out, uint32_t* n_out) {
    if (!ctx || !mean_out || !n_out || (n_rows && (!row_i || !row_j || !row_mi)))
        return fail(LGMI_E_ARG, "NULL argument");
    for (uint64_t r = 0; r < n_rows; ++r)
        if (row_i[r] >= n_sites || row_j[r] >= n_sites) return fail(LGMI_E_ARG, "row %llu: site index out of range", (unsigned long long)r);
    HIPCHK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    Pool& pool = ctx->pool;
    std::vector<void*> scratch;
    struct Guard { Pool& p; std::vector<void*>& s; ~Guard() { for (void* q : s) p.release(q); } } guard{pool, scratch};
    auto salloc = [&](void** p, size_t bytes) { int e = pool.alloc(p, bytes); if (!e) scratch.push_back(*p); return e; };
    uint32_t *di, *dj, *dc; double *dm, *dmean; unsigned long long* ds;
    int rc;
    if ((rc = salloc((void**)&di, n_rows * 4))) return rc;
    if ((rc = salloc((void**)&dj, n_rows * 4))) return rc;
    if ((rc = salloc((void**)&dm, n_rows * 8))) return rc;
    if ((rc = salloc((void**)&ds, n_sites * 8))) return rc;
    if ((rc = salloc((void**)&dc, n_sites * 4))) return rc;
    if ((rc = salloc((void**)&dmean, n_sites * 8))) return rc;
    if (n_rows) {
        HIPCHK(hipMemcpyAsync(di, row_i, n_rows * 4, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(dj, row_j, n_rows * 4, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(dm, row_mi, n_rows * 8, hipMemcpyHostToDevice, st));
    }
    if (n_sites) {
        HIPCHK(hipMemsetAsync(ds, 0, n_sites * 8, st));
        HIPCHK(hipMemsetAsync(dc, 0, n_sites * 4, st));
    }
    launch_rows_mean(st, n_rows, di, dj, dm, ds, dc);
    launch_site_mean(st, (uint32_t)n_sites, ds, dc, dmean);
    HIPCHK(hipGetLastError());
    if (n_sites) {
        HIPCHK(hipMemcpyAsync(mean_out, dmean, n_sites * 8, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(n_out, dc, n_sites * 4, hipMemcpyDeviceToHost, st));
    }
    HIPCHK(hipStreamSynchronize(st));
    return LGMI_OK;
}

// ---------------------------------------------------------------- ECDF (the reference's `mip`)
extern "C" int lgmi_ecdf(lgmi_ctx* ctx, uint64_t n_ref, const double* ref, uint64_t n_query, const double* query,
                         double* out) {
    if (!ctx || (n_query && (!query || !out))) return fail(LGMI_E_ARG, "NULL argument");
    if (n_ref == 0 || !ref) return fail(LGMI_E_ARG, "empty reference sample (the reference's ecdf divides by zero)");
    if (n_ref >= 0x7FFFFFFFull) return fail(LGMI_E_ARG, "reference sample too large");
    for (uint64_t k = 0; k < n_ref; ++k) if (ref[k] != ref[k]) return fail(LGMI_E_ARG, "NaN in the reference sample");
    HIPCHK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    Pool& pool = ctx->pool;
    std::vector<void*> scratch;
    struct Guard { Pool& p; std::vector<void*>& s; ~Guard() { for (void* q : s) p.release(q); } } guard{pool, scratch};
    auto salloc = [&](void** p, size_t bytes) { int e = pool.alloc(p, bytes); if (!e) scratch.push_back(*p); return e; };
    double *d_ref, *d_sorted, *d_q, *d_out; void* d_temp;
    const size_t tb = ecdf_sort_temp_bytes((uint32_t)n_ref);
    int rc;
    if ((rc = salloc((void**)&d_ref, n_ref * 8))) return rc;
    if ((rc = salloc((void**)&d_sorted, n_ref * 8))) return rc;
    if ((rc = salloc((void**)&d_q, n_query * 8))) return rc;
    if ((rc = salloc((void**)&d_out, n_query * 8))) return rc;
    if ((rc = salloc(&d_temp, tb))) return rc;
    HIPCHK(hipMemcpyAsync(d_ref, ref, n_ref * 8, hipMemcpyHostToDevice, st));
    if (n_query) HIPCHK(hipMemcpyAsync(d_q, query, n_query * 8, hipMemcpyHostToDevice, st));
    HIPCHK(launch_ecdf(st, (uint32_t)n_ref, d_ref, d_sorted, d_temp, tb, n_query, d_q, d_out));
    if (n_query) HIPCHK(hipMemcpyAsync(out, d_out, n_query * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return LGMI_OK;
}
