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
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_set>
#include <sys/mman.h>
#include <vector>

#include <chrono>

#include "lgmi_internal.h"
#include "philox.h"
#include "plan.h"

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


// Waiting for a stream: hipStreamQuery polled (back to back for the first 300 us, then every ~30 us) instead of
// hipStreamSynchronize.  On some processes of the
// same box the runtime's blocking wait woke up ~25 ms after a 450-ms stream had drained (every step of the run, none
// of the next run's: bench.py's step_wall_ms against the event time); a bounded poll does not depend on how the
// runtime chose to wait, and costs a few thousand cheap queries per second of waiting.
static hipError_t wait_stream(hipStream_t st) {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t e = hipStreamQuery(st);
        if (e != hipErrorNotReady) return e;
        // short waits (a banded batch is done in a millisecond or two) are spun through; longer ones sleep between polls
        if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(300))
            std::this_thread::sleep_for(std::chrono::microseconds(30));
    }
}

// LGMI_TRACE_HOST=1: host-side milestones of a run on stderr (ms since the call was entered) — where wall time that no
// stream event sees goes
struct HostTrace {
    bool on; std::chrono::steady_clock::time_point t0; std::string line;
    HostTrace() : on(getenv("LGMI_TRACE_HOST") != nullptr), t0(std::chrono::steady_clock::now()) {}
    void mark(const char* what) {
        if (!on) return;
        char b[64];
        snprintf(b, sizeof b, " %s=%.2f", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        line += b;
    }
    ~HostTrace() { if (on) { mark("exit"); fprintf(stderr, "[lgmi host]%s\n", line.c_str()); } }
};

// ---------------------------------------------------------------- device memory pool
// Grow-only cache of device allocations so that repeated runs (bench steps) do not
// pay hipMalloc/hipFree inside the timed region.
struct Pool {
    std::multimap<size_t, void*> free_;
    std::map<void*, size_t> live_;
    int alloc(void** out, size_t bytes) {
        size_t n = std::max<size_t>(256, (bytes + 255) & ~size_t(255));
        // LGMI_POOL_POISON=1 (debugging aid): every block handed out is filled with 0xA5 first, device-synchronously, so that
        // a kernel that reads what it never wrote fails the same way every time instead of once in a hundred runs
        static const bool poison = getenv("LGMI_POOL_POISON") != nullptr;
        auto it = free_.lower_bound(n);
        if (it != free_.end() && it->first <= 2 * n + (1u << 20)) {
            *out = it->second;
            live_[it->second] = it->first;
            const size_t got = it->first;
            free_.erase(it);
            if (poison) { (void)hipDeviceSynchronize(); (void)hipMemset(*out, 0xA5, got); (void)hipDeviceSynchronize(); }
            return LGMI_OK;
        }
        void* p = nullptr;
        static const bool log = getenv("LGMI_POOL_LOG") != nullptr;   // debugging aid: every pool miss, on stderr
        if (log) fprintf(stderr, "[lgmi pool] hipMalloc %zu bytes (%zu free blocks cached)\n", n, free_.size());
        hipError_t e = hipMalloc(&p, n);
        if (e != hipSuccess) {  // give cached blocks back and retry once
            (void)hipGetLastError();                         // (the failed call's error is sticky: a later hipGetLastError() after a
            trim();                                          //  kernel launch would report this out-of-memory as the launch's)
            e = hipMalloc(&p, n);
            if (e != hipSuccess) { (void)hipGetLastError(); return fail(LGMI_E_OOM, "hipMalloc(%zu bytes): %s", n, hipGetErrorString(e)); }
        }
        live_[p] = n;
        *out = p;
        if (poison) { (void)hipDeviceSynchronize(); (void)hipMemset(p, 0xA5, n); (void)hipDeviceSynchronize(); }
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

// Grow-only cache of PINNED host buffers for fetched results.  A fresh pageable destination copies at 12 - 15 GB/s instead
// of 57: a caller that fetches result after result (every footprint batch of a run) pays the pinning once.  Shared between
// the context and the results it handed out, so either may die first.
// How a buffer is pinned (round 4, tools/ubench_pin.hip on the box): hipHostMalloc of 400 MB takes 52 - 72 ms — 50 of them
// the kernel faulting in 100,000 small pages one by one, which eight threads do not speed up — and was 60 of the 91 - 109 ms
// of a context's first lgmi_run on the footprint batch.  The same 400 MB as 2-MB-aligned memory with MADV_HUGEPAGE
// (transparent huge pages are in `madvise` mode on these boxes), touched by eight threads and then registered with
// hipHostRegister: 2.9 + 0.8 ms, and a D2H copy into it runs at the same 57 GB/s.  Without huge pages the same path costs
// 35 + 13 ms: still below hipHostMalloc, which stays the fallback (and the path for small buffers).
struct PinnedPool {
    std::mutex mu;
    std::multimap<size_t, void*> free_;
    std::unordered_set<void*> registered_;                 // buffers of the aligned_alloc + hipHostRegister kind
    static constexpr size_t HUGE_PAGE = size_t(2) << 20;
    void* pin_new(size_t n, size_t* got) {
        void* p = nullptr;
        static const bool plain = [] { const char* e = getenv("LGMI_PINNED"); return e && !strcmp(e, "hostmalloc"); }();
        if (n >= 4 * HUGE_PAGE && !plain) {
            const size_t n2 = (n + HUGE_PAGE - 1) & ~(HUGE_PAGE - 1);
            p = aligned_alloc(HUGE_PAGE, n2);
            if (p) {
                (void)madvise(p, n2, MADV_HUGEPAGE);            // (advice: the path is the same without it, only slower)
                const unsigned T = (unsigned)std::max<size_t>(1, std::min<size_t>(8, n2 >> 24));
                char* const c = static_cast<char*>(p);
                std::vector<std::thread> th;
                auto touch = [c, n2, T](unsigned t) { for (size_t i = n2 * t / T; i < n2 * (t + 1) / T; i += 4096) c[i] = 0; };
                unsigned started = 1;                      // (a thread the system refuses — a container's pid limit — is a range this thread touches itself)
                for (; started < T; ++started) { try { th.emplace_back(touch, started); } catch (...) { break; } }
                touch(0u);
                for (unsigned t = started; t < T; ++t) touch(t);
                for (auto& x : th) x.join();
                if (hipHostRegister(p, n2, hipHostRegisterDefault) == hipSuccess) {
                    std::lock_guard<std::mutex> lk(mu);
                    registered_.insert(p);
                    *got = n2;
                    return p;
                }
                (void)hipGetLastError();
                free(p);
            }
        }
        if (hipHostMalloc(&p, n, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        *got = n;
        return p;
    }
    void* take(size_t bytes, size_t* got) {
        const size_t n = std::max<size_t>(4096, (bytes + 4095) & ~size_t(4095));
        static const bool poison = getenv("LGMI_POOL_POISON") != nullptr;      // (see Pool::alloc)
        void* p = nullptr;
        {
            std::lock_guard<std::mutex> lk(mu);
            auto it = free_.lower_bound(n);
            if (it != free_.end() && it->first <= 2 * n + (1u << 20)) { p = it->second; *got = it->first; free_.erase(it); }
        }
        if (!p) p = pin_new(n, got);
        if (p && poison) memset(p, 0xA5, *got);
        return p;
    }
    void give(void* p, size_t n) { if (p) { std::lock_guard<std::mutex> lk(mu); free_.emplace(n, p); } }
    ~PinnedPool() {
        for (auto& kv : free_) {
            if (registered_.count(kv.second)) { (void)hipHostUnregister(kv.second); free(kv.second); }
            else (void)hipHostFree(kv.second);
        }
    }
};

// ---------------------------------------------------------------- objects
struct lgmi_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev[8] = {};
    unsigned long long* h_scal = nullptr;   // 8 KB of pinned words: [0, 8) the scalars a run reads back (a pageable destination
                                            // makes every small copy a staged, blocking one), [8, 512) comm.cpp's small exchanges
    Pool pool;
    long long* d_G = nullptr;   // round(n ln n * 2^28): permutation statistic (perm.hip)
    double* d_LF = nullptr;     // ln n!
    uint32_t tables_len = 0;
    void* comm = nullptr;       // ncclComm_t (comm.cpp)
    hipStream_t comm_stream = nullptr;   // the gather runs here, beside the kernels of `stream`
    hipStream_t copy_stream = nullptr;   // lgmi_run's copies (planes up in pieces, rows down under the permutation stage): a stream
                                         // of its own, never the one RCCL's collectives were queued on
    hipStream_t prep_stream = nullptr;   // layout prep of a pipelined upload's pieces (made on first use, kept: making and
                                         // destroying a stream per call was milliseconds of every lgmi_run)
    hipEvent_t comm_event = nullptr;     // main stream -> communication stream ordering (comm.cpp: comm_after_main)
    int rank = 0, world = 1;
    size_t mem_total = 0;       // device memory, for the "allocate rows by their upper bound" decision
    std::shared_ptr<PinnedPool> pinned = std::make_shared<PinnedPool>();
    // events of chunked stages, made on first use.  Who uses which: the permutation stage's row ranges (3 each, at most 32
    // ranges), lgmi_run's transfers behind them, the pieces of a pipelined upload (at most 64) and their trace twins, the
    // copy -> prep hand-over of a piece, the tri-flag check, main stream -> copy stream ordering
    enum { EV_PERM = 0, EV_XFER = 96, EV_UP = 128, EV_UPDBG = 192, EV_COPIED = 256, EV_CHECKED = 288, EV_ORDER = 289 };
    std::vector<hipEvent_t> more_ev;
    hipEvent_t event(size_t k) {           // NULL when the runtime refuses one
        while (more_ev.size() <= k) { hipEvent_t e = nullptr; if (hipEventCreate(&e) != hipSuccess) { (void)hipGetLastError(); return nullptr; } more_ev.push_back(e); }
        return more_ev[k];
    }
};

// A plan is a function of the resident batch and a few parameters, not of the run: it is kept with the batch, host
// vectors and device copies, so that a second run on the same batch neither plans nor uploads again.  (Round 3: on 20,000
// footprint-sized blocks the planner's 29 ms and the 60 MB plan upload were most of a 97 ms step whose kernels take
// 22 ms; a dense chromosome's plan is 1 - 2 ms.)  A few entries: the sequential shards of a self-splitting run cycle.
struct PlanCache {
    bool het_only = false; uint32_t sh_rank = 0, sh_world = 1, n_shuffles_key = 0, xg = 0; int ck = 0;
    uint64_t stamp = 0;
    float ms_build = 0.f;
    Plan pl;
    bool on_device = false;
    BlockPlan* d_plans = nullptr; uint32_t* d_xlist = nullptr; uint32_t* d_ylist = nullptr; SiteMap* d_smap = nullptr;
    Tile* d_tiles = nullptr; Tile* d_mtiles = nullptr; uint2* d_items = nullptr; uint2* d_units = nullptr;
    OpGroup* d_opgroups = nullptr; uint32_t* d_xrows = nullptr;
    // the FP4 kernel's operands in wave-load order (k_gather_ops): a function of the resident batch and this plan, like
    // the plan itself — kept with it (round 5) instead of being re-laid by every run: 1.6 ms of a 206-ms north-star step,
    // 1.2 of a 34-ms shard.  (The pool is grow-only: as scratch the same bytes stayed allocated between runs anyway.)
    uint4* d_ops = nullptr; bool ops_ready = false;
    void release(Pool& p) {
        p.release(d_plans); p.release(d_xlist); p.release(d_ylist); p.release(d_smap); p.release(d_tiles);
        p.release(d_mtiles); p.release(d_items); p.release(d_units); p.release(d_opgroups); p.release(d_xrows); p.release(d_ops);
        d_plans = nullptr; d_xlist = d_ylist = nullptr; d_smap = nullptr; d_tiles = d_mtiles = nullptr;
        d_items = d_units = nullptr; d_opgroups = nullptr; d_xrows = nullptr; d_ops = nullptr; on_device = false; ops_ready = false;
    }
};

// what the compact row form needs of a batch to name a row's sites (include/lgmi.h: candidates of a site): shared by
// the resident batch, its results and the host results fetched from them, so any of them may be freed first
struct SiteTable {
    std::vector<uint64_t> block_site_begin;
    std::vector<uint8_t> type;
};

struct lgmi_dbatch {
    lgmi_ctx* ctx = nullptr;
    DevBatch d;
    std::shared_ptr<const SiteTable> table;
    // how many sequential shards a run of this batch needs (split_count builds k shard plans per candidate k: 6 x 33 ms at
    // 150k x 200k on every call, advice r4) — remembered per (parameters, budget)
    struct SplitMemo { bool het_only, want_p, keep_p, want_counts; uint32_t n_shuffles; double budget; int k; uint64_t largest; };
    mutable std::vector<SplitMemo> split_memo;
    mutable std::vector<std::unique_ptr<PlanCache>> plans;   // most recently used plans of this batch (at most PLAN_CACHE_N)
    mutable uint64_t plan_stamp = 0;
    // host copies of the site metadata (planning happens on the host)
    std::vector<uint64_t> block_site_begin;
    std::vector<uint32_t> block_n_reads;
    std::vector<int64_t> pos;
    std::vector<uint8_t> type, tri;
    PodVec<Col> cols;                    // [n_cols]  (PodVec: written once, in parallel — a value-initialising resize of 27 MB
    PodVec<uint32_t> pseudo_site;        // site of each pseudo column          and a second one that moved it were 4 of the
    PodVec<uint32_t> pseudo_of_site;     // column id of the site's pseudo column or NONE       upload's 11 ms on 1.1 M sites)
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
    uint4* d_rec = nullptr;                // per row what the permutation stage needs (EmitArgs::out_rec); gone once it has run
    double* d_mean = nullptr; uint32_t* d_npairs = nullptr;
    unsigned long long* d_sum = nullptr;   // per-site sum of MI in 2^-40 fixed point (what d_mean was made from)
    bool has_p = false, has_counts = false;
    bool sharded = false;                  // per-site figures cover this shard's rows only
    uint32_t n_shuffles = 0;
    bool p_from_exceed = false;            // every row_p is (1 + row_exceed) / (n_shuffles + 1): the gather need not carry it
    // split run (lgmi_run_device_rows + lgmi_dresult_permute): the permutation stage is still to come
    bool perm_pending = false;
    uint64_t* d_nrows = nullptr;           // the row count on the device, for the permutation kernels' grids
    uint64_t cap_rows = 0;
    lgmi_params prm = {};
    // the compact row form (ABI 6): per-site integers that add up over shards, and what a compact fetch derives from them
    std::shared_ptr<const SiteTable> table;     // NULL: gathered from ranks that ran different batches
    bool het_only = false;
    uint32_t first_site = 0xFFFFFFFFu, first_seg = 0;   // the shard's first work item (site, segment of LGMI_EMIT_SEG partners): NONE when it has none
    uint32_t* d_nfirst = nullptr;          // [n_sites] rows whose FIRST site is s
    uint32_t* d_ncand = nullptr;           // [n_sites] pairs s could have emitted as first site
    uint8_t* d_full = nullptr;             // [n_sites]   (from here: made by compact_prepare)
    uint64_t* d_row_begin = nullptr;       // [n_sites + 1]
    uint32_t* d_jlisted = nullptr; uint64_t n_jlisted = 0;
    uint16_t* d_exceed16 = nullptr;
    bool compact_ready = false;
};

struct HostResult : ResultOwner {  // owner_ of a host lgmi_result: pinned buffers that go back to the context's cache
    std::shared_ptr<PinnedPool> pool;
    std::shared_ptr<const SiteTable> table; bool het_only = false;     // compact form: lgmi_result_expand_rows reads these
    std::vector<std::pair<void*, size_t>> bufs;
    template <class T> T* take(size_t count) {
        size_t got = 0;
        void* p = pool->take(std::max<size_t>(count, 1) * sizeof(T), &got);
        if (p) bufs.emplace_back(p, got);
        return static_cast<T*>(p);
    }
    ~HostResult() override { for (auto& b : bufs) pool->give(b.first, b.second); }
};

// ---------------------------------------------------------------- basics
extern "C" int lgmi_abi_version(void) { return LGMI_ABI_VERSION; }
extern "C" const char* lgmi_last_error(void) { return g_err.c_str(); }
extern "C" size_t lgmi_struct_size(int which) {
    switch (which) {
        case 0: return sizeof(lgmi_batch);
        case 1: return sizeof(lgmi_params);
        case 2: return sizeof(lgmi_result);
        case 3: return sizeof(lgmi_run_info);
        case 4: return sizeof(lgmi_synth_spec);
        case 5: return sizeof(lgmi_shard_plan);
        case 6: return sizeof(lgmi_gather_opts);
        case 7: return sizeof(lgmi_comm_info_t);
        default: return 0;
    }
}

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
    HIPCHK(hipHostMalloc((void**)&c->h_scal, 8192, hipHostMallocDefault));   // words [512, 1024): the chunked permutation stage
    size_t free_b = 0;
    HIPCHK(hipMemGetInfo(&free_b, &c->mem_total));
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
    for (auto& ev : ctx->more_ev) if (ev) (void)hipEventDestroy(ev);
    if (ctx->h_scal) (void)hipHostFree(ctx->h_scal);
    if (ctx->comm_stream) { (void)hipStreamSynchronize(ctx->comm_stream); (void)hipStreamDestroy(ctx->comm_stream); }
    if (ctx->copy_stream) { (void)hipStreamSynchronize(ctx->copy_stream); (void)hipStreamDestroy(ctx->copy_stream); }
    if (ctx->prep_stream) { (void)hipStreamSynchronize(ctx->prep_stream); (void)hipStreamDestroy(ctx->prep_stream); }
    if (ctx->comm_event) (void)hipEventDestroy(ctx->comm_event);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

// accessors for comm.cpp (keeps lgmi_ctx private to this file)
namespace lgmi {
hipStream_t ctx_stream(lgmi_ctx* c) { return c->stream; }
hipStream_t ctx_comm_stream(lgmi_ctx* c) {               // created on first use
    if (!c->comm_stream && hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking) != hipSuccess) c->comm_stream = nullptr;
    return c->comm_stream ? c->comm_stream : c->stream;
}
static hipStream_t ctx_prep_stream(lgmi_ctx* c) {        // NULL when the runtime refuses one (the caller then uses the copy stream)
    if (!c->prep_stream && hipStreamCreateWithFlags(&c->prep_stream, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); c->prep_stream = nullptr; }
    return c->prep_stream;
}
hipStream_t ctx_copy_stream(lgmi_ctx* c) {               // created on first use; the main stream when the runtime refuses one
    if (!c->copy_stream && hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); c->copy_stream = nullptr; }
    return c->copy_stream ? c->copy_stream : c->stream;
}
hipEvent_t ctx_comm_event(lgmi_ctx* c) {                 // created on first use
    if (!c->comm_event && hipEventCreateWithFlags(&c->comm_event, hipEventDisableTiming) != hipSuccess) c->comm_event = nullptr;
    return c->comm_event;
}
int ctx_device(lgmi_ctx* c) { return c->device; }
void** ctx_comm_slot(lgmi_ctx* c) { return &c->comm; }
uint64_t* ctx_pinned_words(lgmi_ctx* c, size_t* n_words) { *n_words = 504; return (uint64_t*)(c->h_scal + 8); }
int* ctx_rank_slot(lgmi_ctx* c) { return &c->rank; }
int* ctx_world_slot(lgmi_ctx* c) { return &c->world; }
int set_error(int code, const char* msg) { return fail(code, "%s", msg); }
Pool& ctx_pool(lgmi_ctx* c) { return c->pool; }
void dresult_view(const lgmi_dresult* r, DResultView* v) {
    v->n_rows = r->n_rows; v->n_sites = r->n_sites;
    v->i = r->d_i; v->j = r->d_j; v->mi = r->d_mi;
    v->p = r->has_p ? r->d_p : nullptr; v->exceed = r->has_p ? r->d_exceed : nullptr;     // (p may be NULL: no_row_p)
    v->counts = r->has_counts ? r->d_counts : nullptr;
    v->mean = r->d_mean; v->npairs = r->d_npairs; v->sum = r->d_sum;
    v->n_shuffles = r->n_shuffles; v->p_from_exceed = r->p_from_exceed;
    v->info = r->info;
    v->nfirst = r->d_nfirst; v->ncand = r->d_ncand; v->first_site = r->first_site; v->first_seg = r->first_seg;
    if (r->table) { v->block_site_begin = &r->table->block_site_begin; v->site_type = &r->table->type; v->table_owner = &r->table; }
    v->het_only = r->het_only;
}
// a resident result made by the gather (comm.cpp): arrays come from the context's pool
lgmi_dresult* dresult_new_gathered(lgmi_ctx* c, const DResultView& v) {
    lgmi_dresult* r = new lgmi_dresult();
    r->ctx = c; r->n_rows = v.n_rows; r->n_sites = v.n_sites;
    r->d_i = const_cast<uint32_t*>(v.i); r->d_j = const_cast<uint32_t*>(v.j); r->d_mi = const_cast<double*>(v.mi);
    r->d_p = const_cast<double*>(v.p); r->d_exceed = const_cast<uint32_t*>(v.exceed);
    r->d_counts = const_cast<uint32_t*>(v.counts);
    r->d_mean = const_cast<double*>(v.mean); r->d_npairs = const_cast<uint32_t*>(v.npairs);
    r->d_sum = const_cast<unsigned long long*>(v.sum);
    r->has_p = v.exceed != nullptr; r->has_counts = v.counts != nullptr;
    r->n_shuffles = v.n_shuffles; r->p_from_exceed = v.p_from_exceed;
    r->info = v.info;
    // a gather of shards of one batch keeps the compact form within reach: the summed per-site row counts and the site table
    r->d_nfirst = const_cast<uint32_t*>(v.nfirst); r->d_ncand = const_cast<uint32_t*>(v.ncand);
    if (v.table_owner) r->table = *static_cast<const std::shared_ptr<const SiteTable>*>(v.table_owner);
    r->het_only = v.het_only;
    return r;
}
int pool_alloc(lgmi_ctx* c, void** out, size_t bytes) { return c->pool.alloc(out, bytes); }
void pool_release(lgmi_ctx* c, void* p) { c->pool.release(p); }
}  // namespace lgmi

extern "C" int lgmi_ctx_synchronize(lgmi_ctx* ctx) {
    if (!ctx) return fail(LGMI_E_ARG, "ctx is NULL");
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(wait_stream(ctx->stream));
    return LGMI_OK;
}

// ---------------------------------------------------------------- upload
static void free_dbatch_device(lgmi_dbatch* db) {
    if (!db) return;
    // (from the context's grow-only pool since round 4: a host that uploads batch after batch — one per chunk of footprints —
    //  paid seven hipMalloc and as many hipFree, each a device synchronisation, per upload)
    Pool& pool = db->ctx->pool;
    pool.release(db->d.d_pos); pool.release(db->d.d_type); pool.release(db->d.d_tri);
    pool.release(db->d.d_cols); pool.release(db->d.d_cplanes);
    db->d = DevBatch();
}

extern "C" void lgmi_dbatch_free(lgmi_dbatch* db) {
    if (!db) return;
    (void)hipSetDevice(db->ctx->device);
    (void)hipStreamSynchronize(db->ctx->stream);
    for (auto& pc : db->plans) pc->release(db->ctx->pool);
    db->plans.clear();
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
    // one block's checks; says == false only finds out whether the block is good (the threads of a large batch), true
    // reports the first defect through fail() (on the caller's thread: the message is thread-local)
    auto check_block = [&](uint64_t k, bool says) -> int {
        uint64_t sb = b->block_site_begin[k], se = b->block_site_begin[k + 1];
        if (se < sb || se > b->n_sites) return says ? fail(LGMI_E_ARG, "block_site_begin not monotone at block %llu", (unsigned long long)k) : LGMI_E_ARG;
        uint64_t W = ((uint64_t)b->block_n_reads[k] + 63) / 64;
        for (uint64_t s = sb; s < se; ++s) {
            const bool bad_type = b->site_type[s] > 2, bad_band = (uint64_t)b->site_word_off[s] + b->site_n_words[s] > W,
                       bad_planes = b->site_plane_off[s] + 2ull * b->site_n_words[s] > b->n_plane_words,
                       bad_pos = s > sb && b->site_pos[s] <= b->site_pos[s - 1];
            if (!(bad_type | bad_band | bad_planes | bad_pos)) continue;
            if (!says) return LGMI_E_ARG;
            if (bad_type) return fail(LGMI_E_ARG, "site %llu: type %u", (unsigned long long)s, b->site_type[s]);
            if (bad_band)
                return fail(LGMI_E_ARG, "site %llu: band [%u,+%u) exceeds %llu words of block %llu",
                            (unsigned long long)s, b->site_word_off[s], b->site_n_words[s], (unsigned long long)W,
                            (unsigned long long)k);
            if (bad_planes) return fail(LGMI_E_ARG, "site %llu: planes exceed n_plane_words", (unsigned long long)s);
            return fail(LGMI_E_ARG, "site %llu: positions must increase strictly inside a block", (unsigned long long)s);
        }
        return LGMI_OK;
    };
    const unsigned T_want = b->n_sites >= (1u << 16) ? plan_threads(b->n_blocks) : 1u;
    if (T_want <= 1) {
        for (uint64_t k = 0; k < b->n_blocks; ++k) { const int rc = check_block(k, true); if (rc) return rc; }
        return LGMI_OK;
    }
    // a million sites were 1 ms of a one-shot call on one thread: blocks in ranges of equal site counts, the first bad block
    // of every range, the earliest of them reported exactly as the serial loop would
    Team team(T_want);
    const unsigned T = team.T;                              // (what the system granted: plan.h)
    std::vector<uint64_t> first_bad(T, ~0ull);
    team.run([&](unsigned t) {
        const uint64_t s_lo = b->n_sites * t / T, s_hi = b->n_sites * (t + 1) / T;
        // the blocks whose first site lies in [s_lo, s_hi): block_site_begin is only trusted as far as it has been checked,
        // so the range is found by a bounded binary search over indices and every block re-checks its own bounds
        auto lower = [&](uint64_t v) { uint64_t lo = 0, hi = b->n_blocks; while (lo < hi) { const uint64_t mid = (lo + hi) / 2; if (b->block_site_begin[mid] < v) lo = mid + 1; else hi = mid; } return lo; };
        const uint64_t k0 = t == 0 ? 0 : lower(s_lo), k1 = t + 1 == T ? b->n_blocks : lower(s_hi);
        for (uint64_t k = k0; k < k1; ++k) if (check_block(k, false)) { first_bad[t] = k; break; }
    });
    // (a non-monotone block_site_begin can make the ranges overlap or leave gaps: the blocks are then checked one by one)
    bool monotone = true;
    for (uint64_t k = 0; k < b->n_blocks && monotone; ++k) monotone = b->block_site_begin[k] <= b->block_site_begin[k + 1];
    if (!monotone) { for (uint64_t k = 0; k < b->n_blocks; ++k) { const int rc = check_block(k, true); if (rc) return rc; } return LGMI_OK; }
    uint64_t kb = ~0ull;
    for (unsigned t = 0; t < T; ++t) kb = std::min(kb, first_bad[t]);
    if (kb != ~0ull) return check_block(kb, true);
    return LGMI_OK;
}

template <class T>
static int dev_copy_new(Pool& pool, T** dptr, const T* h, size_t n, hipStream_t st) {
    *dptr = nullptr;
    if (!n) return LGMI_OK;
    int rc = pool.alloc((void**)dptr, n * sizeof(T));
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(*dptr, h, n * sizeof(T), hipMemcpyHostToDevice, st));
    return LGMI_OK;
}

// (Round 3 tried a staged upload of the bit planes — threads copying 32-MB pieces into pinned buffers ahead of the DMA
//  engine — against the plain hipMemcpyAsync from the caller's pageable array: 700 ms host-to-host against 495 ms on the
//  same box, gpurun_out/exp_upload.txt; the runtime's own path moves the 2.5 GB at ~50 GB/s.  Not kept.)
//
// Round 5: the PIPELINED upload of lgmi_run.  With the caller's tri flags (lgmi_batch.site_tri) the column table and the
// plan need nothing of the planes, so everything but the planes goes up first and the planes follow in pieces of sites
// on a stream of their own, each piece's layout prep behind it and an event behind that; the count kernels of the tiles
// whose rows and columns have all arrived run on the main stream meanwhile (run_device_impl).  The 44 ms the 2.5 GB of
// a north-star chromosome take over PCIe then hide under the 97 ms of its count stage.
struct UploadPipe {
    lgmi_ctx* ctx = nullptr;
    const lgmi_batch* b = nullptr;
    hipStream_t us = nullptr;                      // the upload stream (the context's copy stream)
    hipStream_t ps = nullptr;                      // the prep kernels' stream (the context's)
    uint32_t K = 0;
    std::vector<uint64_t> site_begin, word_begin, pseudo_begin;   // [K + 1] chunk k = sites [site_begin[k], site_begin[k + 1]), their plane words, their pseudo columns
    uint64_t* d_planes = nullptr; uint64_t* d_poff = nullptr; uint32_t* d_nw = nullptr; uint32_t* d_pseudo = nullptr;
    uint8_t* d_tri_chk = nullptr; int* d_bad = nullptr;
    // the plan's tiles and operand groups, reordered by the chunk that completes them: chunk k's are [begin[k], begin[k + 1])
    std::vector<uint32_t> mt_begin, t_begin, og_begin;
    uint32_t next = 0;                             // chunks already queued
    ~UploadPipe() {
        if (!ctx) return;
        if (us) (void)hipStreamSynchronize(us);
        if (ps) (void)hipStreamSynchronize(ps);
        (void)hipStreamSynchronize(ctx->stream);
        Pool& p = ctx->pool;
        p.release(d_planes); p.release(d_poff); p.release(d_nw); p.release(d_pseudo); p.release(d_tri_chk); p.release(d_bad);
    }
};

// is the batch one the pipelined upload takes?  (flags given, planes packed in site order, enough of them to matter)
static bool pipe_eligible(const lgmi_batch* b) {
    if (!b || !b->site_tri || b->n_sites < 64 || getenv("LGMI_NO_UPLOAD_PIPE")) return false;
    // from 512 MB of planes on (their copy takes ~10 ms): below that the pieces' fixed costs — eight copies, prep launches and
    // count launches instead of one each, the plan's tiles put in piece order — cost more than the copy they hide (20,000
    // footprint blocks, 150 MB: 29 against 21 ms per call; a 10k x 50k chromosome, 125 MB: 11.9 against 11.3)
    uint64_t min_words = 64ull << 20;
    if (const char* e = getenv("LGMI_UPLOAD_PIPE_MIN_WORDS")) min_words = strtoull(e, nullptr, 10);
    if (b->n_plane_words < min_words) return false;
    uint64_t off = 0;
    for (uint64_t s = 0; s < b->n_sites; ++s) { if (b->site_plane_off[s] != off) return false; off += 2ull * b->site_n_words[s]; }
    return off == b->n_plane_words;
}

static int upload_impl(lgmi_ctx* ctx, const lgmi_batch* b, lgmi_dbatch** out, UploadPipe* pipe) {
    if (!ctx || !out) return fail(LGMI_E_ARG, "ctx/out is NULL");
    *out = nullptr;
    HostTrace tr;
    int rc = validate_batch(b);
    if (rc) return rc;
    tr.mark("upload:validated");
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
    { auto t = std::make_shared<SiteTable>(); t->block_site_begin = db->block_site_begin; t->type = db->type; db->table = t; }
    tr.mark("host_copies");

    hipStream_t st = ctx->stream;
    uint64_t* d_planes = nullptr; uint64_t* d_poff = nullptr; uint32_t* d_nw = nullptr; uint32_t* d_pseudo = nullptr;
    Pool& pool = ctx->pool;
    struct Tmp { Pool& p; hipStream_t st; uint64_t** a; uint64_t** b; uint32_t** c; uint32_t** d; bool keep = false;     // (released once the stream is past them)
                 ~Tmp() { if (keep) return; (void)hipStreamSynchronize(st); p.release(*a); p.release(*b); p.release(*c); p.release(*d); } } tmp{pool, st, &d_planes, &d_poff, &d_nw, &d_pseudo};
    if (pipe) {
        if ((rc = pool.alloc((void**)&d_planes, std::max<uint64_t>(b->n_plane_words, 1) * 8))) return rc;      // filled piece by piece
    } else if ((rc = dev_copy_new(ctx->pool, &d_planes, b->planes, b->n_plane_words, st))) return rc;
    if ((rc = dev_copy_new(ctx->pool, &d_poff, b->site_plane_off, ns, st))) return rc;
    if ((rc = dev_copy_new(ctx->pool, &d_nw, b->site_n_words, ns, st))) return rc;
    if ((rc = dev_copy_new(ctx->pool, &db->d.d_pos, b->site_pos, ns, st))) return rc;
    if ((rc = dev_copy_new(ctx->pool, &db->d.d_type, b->site_type, ns, st))) return rc;
    tr.mark("copies_enqueued");
    db->tri.assign(ns, 0);
    if (ns) {
        if ((rc = pool.alloc((void**)&db->d.d_tri, ns))) return rc;
        if (pipe) {
            // the caller's flags (checked against the planes once those are up: run_device_impl)
            for (uint64_t s = 0; s < ns; ++s) { if (b->site_tri[s] > 1) return fail(LGMI_E_ARG, "site_tri[%llu] = %u", (unsigned long long)s, b->site_tri[s]); db->tri[s] = b->site_tri[s]; }
            HIPCHK(hipMemcpyAsync(db->d.d_tri, db->tri.data(), ns, hipMemcpyHostToDevice, st));
        } else {
            HIPCHK(hipMemsetAsync(db->d.d_tri, 0, ns, st));
            launch_tri_flags(st, (uint32_t)ns, d_nw, d_poff, d_planes, db->d.d_tri);
            HIPCHK(hipMemcpyAsync(db->tri.data(), db->d.d_tri, ns, hipMemcpyDeviceToHost, st));
            HIPCHK(wait_stream(st));
            if (b->site_tri)
                for (uint64_t s = 0; s < ns; ++s)
                    if ((b->site_tri[s] != 0) != (db->tri[s] != 0))
                        return fail(LGMI_E_ARG, "site_tri[%llu] = %u, but the planes of the site %s reads of class 0",
                                    (unsigned long long)s, b->site_tri[s], db->tri[s] ? "hold" : "hold no");
        }
    }
    tr.mark("tri_flags");
    // column table: real sites, then one pseudo column per tri site.  On several threads over site ranges (a prefix over
    // the ranges gives each its offsets): 1.1 M sites of 20,000 footprints were 4 - 7 ms of a one-shot call on one thread
    db->pseudo_of_site.resize(ns);
    uint64_t off = 0;
    {
        Team team(plan_threads(ns / 64));
        const unsigned T = team.T;
        std::vector<uint64_t> w_real(T + 1, 0), w_tri(T + 1, 0), n_tri(T + 1, 0);
        team.run([&](unsigned t) {
            const uint64_t s0 = ns * t / T, s1 = ns * (t + 1) / T;
            uint64_t wr = 0, wt = 0, nt = 0;
            for (uint64_t s = s0; s < s1; ++s) { wr += b->site_n_words[s]; if (db->tri[s]) { wt += b->site_n_words[s]; ++nt; } }
            w_real[t + 1] = wr; w_tri[t + 1] = wt; n_tri[t + 1] = nt;
            team.barrier();
            if (t == 0) {
                for (unsigned k = 0; k < T; ++k) { w_real[k + 1] += w_real[k]; w_tri[k + 1] += w_tri[k]; n_tri[k + 1] += n_tri[k]; }
                db->pseudo_site.resize(n_tri[T]);
                db->cols.resize(ns + n_tri[T]);
            }
            team.barrier();
            uint64_t o = w_real[t], op = w_real[T] + w_tri[t], np = n_tri[t];
            for (uint64_t s = s0; s < s1; ++s) {
                db->cols[s] = Col{o, b->site_word_off[s], b->site_n_words[s]};
                o += b->site_n_words[s];
                uint32_t pc = NONE;
                if (db->tri[s]) {
                    pc = (uint32_t)(ns + np);
                    db->pseudo_site[np] = (uint32_t)s;
                    db->cols[ns + np] = Col{op, b->site_word_off[s], b->site_n_words[s]};
                    op += b->site_n_words[s];
                    ++np;
                }
                db->pseudo_of_site[s] = pc;
            }
        });
        off = w_real[T] + w_tri[T];
    }
    tr.mark("cols_built");
    if (db->cols.size() >= 0xFFFFFFF0ull) return fail(LGMI_E_ARG, "too many columns");
    db->d.n_cols = db->cols.size();
    db->d.n_pairs16 = off;
    if ((rc = dev_copy_new(ctx->pool, &db->d.d_cols, db->cols.data(), db->cols.size(), st))) return rc;
    if ((rc = dev_copy_new(ctx->pool, &d_pseudo, db->pseudo_site.data(), db->pseudo_site.size(), st))) return rc;
    tr.mark("cols_up");
    // one all-zero entry after the last column: k_count_mfma reads it for words outside a column's band
    if ((rc = pool.alloc((void**)&db->d.d_cplanes, (off + 1) * sizeof(ulonglong2)))) return rc;
    HIPCHK(hipMemsetAsync(db->d.d_cplanes + off, 0, sizeof(ulonglong2), st));
    tr.mark("cols");
    if (pipe) {
        // the planes are still on the host: cut the sites into pieces of equal plane words; what follows is queued by
        // run_device_impl, piece by piece, between its count launches
        uint32_t K = 8;
        if (const char* e = getenv("LGMI_UPLOAD_CHUNKS")) K = (uint32_t)std::min(64, std::max(1, atoi(e)));
        pipe->ctx = ctx; pipe->b = b; pipe->us = ctx_copy_stream(ctx); pipe->K = K;
        pipe->site_begin.assign(K + 1, ns); pipe->word_begin.assign(K + 1, b->n_plane_words); pipe->pseudo_begin.assign(K + 1, db->pseudo_site.size());
        uint64_t s = 0, np = 0;
        for (uint32_t k = 0; k < K; ++k) {
            const uint64_t target = b->n_plane_words / K * k;
            while (s < ns && b->site_plane_off[s] < target) { if (db->tri[s]) ++np; ++s; }
            pipe->site_begin[k] = s; pipe->word_begin[k] = s < ns ? b->site_plane_off[s] : b->n_plane_words; pipe->pseudo_begin[k] = np;
        }
        if ((rc = pool.alloc((void**)&pipe->d_tri_chk, std::max<uint64_t>(ns, 1)))) return rc;
        if ((rc = pool.alloc((void**)&pipe->d_bad, 8))) return rc;
        HIPCHK(hipMemsetAsync(pipe->d_tri_chk, 0, std::max<uint64_t>(ns, 1), st));
        HIPCHK(hipMemsetAsync(pipe->d_bad, 0, 8, st));
        pipe->d_planes = d_planes; pipe->d_poff = d_poff; pipe->d_nw = d_nw; pipe->d_pseudo = d_pseudo;
        tmp.keep = true;                                         // the pipe owns them now
        HIPCHK(wait_stream(st));                                 // (small arrays only: the upload stream may read them from here on)
    } else {
        launch_prep_cols(st, 0u, (uint32_t)db->d.n_cols, (uint32_t)ns, db->d.d_cols, d_pseudo, d_poff, d_planes, db->d.d_cplanes);
        HIPCHK(hipGetLastError());
        HIPCHK(wait_stream(st));
    }
    tr.mark("prep_done");
    guard.p = nullptr;
    *out = db;
    return LGMI_OK;
}
extern "C" int lgmi_batch_upload(lgmi_ctx* ctx, const lgmi_batch* b, lgmi_dbatch** out) { return upload_impl(ctx, b, out, nullptr); }

// one piece of a pipelined upload: its planes (copy stream), their layout prep (a stream of its own, so that the next
// piece's copy does not queue behind a prep kernel that waits for a compute unit the count kernels hold), an event the main
// stream waits for.  After the last piece the caller's tri flags are checked against the planes — behind the event: the
// count kernels do not wait for the check, the run's final read-back does (checked: event index 191).
static int pipe_chunk(UploadPipe& p, lgmi_dbatch* db, uint32_t k, hipEvent_t done) {
    const uint64_t w0 = p.word_begin[k], w1 = p.word_begin[k + 1], s0 = p.site_begin[k], s1 = p.site_begin[k + 1];
    const uint32_t ns = (uint32_t)db->d.n_sites;
    if (!p.ps) p.ps = ctx_prep_stream(p.ctx);
    hipStream_t ps = p.ps ? p.ps : p.us;
    if (w1 > w0) HIPCHK(hipMemcpyAsync(p.d_planes + w0, p.b->planes + w0, (w1 - w0) * 8, hipMemcpyHostToDevice, p.us));
    if (ps != p.us) {
        hipEvent_t copied = p.ctx->event(lgmi_ctx::EV_COPIED + (k & 31u));
        if (!copied) return fail(LGMI_E_HIP, "hipEventCreate failed");
        HIPCHK(hipEventRecord(copied, p.us));
        HIPCHK(hipStreamWaitEvent(ps, copied, 0));
    }
    launch_prep_cols(ps, (uint32_t)s0, (uint32_t)(s1 - s0), ns, db->d.d_cols, p.d_pseudo, p.d_poff, p.d_planes, db->d.d_cplanes);
    launch_prep_cols(ps, ns + (uint32_t)p.pseudo_begin[k], (uint32_t)(p.pseudo_begin[k + 1] - p.pseudo_begin[k]), ns, db->d.d_cols, p.d_pseudo,
                     p.d_poff, p.d_planes, db->d.d_cplanes);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(done, ps));
    if (k + 1 == p.K) {                              // all planes are up: the caller's tri flags against them
        launch_tri_flags(ps, ns, p.d_nw, p.d_poff, p.d_planes, p.d_tri_chk);
        launch_flags_differ(ps, ns, db->d.d_tri, p.d_tri_chk, p.d_bad);
        HIPCHK(hipGetLastError());
        hipEvent_t checked = p.ctx->event(lgmi_ctx::EV_CHECKED);
        if (!checked) return fail(LGMI_E_HIP, "hipEventCreate failed");
        HIPCHK(hipEventRecord(checked, ps));
    }
    return LGMI_OK;
}

// ---------------------------------------------------------------- synthetic dense chromosome
extern "C" int lgmi_synth_dense(lgmi_ctx* ctx, const lgmi_synth_spec* sp, lgmi_dbatch** out) {
    if (!ctx || !sp || !out) return fail(LGMI_E_ARG, "NULL argument");
    *out = nullptr;
    if (sp->n_sites == 0 || sp->n_reads == 0) return fail(LGMI_E_ARG, "n_sites and n_reads must be > 0");
    if (sp->dropout_u16 > 65536 || sp->het_noise_u16 > 65536 || sp->tri_frac_u16 > 65536)
        return fail(LGMI_E_ARG, "u16 probabilities must be <= 65536");
    if (sp->reserved) return fail(LGMI_E_ARG, "reserved must be 0");
    const uint32_t nb = sp->n_blocks ? sp->n_blocks : 1u;
    if ((uint64_t)nb * sp->n_sites >= 0x7FFFFFF0ull) return fail(LGMI_E_ARG, "too many sites");
    HIPCHK(hipSetDevice(ctx->device));
    lgmi_dbatch* db = new lgmi_dbatch();
    db->ctx = ctx;
    struct Guard { lgmi_dbatch* p; ~Guard() { if (p) { free_dbatch_device(p); delete p; } } } guard{db};
    const uint32_t P = sp->n_sites, ns = nb * P, W = (sp->n_reads + 63) / 64;
    db->d.n_blocks = nb;
    db->d.n_sites = ns;
    db->block_site_begin.resize(nb + 1);
    for (uint32_t c = 0; c <= nb; ++c) db->block_site_begin[c] = (uint64_t)c * P;
    db->block_n_reads.assign(nb, sp->n_reads);
    db->max_reads = sp->n_reads;
    db->pos.resize(ns); db->type.resize(ns); db->tri.resize(ns);
    db->pseudo_of_site.assign(ns, NONE);
    for (uint32_t c = 0; c < nb; ++c) {
        lgmi_synth_spec bs = *sp;
        bs.seed = sp->seed + c;                     // block c of a multi-chromosome batch == a one-block batch with seed + c
        for (uint32_t s = 0; s < P; ++s) {
            const SynthSite ss = synth_site(bs, s);
            const uint32_t g = c * P + s;
            db->pos[g] = 10000 + 37ll * s;
            db->type[g] = ss.het ? LGMI_TYPE_HET_SNP : (ss.snp ? LGMI_TYPE_SNP : LGMI_TYPE_MISMATCH);
            db->tri[g] = ss.tri ? 1 : 0;
        }
    }
    { auto t = std::make_shared<SiteTable>(); t->block_site_begin = db->block_site_begin; t->type = db->type; db->table = t; }
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
    if ((rc = dev_copy_new(ctx->pool, &db->d.d_pos, db->pos.data(), ns, st))) return rc;
    if ((rc = dev_copy_new(ctx->pool, &db->d.d_type, db->type.data(), ns, st))) return rc;
    if ((rc = dev_copy_new(ctx->pool, &db->d.d_tri, db->tri.data(), ns, st))) return rc;
    if ((rc = dev_copy_new(ctx->pool, &db->d.d_cols, db->cols.data(), db->cols.size(), st))) return rc;
    if ((rc = ctx->pool.alloc((void**)&db->d.d_cplanes, (db->d.n_pairs16 + 1) * sizeof(ulonglong2)))) return rc;
    HIPCHK(hipMemsetAsync(db->d.d_cplanes + db->d.n_pairs16, 0, sizeof(ulonglong2), st));
    uint32_t* d_depth = nullptr; uint32_t* d_pos_ = nullptr;
    struct Tmp { Pool& p; hipStream_t st; uint32_t** a; uint32_t** b;
                 ~Tmp() { (void)hipStreamSynchronize(st); p.release(*a); p.release(*b); } } tmp{ctx->pool, st, &d_depth, &d_pos_};
    if ((rc = ctx->pool.alloc((void**)&d_depth, (size_t)ns * 3 * sizeof(uint32_t)))) return rc;
    HIPCHK(hipMemsetAsync(d_depth, 0, (size_t)ns * 3 * sizeof(uint32_t), st));
    if ((rc = dev_copy_new(ctx->pool, &d_pos_, db->pseudo_of_site.data(), ns, st))) return rc;
    for (uint32_t c = 0; c < nb; ++c) {
        lgmi_synth_spec bs = *sp;
        bs.seed = sp->seed + c;
        launch_synth_depth(st, bs, W, d_depth + (size_t)3 * c * P);
        launch_synth_write(st, bs, W, d_depth + (size_t)3 * c * P, d_pos_ + (size_t)c * P, db->d.d_cplanes, c * P);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(wait_stream(st));
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
    out->site_tri = db->tri.data();
    return LGMI_OK;
}

// ---------------------------------------------------------------- planning (host, every run): plan.cpp
static void plan_env(int* count_kernel, uint32_t* xg) {
    *count_kernel = 0; *xg = 0;   // LGMI_COUNT_KERNEL=valu|mfma|mfma_i8 forces a count kernel (tests, A/B runs)
    if (const char* e = getenv("LGMI_COUNT_KERNEL")) {
        if (!strcmp(e, "valu")) *count_kernel = 1;
        else if (!strcmp(e, "mfma")) *count_kernel = 2;
        else if (!strcmp(e, "mfma_i8")) *count_kernel = 3;
    }
    if (const char* e = getenv("LGMI_XG")) *xg = (uint32_t)std::max(1, atoi(e));
}

static PlanInput plan_input(const lgmi_dbatch* db) {
    PlanInput in;
    in.n_blocks = db->d.n_blocks; in.n_sites = db->d.n_sites;
    in.block_site_begin = db->block_site_begin.data(); in.block_n_reads = db->block_n_reads.data();
    in.type = db->type.data(); in.tri = db->tri.data(); in.cols = db->cols.data();
    in.pseudo_of_site = db->pseudo_of_site.data();
    return in;
}

// the same plan without a GPU (include/lgmi.h: lgmi_plan_shard)
namespace {
struct ShardPlanOwner {
    std::vector<uint32_t> item_site, item_seg, tile_block, tile_x0, tile_y0, tile_edge, xrow, ycol, prow, pcol, xnext;
};
}

extern "C" void lgmi_shard_plan_free(lgmi_shard_plan* p) {
    if (!p) return;
    delete static_cast<ShardPlanOwner*>(p->owner_);
    memset(p, 0, sizeof *p);
}

extern "C" int lgmi_plan_shard(const lgmi_batch* b, int het_only, uint32_t n_shuffles, uint32_t shard_rank,
                               uint32_t shard_world, lgmi_shard_plan* out) {
    if (!out) return fail(LGMI_E_ARG, "out is NULL");
    memset(out, 0, sizeof *out);
    int rc = validate_batch(b);
    if (rc) return rc;
    if (shard_world == 0 || shard_rank >= shard_world) return fail(LGMI_E_ARG, "shard %u of %u", shard_rank, shard_world);
    const uint64_t ns = b->n_sites;
    // host restatement of what lgmi_batch_upload() derives on the device: tri flags and the column table
    std::vector<uint8_t> tri(ns, 0);
    std::vector<Col> cols(ns);
    std::vector<uint32_t> pseudo_of_site(ns, NONE);
    uint64_t off = 0;
    for (uint64_t s = 0; s < ns; ++s) {
        const uint64_t* lo = b->planes + b->site_plane_off[s];
        const uint64_t* hi = lo + b->site_n_words[s];
        uint64_t any0 = 0;
        for (uint32_t k = 0; k < b->site_n_words[s]; ++k) any0 |= lo[k] & hi[k];
        tri[s] = any0 ? 1 : 0;
        cols[s] = Col{off, b->site_word_off[s], b->site_n_words[s]};
        off += b->site_n_words[s];
    }
    for (uint64_t s = 0; s < ns; ++s)
        if (tri[s]) {
            pseudo_of_site[s] = (uint32_t)cols.size();
            cols.push_back(Col{off, b->site_word_off[s], b->site_n_words[s]});
            off += b->site_n_words[s];
        }
    std::vector<uint64_t> bsb(b->block_site_begin, b->block_site_begin + b->n_blocks + (b->block_site_begin ? 1 : 0));
    if (bsb.empty()) bsb.push_back(0);
    PlanInput in;
    in.n_blocks = b->n_blocks; in.n_sites = ns; in.block_site_begin = bsb.data(); in.block_n_reads = b->block_n_reads;
    in.type = b->site_type; in.tri = tri.data(); in.cols = cols.data(); in.pseudo_of_site = pseudo_of_site.data();
    int ck; uint32_t xg;
    plan_env(&ck, &xg);
    Plan whole, mine;
    build_plan(in, het_only != 0, 0, 1, ck, xg, n_shuffles, whole);
    build_plan(in, het_only != 0, shard_rank, shard_world, ck, xg, n_shuffles, mine);
    ShardPlanOwner* o = new ShardPlanOwner();
    for (const uint2& it : mine.items) { o->item_site.push_back(it.x); o->item_seg.push_back(it.y); }
    for (int kind = 0; kind < 2; ++kind)
        for (const Tile& t : (kind ? mine.mtiles : mine.tiles)) {
            o->tile_block.push_back(t.block); o->tile_x0.push_back(t.x0); o->tile_y0.push_back(t.y0);
            o->tile_edge.push_back(kind ? 128u : (uint32_t)TILE);
        }
    for (const SiteMap& m : mine.smap) {
        o->xrow.push_back(m.xrow); o->ycol.push_back(m.ycol); o->prow.push_back(m.prow); o->pcol.push_back(m.pcol);
        o->xnext.push_back(m.xnext);
    }
    out->n_items_total = mine.items.size();
    out->item_begin = mine.item_begin; out->item_end = mine.item_end;
    out->n_examined_total = mine.n_examined_total; out->n_examined = mine.n_examined;
    out->n_tiles_total = whole.tiles.size() + whole.mtiles.size();
    out->n_tiles = o->tile_block.size();
    out->item_site = o->item_site.data(); out->item_seg = o->item_seg.data();
    out->tile_block = o->tile_block.data(); out->tile_x0 = o->tile_x0.data(); out->tile_y0 = o->tile_y0.data();
    out->tile_edge = o->tile_edge.data();
    out->site_xrow = o->xrow.data(); out->site_ycol = o->ycol.data(); out->site_prow = o->prow.data();
    out->site_pcol = o->pcol.data(); out->site_xnext = o->xnext.data();
    out->owner_ = o;
    return LGMI_OK;
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
    p.release(r->d_exceed); p.release(r->d_counts); p.release(r->d_rec); p.release(r->d_mean); p.release(r->d_npairs);
    p.release(r->d_sum); p.release(r->d_nrows);
    p.release(r->d_nfirst); p.release(r->d_ncand); p.release(r->d_full); p.release(r->d_row_begin); p.release(r->d_jlisted);
    p.release(r->d_exceed16);
    delete r;
}

static const size_t PLAN_CACHE_N = 4;
// the batch's plan for these parameters: cached, or built now (host side only; run_device_impl uploads it on first use)
static PlanCache* plan_for(lgmi_ctx* ctx, const lgmi_dbatch* db, bool het_only, uint32_t sh_rank, uint32_t sh_world, uint32_t n_shuffles,
                           bool* fresh) {
    int ck; uint32_t xg;
    plan_env(&ck, &xg);
    if (sh_world == 0) { sh_world = 1; sh_rank = 0; }
    const uint32_t nkey = sh_world > 1 ? n_shuffles : 0u;        // the shuffle count only prices work items: it moves shard boundaries, nothing else
    const bool no_cache = getenv("LGMI_NO_PLAN_CACHE") != nullptr;
    if (!no_cache)
        for (auto& pc : db->plans)
            if (pc->het_only == het_only && pc->sh_rank == sh_rank && pc->sh_world == sh_world && pc->n_shuffles_key == nkey &&
                pc->ck == ck && pc->xg == xg) {
                pc->stamp = ++db->plan_stamp;
                if (fresh) *fresh = false;
                return pc.get();
            }
    if (db->plans.size() >= PLAN_CACHE_N || (no_cache && !db->plans.empty())) {        // drop the least recently used one
        size_t lru = 0;
        for (size_t k = 1; k < db->plans.size(); ++k) if (db->plans[k]->stamp < db->plans[lru]->stamp) lru = k;
        (void)hipStreamSynchronize(ctx->stream);                 // nothing in flight reads its device arrays
        db->plans[lru]->release(ctx->pool);
        db->plans.erase(db->plans.begin() + (long)lru);
    }
    std::unique_ptr<PlanCache> pc(new PlanCache());
    pc->het_only = het_only; pc->sh_rank = sh_rank; pc->sh_world = sh_world; pc->n_shuffles_key = nkey; pc->ck = ck; pc->xg = xg;
    pc->stamp = ++db->plan_stamp;
    const auto t0 = std::chrono::steady_clock::now();
    build_plan(plan_input(db), het_only, sh_rank, sh_world, ck, xg, n_shuffles, pc->pl);
    pc->ms_build = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (fresh) *fresh = true;
    db->plans.push_back(std::move(pc));
    return db->plans.back().get();
}

// A pipelined upload delivers the sites in order: a count tile (an operand group) can run once the LAST site among its
// rows and columns is up.  The plan's tiles and operand groups are put in the order of the piece that completes them
// (stable: inside a piece the planner's L2-friendly order stays) and the pieces' ranges noted.
static void pipe_order_plan(const lgmi_dbatch* db, Plan& pl, UploadPipe& p) {
    const uint32_t ns = (uint32_t)db->d.n_sites, K = p.K;
    auto chunk_of_site = [&](uint32_t s) { return (uint32_t)(std::upper_bound(p.site_begin.begin() + 1, p.site_begin.begin() + K, (uint64_t)s) - (p.site_begin.begin() + 1)); };
    auto chunk_of_col = [&](uint32_t c) { return chunk_of_site(c < ns ? c : db->pseudo_site[c - ns]); };
    // per 32-entry group of every block's y list and x list: the latest piece among its columns / rows (either list is
    // made of runs in site order — other sites, x sites, pseudo columns; x sites with their pseudo rows next to them or
    // behind them all — so a group may straddle two runs)
    std::vector<uint32_t> ygrp_begin(pl.plans.size() + 1, 0), xgrp_begin(pl.plans.size() + 1, 0);
    for (size_t b = 0; b < pl.plans.size(); ++b) {
        ygrp_begin[b + 1] = ygrp_begin[b] + (pl.plans[b].ny + 31u) / 32u;
        xgrp_begin[b + 1] = xgrp_begin[b] + (pl.plans[b].nx + 31u) / 32u;
    }
    std::vector<uint8_t> ygrp(ygrp_begin.back(), 0), xgrp(xgrp_begin.back(), 0);
    for (size_t b = 0; b < pl.plans.size(); ++b) {
        const BlockPlan& bp = pl.plans[b];
        const uint32_t* yl = pl.ylist.data() + bp.yl_off;
        const uint32_t* xl = pl.xlist.data() + bp.xl_off;
        for (uint32_t q = 0; q < bp.ny; ++q) { uint8_t& g = ygrp[ygrp_begin[b] + q / 32u]; g = std::max<uint8_t>(g, (uint8_t)chunk_of_col(yl[q])); }
        for (uint32_t r = 0; r < bp.nx; ++r) { uint8_t& g = xgrp[xgrp_begin[b] + r / 32u]; g = std::max<uint8_t>(g, (uint8_t)chunk_of_col(xl[r])); }
    }
    auto order = [&](std::vector<Tile>& tiles, uint32_t edge, std::vector<uint32_t>& begin) {
        std::vector<uint8_t> ck(tiles.size());
        begin.assign(K + 1, 0);
        for (size_t k = 0; k < tiles.size(); ++k) {
            const Tile& t = tiles[k];
            const BlockPlan& bp = pl.plans[t.block];
            uint32_t c = 0;
            for (uint32_t g = t.x0 / 32u; g < std::min((bp.nx + 31u) / 32u, (t.x0 + edge) / 32u); ++g) c = std::max<uint32_t>(c, xgrp[xgrp_begin[t.block] + g]);
            for (uint32_t g = t.y0 / 32u; g < std::min((bp.ny + 31u) / 32u, (t.y0 + edge) / 32u); ++g) c = std::max<uint32_t>(c, ygrp[ygrp_begin[t.block] + g]);
            ck[k] = (uint8_t)c;
            ++begin[c + 1];
        }
        for (uint32_t k = 0; k < K; ++k) begin[k + 1] += begin[k];
        std::vector<Tile> sorted(tiles.size());
        std::vector<uint32_t> at(begin.begin(), begin.end() - 1);
        for (size_t k = 0; k < tiles.size(); ++k) sorted[at[ck[k]]++] = tiles[k];
        tiles.swap(sorted);
    };
    order(pl.mtiles, 128u, p.mt_begin);
    order(pl.tiles, (uint32_t)TILE, p.t_begin);
    {
        std::vector<uint8_t> ck(pl.op_groups.size());
        p.og_begin.assign(K + 1, 0);
        for (size_t k = 0; k < pl.op_groups.size(); ++k) {
            const OpGroup& g = pl.op_groups[k];
            const BlockPlan& bp = pl.plans[g.block];
            (void)bp;
            ck[k] = g.is_y ? ygrp[ygrp_begin[g.block] + g.group] : xgrp[xgrp_begin[g.block] + g.group];
            ++p.og_begin[ck[k] + 1u];
        }
        for (uint32_t k = 0; k < K; ++k) p.og_begin[k + 1] += p.og_begin[k];
        std::vector<OpGroup> sorted(pl.op_groups.size());
        std::vector<uint32_t> at(p.og_begin.begin(), p.og_begin.end() - 1);
        for (size_t k = 0; k < pl.op_groups.size(); ++k) sorted[at[ck[k]]++] = pl.op_groups[k];
        pl.op_groups.swap(sorted);
    }
}

static int run_device_impl(lgmi_ctx* ctx, const lgmi_dbatch* db, const lgmi_params* prm, lgmi_dresult** out, bool defer_perm,
                           uint64_t cap_hint = 0, UploadPipe* pipe = nullptr) {
    if (!ctx || !db || !prm || !out) return fail(LGMI_E_ARG, "NULL argument");
    *out = nullptr;
    HostTrace tr;
    if (db->ctx != ctx) return fail(LGMI_E_ARG, "batch belongs to another context");
    if (prm->no_row_p > 1) return fail(LGMI_E_ARG, "lgmi_params.no_row_p must be 0 or 1");
    if (prm->reserved1[0] || prm->reserved1[1] || prm->reserved1[2]) return fail(LGMI_E_ARG, "reserved params bytes must be 0");
    if (prm->compact_rows > 1) return fail(LGMI_E_ARG, "lgmi_params.compact_rows must be 0 or 1");
    if ((uint64_t)prm->stream_site_base + db->d.n_sites >= 0xFFFFFFF0ull) return fail(LGMI_E_ARG, "stream_site_base + sites overflows 32 bits");
    if (prm->exact_2x2 > 1) return fail(LGMI_E_ARG, "exact_2x2 must be 0 or 1");
    if (prm->n_shuffles > (1u << 24)) return fail(LGMI_E_ARG, "n_shuffles must be <= 2^24");
    uint32_t sh_world = prm->shard_world, sh_rank = prm->shard_rank;
    if (sh_world == 0) { if (sh_rank) return fail(LGMI_E_ARG, "shard_rank %u with shard_world 0", sh_rank); sh_world = 1; }
    if (sh_rank >= sh_world) return fail(LGMI_E_ARG, "shard_rank %u >= shard_world %u", sh_rank, sh_world);
    HIPCHK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    Pool& pool = ctx->pool;
    const uint32_t ns = (uint32_t)db->d.n_sites;
    const bool want_p = prm->n_shuffles > 0 || prm->exact_2x2;
    const bool want_counts = prm->emit_counts != 0;
    int rc;

    // the log-factorial table also serves the binomial draw of the 2 x 2 path: LF[0 .. n_shuffles]
    if (want_p && db->max_reads >= (1u << 28))
        return fail(LGMI_E_ARG, "the permutation test takes blocks of fewer than 2^28 reads (%u)", db->max_reads);
    if (want_p && (rc = ensure_perm_tables(ctx, std::max(db->max_reads, prm->n_shuffles)))) return rc;   // first use only
    HIPCHK(hipEventRecord(ctx->ev[0], st));
    tr.mark("ev0");
    bool plan_fresh = false;
    PlanCache* pcache = plan_for(ctx, db, prm->het_only != 0, sh_rank, sh_world, prm->n_shuffles, &plan_fresh);
    Plan& pl = pcache->pl;
    if (pipe) {
        // (the tiles are reordered in place: the plan must not be on the device yet — it is the new batch's first)
        if (pcache->on_device || sh_world > 1) return fail(LGMI_E_STATE, "internal: a pipelined upload needs the batch's first, unsharded plan");
        pipe_order_plan(db, pl, *pipe);
    }
    const float ms_plan_host = plan_fresh ? pcache->ms_build : 0.f;      // 0: the batch's cached plan
    tr.mark("planned");
    const size_t n_items = (size_t)(pl.item_end - pl.item_begin);
    uint32_t first_site = NONE, first_seg = 0;
    if (n_items) { first_site = pl.items[pl.item_begin].x; first_seg = pl.items[pl.item_begin].y; }
    if (pl.tiles.size() >= 0x7FFFFFFFull || pl.mtiles.size() >= 0x7FFFFFFFull || n_items >= 0x7FFFFFFFull)
        return fail(LGMI_E_ARG, "too many tiles");
    // the permutation kernels carry row numbers in 32 bits (gen_list)
    if (want_p && pl.n_examined >= 0xFFFFFFFFull)
        return fail(LGMI_E_ARG, "%llu candidate rows: permutation p-values need fewer than 2^32 rows per run (shard the batch)",
                    (unsigned long long)pl.n_examined);

    lgmi_dresult* res = new lgmi_dresult();
    res->ctx = ctx;
    res->n_sites = ns;
    res->has_p = want_p;
    res->has_counts = want_counts;
    res->sharded = sh_world > 1;
    res->n_shuffles = prm->n_shuffles;
    res->p_from_exceed = prm->n_shuffles > 0 && !prm->exact_2x2;
    res->table = db->table; res->het_only = prm->het_only != 0;
    res->first_site = first_site; res->first_seg = first_seg;
    const bool keep_p = want_p && !(res->p_from_exceed && prm->no_row_p);   // row_p as an array of its own
    // scratch (returned to the pool at the end of the call) and the result
    std::vector<void*> scratch;
    struct Guard {
        Pool& p; std::vector<void*>& s; lgmi_dresult* r;
        PlanCache* transient_ops = nullptr; hipStream_t st = nullptr;      // a sequential shard's operand layout is used once: not kept
        ~Guard() {
            for (void* q : s) p.release(q);
            if (transient_ops && transient_ops->d_ops) {
                (void)hipStreamSynchronize(st);
                p.release(transient_ops->d_ops); transient_ops->d_ops = nullptr; transient_ops->ops_ready = false;
            }
            if (r) lgmi_dresult_free(r);
        }
    } guard{pool, scratch, res};
    auto salloc = [&](void** p, size_t bytes) { int e = pool.alloc(p, bytes); if (!e) scratch.push_back(*p); return e; };

    uint4* d_slots; uint32_t* d_rowcnt; uint64_t* d_rowstart; uint32_t* d_cnt;
    int* d_err; unsigned long long* d_wordpairs;
    if (!pcache->on_device) {
        Pool& pp = pool;
        int e = 0;
        pcache->release(pool);       // (a run that failed between these allocations and the upload left pointers behind: advice r3)
        if (!e) e = pp.alloc((void**)&pcache->d_plans, std::max<size_t>(pl.plans.size(), 1) * sizeof(BlockPlan));
        if (!e) e = pp.alloc((void**)&pcache->d_xlist, std::max<size_t>(pl.xlist.size(), 1) * 4);
        if (!e) e = pp.alloc((void**)&pcache->d_items, std::max<size_t>(n_items, 1) * sizeof(uint2));
        if (!e) e = pp.alloc((void**)&pcache->d_units, std::max<size_t>(pl.units.size(), 1) * sizeof(uint2));
        if (!e) e = pp.alloc((void**)&pcache->d_ylist, std::max<size_t>(pl.ylist.size(), 1) * 4);
        if (!e) e = pp.alloc((void**)&pcache->d_smap, std::max<size_t>(pl.smap.size(), 1) * sizeof(SiteMap));
        if (!e) e = pp.alloc((void**)&pcache->d_tiles, std::max<size_t>(pl.tiles.size(), 1) * sizeof(Tile));
        if (!e) e = pp.alloc((void**)&pcache->d_mtiles, std::max<size_t>(pl.mtiles.size(), 1) * sizeof(Tile));
        if (!e) e = pp.alloc((void**)&pcache->d_opgroups, std::max<size_t>(pl.op_groups.size(), 1) * sizeof(OpGroup));
        if (!e) e = pp.alloc((void**)&pcache->d_xrows, std::max<size_t>(pl.xrows.size(), 1) * 4);
        if (e) { pcache->release(pool); return e; }
    }
    BlockPlan* const d_plans = pcache->d_plans; uint32_t* const d_xlist = pcache->d_xlist; uint32_t* const d_ylist = pcache->d_ylist;
    SiteMap* const d_smap = pcache->d_smap; Tile* const d_tiles = pcache->d_tiles; Tile* const d_mtiles = pcache->d_mtiles;
    uint2* const d_items = pcache->d_items; uint2* const d_units = pcache->d_units;
    if ((rc = salloc((void**)&d_slots, pl.total_slots * sizeof(uint4)))) return rc;
    OpGroup* const d_opgroups = pcache->d_opgroups;
    static const bool keep_ops = getenv("LGMI_NO_OPS_CACHE") == nullptr;
    if (!pcache->d_ops) {
        pcache->ops_ready = false;
        if ((rc = pool.alloc((void**)&pcache->d_ops, std::max<uint64_t>(pl.op_total, 1) * sizeof(uint4)))) return rc;
    }
    uint4* const d_ops = pcache->d_ops;
    if (cap_hint) { guard.transient_ops = pcache; guard.st = st; }     // (cap_hint: one of a run's sequential shards, api.cpp: run_device_split)
    const bool lay_ops = !(keep_ops && pcache->ops_ready);          // (LGMI_NO_OPS_CACHE: re-lay them every run, as up to round 4)
    if ((rc = salloc((void**)&d_rowcnt, std::max<size_t>(n_items, 1) * 4))) return rc;
    if ((rc = salloc((void**)&d_rowstart, (n_items + 1) * 8))) return rc;
    if ((rc = pool.alloc((void**)&res->d_sum, std::max<size_t>(ns, 1) * 8))) return rc;
    if ((rc = salloc((void**)&d_cnt, (size_t)ns * 4 + 64))) return rc;
    d_err = (int*)(d_cnt + ns);
    unsigned int* d_gencount = (unsigned int*)(d_cnt + ns) + 1;    // [0] queued rows, [1] k_perm_general's next row, [2] rows k_perm_enum leaves to it, [3] some row is enumerable (zeroed with d_cnt)
    d_wordpairs = nullptr;
    if ((rc = salloc((void**)&d_wordpairs, 8))) return rc;
    if ((rc = pool.alloc((void**)&res->d_mean, (size_t)ns * 8))) return rc;
    if ((rc = pool.alloc((void**)&res->d_npairs, (size_t)ns * 4))) return rc;
    if ((rc = pool.alloc((void**)&res->d_nfirst, std::max<size_t>(ns, 1) * 4))) return rc;
    if ((rc = pool.alloc((void**)&res->d_ncand, std::max<size_t>(ns, 1) * 4))) return rc;
    unsigned long long* d_sum = res->d_sum;

    // The plan goes up through one pinned staging buffer (cached by the context).  Straight from the pageable
    // std::vectors the nine copies blocked the host for 25 - 35 ms in about half of the processes on the same box
    // (LGMI_TRACE_HOST: "enqueued" 33 ms against 0.3 ms; the GPU idle meanwhile): what the runtime does with a
    // pageable source (stage or pin on the fly) is its own choice.
    struct Up { void* d; const void* h; size_t n; };
    const Up ups[] = {
        {d_plans, pl.plans.data(), pl.plans.size() * sizeof(BlockPlan)},
        {d_xlist, pl.xlist.data(), pl.xlist.size() * 4},
        {d_items, pl.items.data() + pl.item_begin, n_items * sizeof(uint2)},
        {d_units, pl.units.data(), pl.units.size() * sizeof(uint2)},
        {d_ylist, pl.ylist.data(), pl.ylist.size() * 4},
        {d_smap, pl.smap.data(), pl.smap.size() * sizeof(SiteMap)},
        {d_tiles, pl.tiles.data(), pl.tiles.size() * sizeof(Tile)},
        {d_mtiles, pl.mtiles.data(), pl.mtiles.size() * sizeof(Tile)},
        {d_opgroups, pl.op_groups.data(), pl.op_groups.size() * sizeof(OpGroup)},
        {pcache->d_xrows, pl.xrows.data(), pl.xrows.size() * 4},
    };
    size_t up_total = 0;
    if (!pcache->on_device) for (const Up& u : ups) up_total += (u.n + 255) & ~size_t(255);
    struct Staging {                       // back to the pool once the stream no longer reads it
        std::shared_ptr<PinnedPool> pool; hipStream_t st; void* p = nullptr; size_t got = 0;
        ~Staging() { if (p) { (void)hipStreamSynchronize(st); pool->give(p, got); } }
    } stage{ctx->pinned, st};
    if (up_total) {
        stage.p = ctx->pinned->take(up_total, &stage.got);
        if (!stage.p) return fail(LGMI_E_OOM, "pinned staging buffer of %zu bytes for the plan", up_total);
        size_t off = 0;
        for (const Up& u : ups) {
            if (!u.n) continue;
            memcpy((char*)stage.p + off, u.h, u.n);
            HIPCHK(hipMemcpyAsync(u.d, (char*)stage.p + off, u.n, hipMemcpyHostToDevice, st));
            off += (u.n + 255) & ~size_t(255);
        }
    }
    pcache->on_device = true;
    tr.mark("uploaded");
    if (ns) HIPCHK(hipMemsetAsync(d_sum, 0, (size_t)ns * 8, st));
    if (ns) HIPCHK(hipMemsetAsync(res->d_nfirst, 0, (size_t)ns * 4, st));
    HIPCHK(hipMemsetAsync(d_cnt, 0, (size_t)ns * 4 + 64, st));
    HIPCHK(hipMemsetAsync(d_wordpairs, 0, 8, st));

    // FP4 matrix-core blocks: operands re-laid in wave-load order (part of "prep": ~2 ms at north-star)
    if (!pipe) {
        if (pl.mfma_fp4 && !pl.mtiles.empty() && lay_ops) {
            launch_gather_ops(st, (uint32_t)pl.op_groups.size(), pl.op_max_steps, d_opgroups, d_plans, d_xlist, d_ylist,
                              db->d.d_cols, db->d.d_cplanes, d_ops);
            pcache->ops_ready = true;                            // (queued on the batch's one stream: every later run is behind it)
        }
        HIPCHK(hipEventRecord(ctx->ev[1], st));
        if (pl.mfma_fp4)
            launch_count_mfma_fp4(st, (uint32_t)pl.mtiles.size(), d_mtiles, d_plans, d_ops, d_slots);
        else
            launch_count_mfma(st, (uint32_t)pl.mtiles.size(), d_mtiles, d_plans, d_xlist, d_ylist, db->d.d_cols, db->d.d_cplanes,
                              db->d.d_cplanes + db->d.n_pairs16, d_slots);
        launch_count(st, (uint32_t)pl.tiles.size(), d_tiles, d_plans, d_xlist, d_ylist, db->d.d_cols, db->d.d_cplanes,
                     d_slots);
    } else {
        // the pipelined upload (lgmi_run): piece k of the planes goes up on the upload stream while the tiles that pieces
        // 0 .. k - 1 completed are counted here; the count stage's time (ms_count) includes what it waited for the planes
        HIPCHK(hipEventRecord(ctx->ev[1], st));
        for (uint32_t k = 0; k < pipe->K; ++k) {
            hipEvent_t up = ctx->event(lgmi_ctx::EV_UP + k);
            if (!up) return fail(LGMI_E_HIP, "hipEventCreate failed");
            if ((rc = pipe_chunk(*pipe, const_cast<lgmi_dbatch*>(db), k, up))) return rc;
            HIPCHK(hipStreamWaitEvent(st, up, 0));
            const uint32_t og0 = pipe->og_begin[k], og1 = pipe->og_begin[k + 1], mt0 = pipe->mt_begin[k], mt1 = pipe->mt_begin[k + 1],
                           t0 = pipe->t_begin[k], t1 = pipe->t_begin[k + 1];
            if (pl.mfma_fp4 && og1 > og0)                      // (a pipelined run is the batch's first: nothing laid out yet)
                launch_gather_ops(st, og1 - og0, pl.op_max_steps, d_opgroups + og0, d_plans, d_xlist, d_ylist, db->d.d_cols, db->d.d_cplanes, d_ops);
            if (pl.mfma_fp4) launch_count_mfma_fp4(st, mt1 - mt0, d_mtiles + mt0, d_plans, d_ops, d_slots);
            else launch_count_mfma(st, mt1 - mt0, d_mtiles + mt0, d_plans, d_xlist, d_ylist, db->d.d_cols, db->d.d_cplanes,
                                   db->d.d_cplanes + db->d.n_pairs16, d_slots);
            launch_count(st, t1 - t0, d_tiles + t0, d_plans, d_xlist, d_ylist, db->d.d_cols, db->d.d_cplanes, d_slots);
            HIPCHK(hipGetLastError());
            if (tr.on && ctx->event(lgmi_ctx::EV_UPDBG + k)) HIPCHK(hipEventRecord(ctx->event(lgmi_ctx::EV_UPDBG + k), st));
            tr.mark("piece");
        }
        if (pl.mfma_fp4 && !pl.mtiles.empty()) pcache->ops_ready = true;
        if (tr.on) {                                         // when each piece's planes were up / its tiles counted, on the device's clock
            HIPCHK(hipStreamSynchronize(st));
            for (uint32_t k = 0; k < pipe->K; ++k) {
                float up = 0.f, cnt = 0.f;
                (void)hipEventElapsedTime(&up, ctx->ev[1], ctx->event(lgmi_ctx::EV_UP + k));
                (void)hipEventElapsedTime(&cnt, ctx->ev[1], ctx->event(lgmi_ctx::EV_UPDBG + k));
                fprintf(stderr, "[lgmi pipe] piece %u: planes up at %.2f ms, its %u tiles counted at %.2f ms\n", k, up, pipe->mt_begin[k + 1] - pipe->mt_begin[k] + pipe->t_begin[k + 1] - pipe->t_begin[k], cnt);
            }
        }
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(ctx->ev[2], st));

    EmitArgs ea{};
    ea.n_sites = ns; ea.min_common = prm->min_common; ea.het_only = prm->het_only != 0;
    ea.plans = d_plans; ea.smap = d_smap; ea.xsites = d_ylist; ea.xrows = pcache->d_xrows; ea.rows_are_ranks = pl.rows_are_ranks ? 1 : 0; ea.cols = db->d.d_cols;
    ea.type = db->d.d_type; ea.tri = db->d.d_tri;
    ea.slots = d_slots;
    ea.n_items = (uint32_t)n_items; ea.items = d_items;
    ea.n_units = (uint32_t)pl.units.size(); ea.units = d_units;
    ea.row_cnt = d_rowcnt; ea.row_start = d_rowstart;
    unsigned long long* d_unitwords = nullptr; uint64_t* d_scantmp = nullptr;
    if ((rc = salloc((void**)&d_unitwords, std::max<size_t>(pl.units.size(), 1) * 8))) return rc;
    if ((rc = salloc((void**)&d_scantmp, scan_tmp_words((uint32_t)n_items) * 8))) return rc;
    ea.site_sum = d_sum; ea.site_cnt = d_cnt; ea.err_flag = d_err; ea.unit_words = d_unitwords;
    launch_emit_count(st, ea);
    launch_scan(st, d_rowcnt, d_rowstart, (uint32_t)n_items, d_scantmp);
    launch_site_rows(st, (uint32_t)n_items, d_items, d_rowcnt, ns, d_smap, d_plans, res->d_nfirst, res->d_ncand);
    launch_sum_u64(st, d_unitwords, (uint32_t)pl.units.size(), d_wordpairs);
    HIPCHK(hipGetLastError());
    // Row arrays: the number of rows is only known on the device here.  When the upper bound (every examined pair
    // is emitted — what a dense block does) fits a third of the device memory the arrays are sized by it and the
    // stream runs on without the host; otherwise one 8-byte read-back sizes them exactly.
    const size_t row_bytes = 16 + ((want_counts || want_p) ? 36 : 0) + (want_p ? 20 : 0) + (keep_p ? 8 : 0);
    uint64_t cap_rows = pl.n_examined;
    const bool by_bound = (double)cap_rows * (double)row_bytes <= (double)ctx->mem_total / 3.0 && !getenv("LGMI_EXACT_ROW_ALLOC");
    uint64_t n_rows = 0;
    if (!by_bound) {
        HIPCHK(hipMemcpyAsync(&n_rows, d_rowstart + n_items, 8, hipMemcpyDeviceToHost, st));
        HIPCHK(wait_stream(st));
        cap_rows = n_rows;
    }
    // (a sequence of shards: every shard's arrays as large as the largest shard's, so that the pool's blocks of the first
    //  serve all the others — shards are balanced by cost and their row counts grow along the batch)
    const size_t nr = (size_t)std::max<uint64_t>(std::max(cap_rows, by_bound ? cap_hint : 0), 1);
    if ((rc = pool.alloc((void**)&res->d_i, nr * 4))) return rc;
    if ((rc = pool.alloc((void**)&res->d_j, nr * 4))) return rc;
    if ((rc = pool.alloc((void**)&res->d_mi, nr * 8))) return rc;
    if (want_counts || want_p) if ((rc = pool.alloc((void**)&res->d_counts, nr * 36))) return rc;
    if (want_p) {
        if (keep_p && (rc = pool.alloc((void**)&res->d_p, nr * 8))) return rc;
        if ((rc = pool.alloc((void**)&res->d_exceed, nr * 4))) return rc;
        if ((rc = pool.alloc((void**)&res->d_rec, nr * 16))) return rc;
    }
    ea.out_i = res->d_i; ea.out_j = res->d_j; ea.out_mi = res->d_mi; ea.out_counts = res->d_counts;
    ea.out_rec = res->d_rec; ea.counts_sparse = (want_p && !want_counts) ? 1 : 0;
    launch_emit_write(st, ea);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(ctx->ev[3], st));
    HIPCHK(hipEventRecord(ctx->ev[6], st));
    bool exact_timed = false;
    if (defer_perm && want_p) {
        // split run: the rows are final here; the permutation stage runs later on this result (lgmi_dresult_permute)
        if ((rc = pool.alloc((void**)&res->d_nrows, 8))) return rc;
        HIPCHK(hipMemcpyAsync(res->d_nrows, d_rowstart + n_items, 8, hipMemcpyDeviceToDevice, st));
        res->perm_pending = true;
        res->cap_rows = cap_rows;
        res->prm = *prm;
    } else if (want_p && cap_rows) {
        uint32_t* d_genlist;
        if ((rc = salloc((void**)&d_genlist, (size_t)cap_rows * 4))) return rc;
        PermArgs pa{};
        pa.n_rows_dev = d_rowstart + n_items; pa.max_rows = cap_rows;
        pa.row_i = res->d_i; pa.row_j = res->d_j; pa.counts = res->d_counts; pa.rec = res->d_rec; pa.G = ctx->d_G; pa.LF = ctx->d_LF;
        pa.n_shuffles = prm->n_shuffles; pa.seed = prm->seed; pa.exact_2x2 = prm->exact_2x2; pa.site_base = prm->stream_site_base;
        pa.out_p = res->d_p; pa.out_exceed = res->d_exceed; pa.gen_list = d_genlist; pa.gen_count = d_gencount;
        launch_perm_fast(st, pa);
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(ctx->ev[6], st));
        launch_perm_general(st, pa, ctx->ev[7]);
        HIPCHK(hipGetLastError());
        exact_timed = true;
    }
    HIPCHK(hipEventRecord(ctx->ev[4], st));
    launch_site_mean(st, ns, d_sum, d_cnt, res->d_mean);
    if (ns) HIPCHK(hipMemcpyAsync(res->d_npairs, d_cnt, (size_t)ns * 4, hipMemcpyDeviceToDevice, st));
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(ctx->ev[5], st));
    unsigned long long* hs = ctx->h_scal;
    hs[0] = hs[1] = hs[2] = hs[3] = hs[4] = 0;
    HIPCHK(hipMemcpyAsync(&hs[0], d_err, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(&hs[1], d_gencount, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(&hs[4], d_gencount + 6, 4, hipMemcpyDeviceToHost, st));    // rows k_perm_six finished
    HIPCHK(hipMemcpyAsync(&hs[2], d_wordpairs, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(&hs[3], d_rowstart + n_items, 8, hipMemcpyDeviceToHost, st));
    hs[7] = 0;
    if (pipe) {                                              // the tri-flag check ran behind the last piece's event: wait for it here
        HIPCHK(hipStreamWaitEvent(st, ctx->event(lgmi_ctx::EV_CHECKED), 0));
        HIPCHK(hipMemcpyAsync(&hs[7], pipe->d_bad, 4, hipMemcpyDeviceToHost, st));
    }
    tr.mark("enqueued");
    HIPCHK(wait_stream(st));
    tr.mark("drained");
    if (pipe && (uint32_t)hs[7]) return fail(LGMI_E_ARG, "lgmi_batch.site_tri disagrees with the planes: some site's flag says the opposite of its lo & hi bits");
    const int err = (int)(uint32_t)hs[0]; const unsigned int n_general = (unsigned int)hs[1];
    const unsigned long long wp = hs[2];
    n_rows = hs[3];
    if (err) return fail(LGMI_E_DOMAIN, "math domain error: a pair with 0 common reads reached the MI (min_common == 0)");
    if (n_rows > cap_rows) return fail(LGMI_E_STATE, "internal: %llu rows exceed the planned bound %llu",
                                       (unsigned long long)n_rows, (unsigned long long)cap_rows);
    res->n_rows = n_rows;

    lgmi_run_info& inf = res->info;
    inf.n_rows = n_rows;
    inf.n_examined = pl.n_examined;
    inf.n_examined_total = pl.n_examined_total;
    inf.n_general_rows = n_general;
    inf.n_six_rows = (uint32_t)hs[4];
    inf.n_tile_pairs = (uint64_t)pl.tiles.size() * TILE * TILE + (uint64_t)pl.mtiles.size() * 128 * 128;
    inf.word_pairs = wp;
    inf.bytes_in = pl.bytes_in;
    inf.bytes_out = n_rows * (16ull + (want_p ? 4ull : 0ull) + (keep_p ? 8ull : 0ull) + (want_counts ? 36ull : 0ull));
    inf.n_count_launches = (pl.tiles.empty() ? 0 : 1) + (pl.mtiles.empty() ? 0 : 1);
    inf.n_mfma_tiles = (uint32_t)std::min<size_t>(pl.mtiles.size(), 0xFFFFFFFFu);
    inf.mfma_dtype = pl.mtiles.empty() ? 0u : (pl.mfma_fp4 ? 2u : 1u);
    inf.n_seq_shards = 1;
    inf.ms_plan_host = ms_plan_host;
    HIPCHK(hipEventElapsedTime(&inf.ms_prep, ctx->ev[0], ctx->ev[1]));
    HIPCHK(hipEventElapsedTime(&inf.ms_count, ctx->ev[1], ctx->ev[2]));
    HIPCHK(hipEventElapsedTime(&inf.ms_emit, ctx->ev[2], ctx->ev[3]));
    HIPCHK(hipEventElapsedTime(&inf.ms_perm, ctx->ev[3], ctx->ev[4]));
    HIPCHK(hipEventElapsedTime(&inf.ms_perm_fast, ctx->ev[3], ctx->ev[6]));
    HIPCHK(hipEventElapsedTime(&inf.ms_perm_general, ctx->ev[6], ctx->ev[4]));
    inf.ms_perm_exact = 0.f;
    if (exact_timed) HIPCHK(hipEventElapsedTime(&inf.ms_perm_exact, ctx->ev[6], ctx->ev[7]));
    HIPCHK(hipEventElapsedTime(&inf.ms_mean, ctx->ev[4], ctx->ev[5]));
    HIPCHK(hipEventElapsedTime(&inf.ms_total, ctx->ev[0], ctx->ev[5]));
    if (!want_counts && res->d_counts && !res->perm_pending) { pool.release(res->d_counts); res->d_counts = nullptr; }
    if (res->d_rec && !res->perm_pending) { pool.release(res->d_rec); res->d_rec = nullptr; }
    guard.r = nullptr;
    *out = res;
    return LGMI_OK;
}

// ---- one GPU, one call, more work than one launch sequence can hold: the run is cut into sequential shards with the
// planner the multi-GPU path uses (csrc/plan.cpp: any contiguous range of the ordered work items is a shard whose rows
// are that range of the unsharded rows, and the per-site sums are integers).  Each shard's slots, operands and
// bound-sized row arrays live only while it runs; its rows are appended to the final arrays, which hold what the
// caller asked for and nothing else.  When: the working set of the single sequence — slot matrix + operands + row arrays
// sized by the examined-pair bound — exceeds the budget (LGMI_MEM_BUDGET_MB, default 70 % of the device memory), or
// p-values are wanted for 2^32 candidate rows or more (the permutation kernels carry row numbers in 32 bits).
// The reference's analogue: the chunked Pool.map over footprints (src/giremi/script/giremi.py:367-394).
static size_t row_bytes_working(bool want_p, bool keep_p, bool want_counts) {
    return 16 + ((want_counts || want_p) ? 36 : 0) + (want_p ? 20 : 0) + (keep_p ? 8 : 0);
}
static size_t row_bytes_final(bool want_p, bool keep_p, bool want_counts) {
    return 16 + (want_p ? 4 : 0) + (keep_p ? 8 : 0) + (want_counts ? 36 : 0);
}

static int split_count(lgmi_ctx* ctx, const lgmi_dbatch* db, const lgmi_params* prm, const Plan& pl, uint64_t* largest = nullptr,
                       float* ms_host = nullptr) {
    const auto t_enter = std::chrono::steady_clock::now();
    const bool want_p = prm->n_shuffles > 0 || prm->exact_2x2;
    const bool keep_p = want_p && !(prm->n_shuffles > 0 && !prm->exact_2x2 && prm->no_row_p);
    const bool want_counts = prm->emit_counts != 0;
    double budget = 0.70 * (double)ctx->mem_total;
    if (const char* e = getenv("LGMI_MEM_BUDGET_MB")) { const double v = atof(e); if (v > 0.0) budget = v * 1048576.0; }
    for (const auto& m : db->split_memo)
        if (m.het_only == (prm->het_only != 0) && m.want_p == want_p && m.keep_p == keep_p && m.want_counts == want_counts &&
            m.n_shuffles == prm->n_shuffles && m.budget == budget) { if (largest) *largest = m.largest; return m.k; }
    uint64_t most_seen = 0;
    const double fixed = (double)pl.total_slots * 16.0 + (double)pl.op_total * 16.0;       // what every shard allocates in full
    const double rows_work = (double)pl.n_examined * (double)row_bytes_working(want_p, keep_p, want_counts);
    const double rows_final = (double)pl.n_examined * (double)row_bytes_final(want_p, keep_p, want_counts);
    int k = 1;
    double room = budget - fixed - rows_final;
    if (fixed + rows_work > budget) {
        // what is left for a shard's own row arrays once the slot matrix, the operands and the final rows are in;
        // when those three alone exceed the budget (they are floors: the slot matrix keeps its full addressing and the
        // final rows are what the caller asked for) the shards are sized to a quarter of it
        if (room < 0.25 * budget) room = 0.25 * budget;
        k = (int)std::ceil(1.1 * rows_work / room);          // a first guess: even row counts
    }
    const bool rows32 = want_p && pl.n_examined >= 0xFFFFFFFFull;
    if (rows32) k = std::max<int>(k, (int)(pl.n_examined / 0xC0000000ull) + 1);
    if (k > 1) {
        // Shards are balanced by COST — with p-values mostly the larger-than-2x2 rows, and since round 4 the rows that walk
        // a column of a large slot matrix — so their row counts can be far from even (the last of eight shards of a dense
        // chromosome holds a third more rows than the first).  The shards' plans are built (host only) and the count raised
        // until the LARGEST shard's working rows fit the room, and — the permutation kernels carry row numbers in 32 bits —
        // every shard examines fewer than 2^32 pairs.
        int ck; uint32_t xg;
        plan_env(&ck, &xg);
        const double per_row = (double)row_bytes_working(want_p, keep_p, want_counts);
        for (; k < 1024; k += std::max(1, k / 4)) {
            uint64_t most = 0;
            for (int r = 0; r < k; ++r) {
                Plan sp;
                build_plan(plan_input(db), prm->het_only != 0, (uint32_t)r, (uint32_t)k, ck, xg, prm->n_shuffles, sp);
                most = std::max(most, sp.n_examined);
            }
            const bool fits_mem = fixed + rows_work <= budget || 1.05 * (double)most * per_row <= room;
            const bool fits_32 = !want_p || most < 0xFFFFFFF0ull;
            most_seen = most;
            if (fits_mem && fits_32) break;
        }
        if (k >= 1024) {                                     // the clamp: the hint must be the clamped count's largest shard, not the last candidate's
            k = 1024; most_seen = 0;
            for (int r = 0; r < k; ++r) { Plan sp; build_plan(plan_input(db), prm->het_only != 0, (uint32_t)r, (uint32_t)k, ck, xg, prm->n_shuffles, sp); most_seen = std::max(most_seen, sp.n_examined); }
        }
    }
    k = std::min(std::max(k, 1), 1024);
    if (largest) *largest = most_seen;
    if (db->split_memo.size() >= 8) db->split_memo.erase(db->split_memo.begin());
    db->split_memo.push_back({prm->het_only != 0, want_p, keep_p, want_counts, prm->n_shuffles, budget, k, most_seen});
    if (ms_host) *ms_host = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_enter).count();
    return k;
}

static int run_device_split(lgmi_ctx* ctx, const lgmi_dbatch* db, const lgmi_params* prm, int k, uint64_t bound_rows,
                            lgmi_dresult** out, uint64_t largest_shard = 0, float ms_split_host = 0.f) {
    const bool want_p = prm->n_shuffles > 0 || prm->exact_2x2;
    const bool p_from_exceed = prm->n_shuffles > 0 && !prm->exact_2x2;
    const bool keep_p = want_p && !(p_from_exceed && prm->no_row_p);
    const bool want_counts = prm->emit_counts != 0;
    HIPCHK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    Pool& pool = ctx->pool;
    const uint32_t ns = (uint32_t)db->d.n_sites;
    lgmi_dresult* res = new lgmi_dresult();
    res->ctx = ctx; res->n_sites = ns; res->has_p = want_p; res->has_counts = want_counts;
    res->n_shuffles = prm->n_shuffles; res->p_from_exceed = p_from_exceed;
    res->table = db->table; res->het_only = prm->het_only != 0;
    struct Guard { lgmi_dresult* r; ~Guard() { if (r) lgmi_dresult_free(r); } } guard{res};
    int rc;
    const size_t nr = (size_t)std::max<uint64_t>(bound_rows, 1);
    if ((rc = pool.alloc((void**)&res->d_i, nr * 4))) return rc;
    if ((rc = pool.alloc((void**)&res->d_j, nr * 4))) return rc;
    if ((rc = pool.alloc((void**)&res->d_mi, nr * 8))) return rc;
    if (want_counts && (rc = pool.alloc((void**)&res->d_counts, nr * 36))) return rc;
    if (want_p && (rc = pool.alloc((void**)&res->d_exceed, nr * 4))) return rc;
    if (keep_p && (rc = pool.alloc((void**)&res->d_p, nr * 8))) return rc;
    if ((rc = pool.alloc((void**)&res->d_sum, std::max<size_t>(ns, 1) * 8))) return rc;
    if ((rc = pool.alloc((void**)&res->d_npairs, std::max<size_t>(ns, 1) * 4))) return rc;
    if ((rc = pool.alloc((void**)&res->d_mean, std::max<size_t>(ns, 1) * 8))) return rc;
    if ((rc = pool.alloc((void**)&res->d_nfirst, std::max<size_t>(ns, 1) * 4))) return rc;
    if ((rc = pool.alloc((void**)&res->d_ncand, std::max<size_t>(ns, 1) * 4))) return rc;
    if (ns) {
        HIPCHK(hipMemsetAsync(res->d_sum, 0, (size_t)ns * 8, st));
        HIPCHK(hipMemsetAsync(res->d_npairs, 0, (size_t)ns * 4, st));
        HIPCHK(hipMemsetAsync(res->d_nfirst, 0, (size_t)ns * 4, st));
    }
    lgmi_run_info tot = {};
    uint64_t off = 0;
    for (int s = 0; s < k; ++s) {
        lgmi_params ps = *prm;
        ps.shard_rank = (uint16_t)s; ps.shard_world = (uint16_t)k;
        lgmi_dresult* part = nullptr;
        if ((rc = run_device_impl(ctx, db, &ps, &part, false, largest_shard))) return rc;
        struct PartGuard { lgmi_dresult* p; ~PartGuard() { lgmi_dresult_free(p); } } pg{part};
        const uint64_t n = part->n_rows;
        if (off + n > bound_rows) return fail(LGMI_E_STATE, "internal: the shards' rows exceed the planned bound");
        auto d2d = [&](void* dst, const void* src, size_t bytes) -> hipError_t {
            return bytes ? hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st) : hipSuccess;
        };
        HIPCHK(d2d(res->d_i + off, part->d_i, n * 4));
        HIPCHK(d2d(res->d_j + off, part->d_j, n * 4));
        HIPCHK(d2d(res->d_mi + off, part->d_mi, n * 8));
        if (want_counts) HIPCHK(d2d(res->d_counts + 9 * off, part->d_counts, n * 36));
        if (want_p) HIPCHK(d2d(res->d_exceed + off, part->d_exceed, n * 4));
        if (keep_p) HIPCHK(d2d(res->d_p + off, part->d_p, n * 8));
        launch_sites_add(st, ns, res->d_sum, part->d_sum, res->d_npairs, part->d_npairs);
        launch_add_u32(st, ns, res->d_nfirst, part->d_nfirst);                // (rows by first site: integers, like the pair counts)
        if (s == 0 && ns) HIPCHK(hipMemcpyAsync(res->d_ncand, part->d_ncand, (size_t)ns * 4, hipMemcpyDeviceToDevice, st));
        HIPCHK(hipGetLastError());
        HIPCHK(wait_stream(st));                       // the part's arrays go back to the pool next
        const lgmi_run_info& pi = part->info;
        if (s == 0) { res->first_site = part->first_site; res->first_seg = part->first_seg; }
        if (s == 0) tot = pi;
        else {
            tot.n_rows += pi.n_rows; tot.n_examined += pi.n_examined; tot.n_tile_pairs += pi.n_tile_pairs;
            tot.word_pairs += pi.word_pairs; tot.bytes_out += pi.bytes_out; tot.n_general_rows += pi.n_general_rows; tot.n_six_rows += pi.n_six_rows; tot.ms_perm_exact += pi.ms_perm_exact;
            tot.ms_total += pi.ms_total; tot.ms_prep += pi.ms_prep; tot.ms_count += pi.ms_count; tot.ms_emit += pi.ms_emit;
            tot.ms_perm += pi.ms_perm; tot.ms_mean += pi.ms_mean; tot.ms_plan_host += pi.ms_plan_host;
            tot.ms_perm_fast += pi.ms_perm_fast; tot.ms_perm_general += pi.ms_perm_general;
            tot.n_count_launches += pi.n_count_launches;
            tot.n_mfma_tiles = (uint32_t)std::min<uint64_t>((uint64_t)tot.n_mfma_tiles + pi.n_mfma_tiles, 0xFFFFFFFFull);
        }
        off += n;
    }
    launch_site_mean(st, ns, res->d_sum, res->d_npairs, res->d_mean);
    HIPCHK(hipGetLastError());
    HIPCHK(wait_stream(st));
    res->n_rows = off;
    tot.n_rows = off;
    tot.n_examined = tot.n_examined_total;             // the whole batch was run
    tot.n_seq_shards = (uint32_t)k;
    tot.ms_plan_host += ms_split_host;                 // (choosing k builds shard plans on the host: not free, so not hidden)
    res->info = tot;
    guard.r = nullptr;
    *out = res;
    return LGMI_OK;
}

extern "C" int lgmi_run_device(lgmi_ctx* ctx, const lgmi_dbatch* db, const lgmi_params* prm, lgmi_dresult** out) {
    // (the batch's plans live in ITS context's pool: nothing of it is planned, evicted or released through another one)
    if (ctx && db && db->ctx != ctx) return fail(LGMI_E_ARG, "batch belongs to another context");
    if (ctx && db && prm && out && prm->shard_world <= 1 && !getenv("LGMI_NO_AUTO_SPLIT")) {
        const Plan& pl = plan_for(ctx, db, prm->het_only != 0, 0, 1, prm->n_shuffles, nullptr)->pl;   // (the one the run itself uses)
        uint64_t largest = 0;
        float ms_split = 0.f;
        const int k = split_count(ctx, db, prm, pl, &largest, &ms_split);
        if (k > 1) return run_device_split(ctx, db, prm, k, pl.n_examined, out, largest, ms_split);
    }
    return run_device_impl(ctx, db, prm, out, false);
}

// ---- split run: rows first, permutation stage later (a multi-GPU host gathers (i, j, mi) while the stage runs)
extern "C" int lgmi_run_device_rows(lgmi_ctx* ctx, const lgmi_dbatch* db, const lgmi_params* prm, lgmi_dresult** out) {
    return run_device_impl(ctx, db, prm, out, true);
}

// The permutation stage of a result whose rows are final, optionally in `n_chunks` contiguous row ranges: the kernels key
// their Philox counters by the PAIR (row_i, row_j), never by a row number or the launch geometry, so a range of rows gives
// the same counts whether it is run alone or inside the whole.  After each range `after_chunk(first row, rows)` is called
// with the range's kernels queued on the main stream (lgmi_run ships the range's counts to the host while the next range
// is computed).  One range = the plain stage.
template <class F>
static int permute_impl(lgmi_ctx* ctx, lgmi_dresult* res, uint32_t n_chunks, F after_chunk) {
    HIPCHK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    Pool& pool = ctx->pool;
    int rc = LGMI_OK;
    uint32_t* d_genlist = nullptr; unsigned int* d_gencount = nullptr; uint64_t* d_chunk_rows = nullptr;
    struct Guard { Pool& p; uint32_t** a; unsigned int** b; uint64_t** c; ~Guard() { p.release(*a); p.release(*b); p.release(*c); } } guard{pool, &d_genlist, &d_gencount, &d_chunk_rows};
    HIPCHK(hipEventRecord(ctx->ev[3], st));
    HIPCHK(hipEventRecord(ctx->ev[6], st));
    uint64_t n_general = 0, n_six = 0;
    // a chunked stage needs the row count on the host (it has it: the rows were waited for) and at least a few thousand rows
    const uint64_t n_rows = res->n_rows;
    if (n_chunks < 1 || n_rows < 65536ull * n_chunks) n_chunks = 1;
    if (n_chunks > 32) n_chunks = 32;
    const bool chunked = n_chunks > 1;
    float ms_fast = 0.f, ms_exact = 0.f;
    enum { HS_ROWS = 512, HS_GEN = 576 };                    // pinned words: the chunks' row counts (up), their queue counters (down)
    if (chunked) for (uint32_t k = 0; k < 3 * n_chunks; ++k) if (!ctx->event(k)) return fail(LGMI_E_HIP, "hipEventCreate failed");
    // the ranges' first rows: equal ranges, except that the LAST one is half as long (what is left to copy when the stage's
    // last kernel ends is the last range's counts) — every start even, so that the 16-bit narrowing stays aligned
    uint64_t range_begin[33];
    {
        const uint64_t unit = chunked ? (n_rows + 2 * n_chunks - 2) / (2 * n_chunks - 1) : 0;      // the last range: one unit, the others two
        for (uint32_t c = 0; c <= n_chunks; ++c) range_begin[c] = c == n_chunks ? n_rows : std::min<uint64_t>(n_rows, (2 * c * unit + 1) & ~1ull);
    }
    if (res->cap_rows) {
        uint64_t chunk_cap = res->cap_rows;
        if (chunked) { chunk_cap = 0; for (uint32_t c = 0; c < n_chunks; ++c) chunk_cap = std::max(chunk_cap, range_begin[c + 1] - range_begin[c]); }
        if ((rc = pool.alloc((void**)&d_genlist, (size_t)chunk_cap * 4))) return rc;
        if ((rc = pool.alloc((void**)&d_gencount, 64 * (size_t)n_chunks))) return rc;      // per chunk: [0] queued rows, [1] k_perm_general's next row, [2] rows k_perm_enum leaves to it, [3] some row is enumerable, ...
        HIPCHK(hipMemsetAsync(d_gencount, 0, 64 * (size_t)n_chunks, st));
        unsigned long long* const hs = ctx->h_scal;
        if (chunked) {
            if ((rc = pool.alloc((void**)&d_chunk_rows, 8 * (size_t)n_chunks))) return rc;
            for (uint32_t c = 0; c < n_chunks; ++c) hs[HS_ROWS + c] = range_begin[c + 1] - range_begin[c];
            HIPCHK(hipMemcpyAsync(d_chunk_rows, &hs[HS_ROWS], 8 * (size_t)n_chunks, hipMemcpyHostToDevice, st));
        }
        for (uint32_t c = 0; c < n_chunks; ++c) {
            const uint64_t r0 = chunked ? range_begin[c] : 0, nr = chunked ? range_begin[c + 1] - range_begin[c] : res->cap_rows;
            PermArgs pa{};
            pa.n_rows_dev = chunked ? d_chunk_rows + c : res->d_nrows; pa.max_rows = nr;
            pa.row_i = res->d_i + r0; pa.row_j = res->d_j + r0; pa.counts = res->d_counts ? res->d_counts + 9 * r0 : nullptr; pa.rec = res->d_rec + r0;
            pa.G = ctx->d_G; pa.LF = ctx->d_LF;
            pa.n_shuffles = res->prm.n_shuffles; pa.seed = res->prm.seed; pa.exact_2x2 = res->prm.exact_2x2; pa.site_base = res->prm.stream_site_base;
            pa.out_p = res->d_p ? res->d_p + r0 : nullptr; pa.out_exceed = res->d_exceed + r0; pa.gen_list = d_genlist; pa.gen_count = d_gencount + 16 * c;
            hipEvent_t e_fast = chunked ? ctx->event(3 * c) : ctx->ev[6], e_exact = chunked ? ctx->event(3 * c + 1) : ctx->ev[7];
            launch_perm_fast(st, pa);
            HIPCHK(hipGetLastError());
            HIPCHK(hipEventRecord(e_fast, st));
            launch_perm_general(st, pa, e_exact);                // (records e_exact itself, also when it has nothing to do)
            HIPCHK(hipGetLastError());
            if (chunked) {
                HIPCHK(hipEventRecord(ctx->event(3 * c + 2), st));
                if ((rc = after_chunk(r0, nr))) return rc;
            }
        }
        HIPCHK(hipMemcpyAsync(&hs[HS_GEN], d_gencount, 64 * (size_t)n_chunks, hipMemcpyDeviceToHost, st));
    }
    HIPCHK(hipEventRecord(ctx->ev[4], st));
    if (!(chunked && res->cap_rows) && (rc = after_chunk(0, n_rows))) return rc;
    HIPCHK(wait_stream(st));
    if (res->cap_rows) {
        const unsigned int* gc = reinterpret_cast<const unsigned int*>(&ctx->h_scal[HS_GEN]);
        for (uint32_t c = 0; c < n_chunks; ++c) { n_general += gc[16 * c]; n_six += gc[16 * c + 6]; }
        if (chunked)
            for (uint32_t c = 0; c < n_chunks; ++c) {        // stage times of a chunked run: the chunks' sums
                float a = 0.f, b = 0.f;
                HIPCHK(hipEventElapsedTime(&a, c == 0 ? ctx->ev[3] : ctx->event(3 * c - 1), ctx->event(3 * c)));
                HIPCHK(hipEventElapsedTime(&b, ctx->event(3 * c), ctx->event(3 * c + 1)));
                ms_fast += a; ms_exact += b;
            }
    }
    lgmi_run_info& inf = res->info;
    inf.n_general_rows = n_general;
    inf.n_six_rows = (uint32_t)n_six;
    HIPCHK(hipEventElapsedTime(&inf.ms_perm, ctx->ev[3], ctx->ev[4]));
    if (chunked) {
        inf.ms_perm_fast = ms_fast; inf.ms_perm_exact = ms_exact; inf.ms_perm_general = inf.ms_perm - ms_fast;
    } else {
        HIPCHK(hipEventElapsedTime(&inf.ms_perm_fast, ctx->ev[3], ctx->ev[6]));
        HIPCHK(hipEventElapsedTime(&inf.ms_perm_general, ctx->ev[6], ctx->ev[4]));
        inf.ms_perm_exact = 0.f;
        if (res->cap_rows) HIPCHK(hipEventElapsedTime(&inf.ms_perm_exact, ctx->ev[6], ctx->ev[7]));
    }
    inf.ms_total += inf.ms_perm;
    res->perm_pending = false;
    if (!res->has_counts && res->d_counts) { pool.release(res->d_counts); res->d_counts = nullptr; }
    if (res->d_rec) { pool.release(res->d_rec); res->d_rec = nullptr; }
    return LGMI_OK;
}

extern "C" int lgmi_dresult_permute(lgmi_ctx* ctx, lgmi_dresult* res) {
    if (!ctx || !res) return fail(LGMI_E_ARG, "NULL argument");
    if (res->ctx != ctx) return fail(LGMI_E_ARG, "result belongs to another context");
    if (!res->perm_pending) return LGMI_OK;              // nothing was deferred (no p requested, or already done)
    return permute_impl(ctx, res, 1u, [](uint64_t, uint64_t) { return LGMI_OK; });
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
    v->row_p = r->has_p ? r->d_p : nullptr;          /* NULL when lgmi_params.no_row_p dropped it */
    v->n_shuffles = r->n_shuffles; v->row_p_derived = (r->has_p && !r->d_p) ? 1u : 0u;
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

// ---- the compact row form on the device: full flags, row_begin, the listed partners (include/lgmi.h, lgmi_result).
// Made on the main stream from the per-site integers every run leaves behind (d_nfirst, d_ncand); one 8-byte read-back
// sizes the list.  Idempotent.
static int compact_prepare(lgmi_dresult* r) {
    if (r->compact_ready) return LGMI_OK;
    if (!r->table || !r->d_nfirst || !r->d_ncand)
        return fail(LGMI_E_STATE, "this result has no compact form (gathered from ranks that ran different batches)");
    lgmi_ctx* ctx = r->ctx;
    hipStream_t st = ctx->stream;
    Pool& pool = ctx->pool;
    const uint32_t ns = (uint32_t)r->n_sites;
    int rc;
    uint32_t* d_listed = nullptr; uint64_t* d_list_begin = nullptr; uint64_t* d_scantmp = nullptr;
    struct Guard { Pool& p; hipStream_t st; uint32_t** a; uint64_t** b; uint64_t** c;
                   ~Guard() { (void)hipStreamSynchronize(st); p.release(*a); p.release(*b); p.release(*c); } } guard{pool, st, &d_listed, &d_list_begin, &d_scantmp};
    if ((rc = pool.alloc((void**)&r->d_full, std::max<size_t>(ns, 1)))) return rc;
    if ((rc = pool.alloc((void**)&r->d_row_begin, ((size_t)ns + 1) * 8))) return rc;
    if ((rc = pool.alloc((void**)&d_listed, std::max<size_t>(ns, 1) * 4))) return rc;
    if ((rc = pool.alloc((void**)&d_list_begin, ((size_t)ns + 1) * 8))) return rc;
    if ((rc = pool.alloc((void**)&d_scantmp, scan_tmp_words(ns) * 8))) return rc;
    launch_compact_sites(st, ns, r->d_nfirst, r->d_ncand, r->d_full, d_listed, r->d_row_begin, d_list_begin, d_scantmp);
    HIPCHK(hipGetLastError());
    unsigned long long* hs = ctx->h_scal;
    hs[5] = hs[6] = 0;
    HIPCHK(hipMemcpyAsync(&hs[5], d_list_begin + ns, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(&hs[6], r->d_row_begin + ns, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(wait_stream(st));
    if (hs[6] != r->n_rows) return fail(LGMI_E_STATE, "internal: per-site row counts add up to %llu, the result has %llu rows",
                                        (unsigned long long)hs[6], (unsigned long long)r->n_rows);
    r->n_jlisted = hs[5];
    if (r->n_jlisted) {
        if ((rc = pool.alloc((void**)&r->d_jlisted, (size_t)r->n_jlisted * 4))) return rc;
        launch_list_partners(st, ns, r->d_full, r->d_row_begin, d_list_begin, r->d_j, r->d_jlisted);
        HIPCHK(hipGetLastError());
    }
    r->compact_ready = true;
    return LGMI_OK;
}
// permutation counts in 16 bits: every count is at most n_shuffles
static bool exceed_fits16(const lgmi_dresult* r) { return r->has_p && r->p_from_exceed && r->n_shuffles <= 65535u; }

// HBM -> pinned host buffers in two parts, so that lgmi_run can bring the rows over while the permutation stage still
// runs: part 1 = everything that is final after the emit stage, part 2 = row_p / row_exceed of the rows [r0, r0 + nr)
// (part 2 is called once per chunk of a chunked permutation stage; its host arrays are taken on the first call)
static int fetch_part(lgmi_dresult* r, HostResult* h, lgmi_result* out, hipStream_t st, int part, bool compact,
                      uint64_t r0 = 0, uint64_t nr = ~0ull) {
    const size_t n = (size_t)r->n_rows, ns = (size_t)r->n_sites;
    auto d2h = [&](void* dst, const void* src, size_t bytes) -> hipError_t {
        return bytes ? hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, st) : hipSuccess;
    };
    if (part == 1) {
        double* hmi = h->take<double>(n);
        double* hmean = h->take<double>(ns); uint32_t* hnp = h->take<uint32_t>(ns);
        uint32_t* hc = r->has_counts ? h->take<uint32_t>(n * 9) : nullptr;
        if (!hmi || !hmean || !hnp || (r->has_counts && !hc)) return fail(LGMI_E_OOM, "pinned host memory for %zu result rows", n);
        if (compact) {
            uint64_t* hb = h->take<uint64_t>(ns + 1); uint8_t* hf = h->take<uint8_t>(ns);
            uint32_t* hl = r->n_jlisted ? h->take<uint32_t>((size_t)r->n_jlisted) : nullptr;
            if (!hb || !hf || (r->n_jlisted && !hl)) return fail(LGMI_E_OOM, "pinned host memory for %zu result rows", n);
            HIPCHK(d2h(hb, r->d_row_begin, (ns + 1) * 8));
            HIPCHK(d2h(hf, r->d_full, ns));
            if (hl) HIPCHK(d2h(hl, r->d_jlisted, (size_t)r->n_jlisted * 4));
            out->compact = 1; out->row_begin = hb; out->site_row_full = hf; out->row_j_listed = hl; out->n_row_j_listed = r->n_jlisted;
            h->table = r->table; h->het_only = r->het_only;
        } else {
            uint32_t* hi = h->take<uint32_t>(n); uint32_t* hj = h->take<uint32_t>(n);
            if (!hi || !hj) return fail(LGMI_E_OOM, "pinned host memory for %zu result rows", n);
            HIPCHK(d2h(hi, r->d_i, n * 4));
            HIPCHK(d2h(hj, r->d_j, n * 4));
            out->row_i = hi; out->row_j = hj;
        }
        HIPCHK(d2h(hmi, r->d_mi, n * 8));
        HIPCHK(d2h(hmean, r->d_mean, ns * 8));
        HIPCHK(d2h(hnp, r->d_npairs, ns * 4));
        if (r->has_counts) HIPCHK(d2h(hc, r->d_counts, n * 36));
        out->n_rows = n; out->n_sites = ns;
        out->row_mi = hmi; out->row_counts = hc;
        out->site_mean_mi = hmean; out->site_n_pairs = hnp;
    } else if (r->has_p) {
        // row_p travels only when it exists as an array (lgmi_params.no_row_p: it is (1 + exceed) / (n_shuffles + 1),
        // 8 of the 28 bytes a row used to cost on the way to the host)
        const bool narrow = compact && exceed_fits16(r);
        if (!out->row_exceed && !out->row_exceed16) {
            double* hp = r->d_p ? h->take<double>(n) : nullptr;
            uint32_t* hex = narrow ? nullptr : h->take<uint32_t>(n);
            uint16_t* hex16 = narrow ? h->take<uint16_t>(n) : nullptr;
            if ((r->d_p && !hp) || (!hex && !hex16)) return fail(LGMI_E_OOM, "pinned host memory for %zu result rows", n);
            out->row_p = hp; out->row_exceed = hex; out->row_exceed16 = hex16;
            out->n_shuffles = r->n_shuffles; out->row_p_derived = r->d_p ? 0u : 1u;
        }
        if (r0 > n) r0 = n;
        if (nr > n - r0) nr = n - r0;
        if (r->d_p) HIPCHK(d2h(const_cast<double*>(out->row_p) + r0, r->d_p + r0, nr * 8));
        if (narrow) HIPCHK(d2h(const_cast<uint16_t*>(out->row_exceed16) + r0, r->d_exceed16 + r0, nr * 2));
        else HIPCHK(d2h(const_cast<uint32_t*>(out->row_exceed) + r0, r->d_exceed + r0, nr * 4));
    }
    return LGMI_OK;
}

static int fetch_impl(lgmi_dresult* r, lgmi_result* out, bool compact) {
    if (!r || !out) return fail(LGMI_E_ARG, "NULL argument");
    memset(out, 0, sizeof *out);
    if (r->perm_pending) return fail(LGMI_E_STATE, "the permutation stage of this result has not run (lgmi_dresult_permute)");
    HIPCHK(hipSetDevice(r->ctx->device));
    hipStream_t st = r->ctx->stream;
    int rc;
    if (compact) {
        if ((rc = compact_prepare(r))) return rc;
        if (exceed_fits16(r) && !r->d_exceed16 && r->n_rows) {
            if ((rc = r->ctx->pool.alloc((void**)&r->d_exceed16, (size_t)r->n_rows * 2 + 4))) return rc;
            launch_narrow_u16(st, r->n_rows, r->d_exceed, r->d_exceed16);
            HIPCHK(hipGetLastError());
        }
    }
    HostResult* h = new HostResult();
    h->pool = r->ctx->pinned;
    struct Guard { HostResult* p; ~Guard() { delete p; } } guard{h};
    rc = fetch_part(r, h, out, st, 1, compact);
    if (!rc) rc = fetch_part(r, h, out, st, 2, compact);
    if (rc) { (void)hipStreamSynchronize(st); memset(out, 0, sizeof *out); return rc; }
    HIPCHK(wait_stream(st));
    out->owner_ = static_cast<ResultOwner*>(h);
    guard.p = nullptr;
    return LGMI_OK;
}
extern "C" int lgmi_dresult_fetch(lgmi_dresult* r, lgmi_result* out) { return fetch_impl(r, out, false); }
extern "C" int lgmi_dresult_fetch_compact(lgmi_dresult* r, lgmi_result* out) { return fetch_impl(r, out, true); }

// compact form -> plain row_i / row_j on the host (include/lgmi.h).  Threads take site ranges of equal row counts.
extern "C" int lgmi_result_expand_rows(const lgmi_result* res, uint32_t* row_i, uint32_t* row_j) {
    if (!res) return fail(LGMI_E_ARG, "res is NULL");
    const uint64_t n = res->n_rows, ns = res->n_sites;
    if (!res->compact) {
        if ((row_i && !res->row_i) || (row_j && !res->row_j)) return fail(LGMI_E_STATE, "the result holds no row_i / row_j");
        if (row_i && n) memcpy(row_i, res->row_i, n * 4);
        if (row_j && n) memcpy(row_j, res->row_j, n * 4);
        return LGMI_OK;
    }
    const HostResult* h = dynamic_cast<const HostResult*>(static_cast<const ResultOwner*>(res->owner_));
    if (!h || !h->table || !res->row_begin || !res->site_row_full) return fail(LGMI_E_STATE, "not a compact result of this library");
    const SiteTable& tb = *h->table;
    if (tb.type.size() != ns) return fail(LGMI_E_STATE, "internal: site table of %zu sites, result of %llu", tb.type.size(), (unsigned long long)ns);
    const bool het_only = h->het_only;
    const uint64_t nb = tb.block_site_begin.size() - 1;
    const uint64_t* const rb = res->row_begin;
    // offsets of the listed partners: a prefix over the sites that are not full
    PodVec<uint64_t> lb;
    lb.resize(ns + 1);
    { uint64_t o = 0; for (uint64_t s = 0; s < ns; ++s) { lb[s] = o; if (!res->site_row_full[s]) o += rb[s + 1] - rb[s]; } lb[ns] = o; }
    if (lb[ns] != res->n_row_j_listed) return fail(LGMI_E_STATE, "internal: %llu listed partners, the flags ask for %llu",
                                                   (unsigned long long)res->n_row_j_listed, (unsigned long long)lb[ns]);
    Team team((unsigned)std::max<uint64_t>(1, std::min<uint64_t>(plan_threads(1u << 20), n >> 20)));
    const unsigned T = team.T;
    team.run([&](unsigned t) {
        // a thread writes the sites whose first row falls into its share of the rows (it walks every block's site list — a
        // site's implicit partners are the later sites of its block — and skips the blocks outside its share)
        const uint64_t r_lo = n * t / T, r_hi = n * (t + 1) / T;
        std::vector<uint32_t> xs;
        for (uint64_t b = 0; b < nb; ++b) {
            const uint64_t sb = tb.block_site_begin[b], se = tb.block_site_begin[b + 1];
            if (rb[se] <= r_lo || rb[sb] >= r_hi) continue;
            bool have_xs = false;
            uint32_t xnext = 0;                                   // x sites of the block at or before s
            for (uint64_t s = sb; s < se; ++s) {
                const bool is_x = !het_only || tb.type[s] == LGMI_TYPE_HET_SNP;
                if (is_x) ++xnext;
                const uint64_t r0 = rb[s], cnt = rb[s + 1] - r0;
                if (!cnt || r0 < r_lo || r0 >= r_hi) continue;   // (a site's rows belong to the thread its first row falls to)
                if (row_i) std::fill(row_i + r0, row_i + r0 + cnt, (uint32_t)s);
                if (!row_j) continue;
                if (!res->site_row_full[s]) { memcpy(row_j + r0, res->row_j_listed + lb[s], cnt * 4); continue; }
                if (is_x) { for (uint64_t k = 0; k < cnt; ++k) row_j[r0 + k] = (uint32_t)(s + 1 + k); continue; }
                if (!have_xs) {                                   // the block's x sites, once per thread that needs them
                    xs.clear();
                    for (uint64_t q = sb; q < se; ++q) if (tb.type[q] == LGMI_TYPE_HET_SNP) xs.push_back((uint32_t)q);
                    have_xs = true;
                }
                memcpy(row_j + r0, xs.data() + xnext, cnt * 4);
            }
        }
    });
    return LGMI_OK;
}

// upload + run + fetch in one call.  With permutation p-values the run is split: the rows start their way to the host
// (communication stream) while the permutation stage runs on the main stream — and the stage itself runs in a few row
// ranges (LGMI_PERM_CHUNKS, default 4 from 32 M rows on), each range's counts following the rows as soon as its kernels
// are done, so that what is left to copy when the last kernel ends is one range's counts.
extern "C" int lgmi_run(lgmi_ctx* ctx, const lgmi_batch* batch, const lgmi_params* prm, lgmi_result* out,
                        lgmi_run_info* info) {
    if (!out) return fail(LGMI_E_ARG, "out is NULL");
    memset(out, 0, sizeof *out);
    HostTrace tr;
    lgmi_dbatch* db = nullptr;
    // the planes may follow the rest of the batch piece by piece, under the count kernels (UploadPipe) — when the caller's
    // tri flags let the run be planned without them, the planes are packed in site order and one launch sequence holds the run
    std::unique_ptr<UploadPipe> pipe;
    const bool piped = prm && prm->shard_world <= 1 && pipe_eligible(batch);
    if (piped) pipe.reset(new UploadPipe());
    int rc = upload_impl(ctx, batch, &db, pipe.get());
    if (rc) return rc;
    tr.mark("run:uploaded");
    lgmi_dresult* dr = nullptr;
    bool split = false;
    if (prm && prm->shard_world <= 1 && !getenv("LGMI_NO_AUTO_SPLIT")) {
        // more than one launch sequence holds: sequential shards (lgmi_run_device), no overlap of fetch and permutation
        const Plan& pl = plan_for(ctx, db, prm->het_only != 0, 0, 1, prm->n_shuffles, nullptr)->pl;
        const uint64_t bound = pl.n_examined;               // (before the shards' plans may evict this one)
        uint64_t largest = 0;
        float ms_split = 0.f;
        const int k = split_count(ctx, db, prm, pl, &largest, &ms_split);
        if (k > 1) {
            if (pipe) {                                      // the shards need all the planes: bring them up now, in one go
                for (uint32_t c = 0; c < pipe->K && !rc; ++c) { hipEvent_t e = ctx->event(lgmi_ctx::EV_UP + c); rc = e ? pipe_chunk(*pipe, db, c, e) : fail(LGMI_E_HIP, "hipEventCreate failed"); }
                if (!rc && (hipStreamSynchronize(pipe->us) != hipSuccess || (pipe->ps && hipStreamSynchronize(pipe->ps) != hipSuccess))) rc = fail(LGMI_E_HIP, "upload failed");
                if (!rc) { int bad = 0; if (hipMemcpy(&bad, pipe->d_bad, 4, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(LGMI_E_HIP, "upload check failed"); else if (bad) rc = fail(LGMI_E_ARG, "lgmi_batch.site_tri disagrees with the planes"); }
                pipe.reset();
            }
            split = true;
            if (!rc) rc = run_device_split(ctx, db, prm, k, bound, &dr, largest, ms_split);
        }
    }
    tr.mark("planned");
    if (!split && !rc) {
        rc = run_device_impl(ctx, db, prm, &dr, true, 0, pipe.get());
        tr.mark("impl_done");
        pipe.reset();                                        // (waits for the upload stream; releases the planes' staging copies)
    }
    tr.mark("rows");
    if (!rc) {
        const bool compact = prm->compact_rows != 0;
        if (compact) rc = compact_prepare(dr);
        tr.mark("compact");
        HostResult* h = new HostResult();
        h->pool = ctx->pinned;
        hipStream_t cs = ctx_copy_stream(ctx), ms = ctx->stream;
        hipEvent_t ce = ctx->event(lgmi_ctx::EV_ORDER);
        if (!rc && cs != ms) {                                   // the communication stream behind what made the rows (and the compact form)
            if (!ce) rc = fail(LGMI_E_HIP, "no event for the communication stream");
            else if (hipEventRecord(ce, ms) != hipSuccess || hipStreamWaitEvent(cs, ce, 0) != hipSuccess) rc = fail(LGMI_E_HIP, "stream ordering failed in lgmi_run");
        }
        if (!rc) rc = fetch_part(dr, h, out, cs, 1, compact);   // in flight under the permutation stage
        tr.mark("fetch1_posted");
        if (!rc && dr->perm_pending) {
            const bool narrow = compact && exceed_fits16(dr);
            if (narrow && dr->n_rows) rc = ctx->pool.alloc((void**)&dr->d_exceed16, (size_t)dr->n_rows * 2 + 4);
            uint32_t n_chunks = dr->n_rows >= (32ull << 20) ? 4u : 1u;
            if (const char* e = getenv("LGMI_PERM_CHUNKS")) n_chunks = (uint32_t)std::max(1, atoi(e));
            uint32_t k_ev = 0;
            if (!rc) rc = permute_impl(ctx, dr, n_chunks, [&](uint64_t r0, uint64_t nr) -> int {
                // the range's counts: narrowed behind its kernels, then on their way while the next range is computed
                if (narrow) { launch_narrow_u16(ms, nr, dr->d_exceed + r0, dr->d_exceed16 + r0); HIPCHK(hipGetLastError()); }
                if (cs != ms) {
                    hipEvent_t e = ctx->event(lgmi_ctx::EV_XFER + k_ev++);
                    if (!e) return fail(LGMI_E_HIP, "hipEventCreate failed");
                    HIPCHK(hipEventRecord(e, ms));
                    HIPCHK(hipStreamWaitEvent(cs, e, 0));
                }
                return fetch_part(dr, h, out, cs, 2, compact, r0, nr);
            });
        } else if (!rc) {
            // no deferred stage (no p-values, or a run cut into sequential shards whose shards ran theirs): one copy
            if (compact && exceed_fits16(dr) && dr->n_rows) {
                rc = ctx->pool.alloc((void**)&dr->d_exceed16, (size_t)dr->n_rows * 2 + 4);
                if (!rc) { launch_narrow_u16(cs, dr->n_rows, dr->d_exceed, dr->d_exceed16); if (hipGetLastError() != hipSuccess) rc = fail(LGMI_E_HIP, "narrowing kernel"); }
            }
            if (!rc) rc = fetch_part(dr, h, out, cs, 2, compact);
        }
        tr.mark("permuted");
        if (!rc && hipStreamSynchronize(cs) != hipSuccess) rc = fail(LGMI_E_HIP, "hipStreamSynchronize failed in lgmi_run");
        if (rc) { (void)hipStreamSynchronize(cs); (void)hipStreamSynchronize(ms); delete h; memset(out, 0, sizeof *out); }
        else {
            out->owner_ = static_cast<ResultOwner*>(h);
            dr->info.bytes_out = dr->n_rows * (8ull + (dr->has_counts ? 36ull : 0ull) + (dr->d_p ? 8ull : 0ull) +
                                               (dr->has_p ? (out->row_exceed16 ? 2ull : 4ull) : 0ull) + (compact ? 0ull : 8ull)) +
                                 (compact ? 4ull * dr->n_jlisted + 9ull * dr->n_sites : 0ull);
            if (info) *info = dr->info;
        }
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
    HIPCHK(wait_stream(st));
    return LGMI_OK;
}

// ---------------------------------------------------------------- device self-test of le_exp (perm.hip)
extern "C" int lgmi_selftest_le_exp(lgmi_ctx* ctx, uint64_t n, const double* x2, const double* t, uint8_t* fast,
                                    uint8_t* det, double* e_hw, double* e_det) {
    if (!ctx || (n && (!x2 || !t || !fast || !det || !e_hw || !e_det))) return fail(LGMI_E_ARG, "NULL argument");
    HIPCHK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    Pool& pool = ctx->pool;
    std::vector<void*> scratch;
    struct Guard { Pool& p; std::vector<void*>& s; ~Guard() { for (void* q : s) p.release(q); } } guard{pool, scratch};
    auto salloc = [&](void** p, size_t bytes) { int e = pool.alloc(p, bytes); if (!e) scratch.push_back(*p); return e; };
    double *dx, *dt, *dh, *dd; uint8_t *df, *ddet;
    int rc;
    if ((rc = salloc((void**)&dx, n * 8)) || (rc = salloc((void**)&dt, n * 8)) || (rc = salloc((void**)&dh, n * 8)) ||
        (rc = salloc((void**)&dd, n * 8)) || (rc = salloc((void**)&df, n)) || (rc = salloc((void**)&ddet, n))) return rc;
    if (n) {
        HIPCHK(hipMemcpyAsync(dx, x2, n * 8, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(dt, t, n * 8, hipMemcpyHostToDevice, st));
        launch_selftest_le_exp(st, n, dx, dt, df, ddet, dh, dd);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(fast, df, n, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(det, ddet, n, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(e_hw, dh, n * 8, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(e_det, dd, n * 8, hipMemcpyDeviceToHost, st));
    }
    HIPCHK(wait_stream(st));
    return LGMI_OK;
}

// ---------------------------------------------------------------- device self-test of mi_log (emit.hip)
extern "C" int lgmi_selftest_log(lgmi_ctx* ctx, uint64_t n, const double* x, double* out) {
    if (!ctx || (n && (!x || !out))) return fail(LGMI_E_ARG, "NULL argument");
    HIPCHK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    Pool& pool = ctx->pool;
    double *dx = nullptr, *dy = nullptr;
    struct Guard { Pool& p; double** a; double** b; ~Guard() { p.release(*a); p.release(*b); } } guard{pool, &dx, &dy};
    int rc;
    if ((rc = pool.alloc((void**)&dx, std::max<uint64_t>(n, 1) * 8)) || (rc = pool.alloc((void**)&dy, std::max<uint64_t>(n, 1) * 8))) return rc;
    if (n) {
        HIPCHK(hipMemcpyAsync(dx, x, n * 8, hipMemcpyHostToDevice, st));
        launch_selftest_log(st, n, dx, dy);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(out, dy, n * 8, hipMemcpyDeviceToHost, st));
    }
    HIPCHK(wait_stream(st));
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
    HIPCHK(wait_stream(st));
    return LGMI_OK;
}
