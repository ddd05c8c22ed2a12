// plan.h — per-run planning on the host: slot-matrix layout of every block, the count-tile list, the emit work
// items, and the cut of all of it into cost-balanced shards (one per GPU).  Pure host code: no HIP call, so the
// same planner serves lgmi_run_device() and the GPU-less lgmi_plan_shard().
//
// Reference shape being replaced: script/giremi.py:367-394 cuts the footprint list into chunks for Pool.map; pairs
// never cross a (footprint, strand) block (src/giremi/mismatch.py:387-391) and inside a block every pair is
// independent (src/giremi/mutual_information.py:12), so ANY cut of the ordered row list is a valid shard.
#pragma once
#include <atomic>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <utility>
#include <vector>

#include "lgmi_internal.h"

namespace lgmi {

// std::vector whose resize() leaves trivially-constructible elements uninitialised: the planner's arrays of a million
// sites are written once, in parallel, by the threads that then read them (a value-initialising resize would touch every
// page first, on one thread)
template <class T> struct NoInitAlloc : std::allocator<T> {
    template <class U> struct rebind { using other = NoInitAlloc<U>; };
    template <class U, class... A> void construct(U* p, A&&... a) {
        if constexpr (sizeof...(A) == 0) ::new ((void*)p) U; else ::new ((void*)p) U(std::forward<A>(a)...);
    }
};
template <class T> using PodVec = std::vector<T, NoInitAlloc<T>>;

// one team of host threads for a multi-phase pass over a batch: spawned once, phases separated by a spinning barrier (the
// phases are a few hundred microseconds each: a spawn / join round per phase costs more than the work).
// The threads are made by the CONSTRUCTOR and parked: T is what the system granted, not what was asked for — under a
// container's thread / pid limit std::thread throws, and an exception that unwinds a vector of joinable threads (or leaves
// fewer threads than a barrier waits for) ended the host process where an LGMI_E_* code belongs (advice r4).  Callers size
// their per-thread arrays and ranges by team.T after construction.
struct Team {
    unsigned T = 1;
    std::atomic<unsigned> count{0}, gen{0};
    explicit Team(unsigned want) {
        for (unsigned t = 1; t < want; ++t) {
            try { th.emplace_back([this, t] { park(t); }); } catch (...) { break; }
            ++T;
        }
    }
    ~Team() {
        { std::lock_guard<std::mutex> lk(mu); quit = true; }
        cv.notify_all();
        for (auto& x : th) x.join();
    }
    Team(const Team&) = delete;
    Team& operator=(const Team&) = delete;
    void barrier() {
        if (T <= 1) return;
        const unsigned g = gen.load(std::memory_order_acquire);
        if (count.fetch_add(1, std::memory_order_acq_rel) + 1 == T) { count.store(0, std::memory_order_relaxed); gen.fetch_add(1, std::memory_order_acq_rel); }
        else while (gen.load(std::memory_order_acquire) == g) std::this_thread::yield();
    }
    template <class F> void run(F body) {                   // body(thread), on every thread of the team; returns when all are done
        if (T > 1) {
            { std::lock_guard<std::mutex> lk(mu); job = [&body](unsigned t) { body(t); }; ++job_no; left = T - 1; }
            cv.notify_all();
        }
        body(0u);
        if (T > 1) { std::unique_lock<std::mutex> lk(mu); done.wait(lk, [this] { return left == 0; }); }
    }
private:
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv, done;
    std::function<void(unsigned)> job;
    unsigned job_no = 0, left = 0;
    bool quit = false;
    void park(unsigned t) {
        unsigned seen = 0;
        for (;;) {
            std::function<void(unsigned)> f;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return quit || job_no != seen; });
                if (quit) return;
                seen = job_no; f = job;
            }
            f(t);
            { std::lock_guard<std::mutex> lk(mu); if (--left == 0) done.notify_all(); }
        }
    }
};

// what planning reads of a batch (host views; cols holds the real sites then the pseudo columns)
struct PlanInput {
    uint64_t n_blocks = 0, n_sites = 0;
    const uint64_t* block_site_begin = nullptr;   // [n_blocks + 1]
    const uint32_t* block_n_reads = nullptr;      // [n_blocks]
    const uint8_t* type = nullptr;                // [n_sites]
    const uint8_t* tri = nullptr;                 // [n_sites] 1: the site has class-0 reads (pseudo column exists)
    const Col* cols = nullptr;                    // [n_cols]
    const uint32_t* pseudo_of_site = nullptr;     // [n_sites] column id of the pseudo column or NONE
};

struct Plan {
    std::vector<BlockPlan> plans;
    PodVec<uint32_t> xlist, ylist;
    PodVec<uint32_t> xrows;         // parallel to ylist; its x part holds the slot-matrix ROW of the x site of that rank (pseudo rows sit in between)
    PodVec<SiteMap> smap;
    bool mfma_fp4 = true;           // every matrix-core block has fewer than 2^24 reads: f32 accumulation is exact
    bool rows_are_ranks = true;     // pseudo rows behind the last x site of every block: a real x row's number is its site's x rank
    std::vector<Tile> tiles;        // 64 x 64 tiles for k_count (VALU popcount) — this shard's
    std::vector<Tile> mtiles;       // 128 x 128 tiles for the matrix-core count kernels — this shard's
    std::vector<OpGroup> op_groups; // FP4 matrix-core blocks: the 32-column operand groups this shard's tiles read
    uint64_t op_total = 0;          // uint4 entries of the re-laid operand buffer
    uint32_t op_max_steps = 0;
    PodVec<uint2> items;            // emit work items of the WHOLE batch: (site, segment of EMIT_SEG partners), in row order
    uint64_t item_begin = 0, item_end = 0;   // this shard's items
    std::vector<uint2> units;       // emit work units of this shard: (first item - item_begin, n items | kind << 16)
    uint64_t total_slots = 0, n_examined = 0, n_examined_total = 0, bytes_in = 0;
};

// count_kernel: 0 auto, 1 VALU popcount only, 2 matrix cores only, 3 matrix cores with int8 operands only;
// xg_override: 0 = default group of x-tile rows that sweep the y tiles together
// n_shuffles only prices the work items (a pair with a tri-allelic site costs n_shuffles table draws): it moves the
// shard boundaries, never the rows
// threads a pass over n_units blocks / chunks is worth (LGMI_PLAN_THREADS; one per 256 units, 16 at most)
unsigned plan_threads(uint64_t n_units);

void build_plan(const PlanInput& in, bool het_only, uint32_t shard_rank, uint32_t shard_world, int count_kernel,
                uint32_t xg_override, uint32_t n_shuffles, Plan& pl);

}  // namespace lgmi
