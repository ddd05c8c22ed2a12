// plan.h — per-run planning on the host: slot-matrix layout of every block, the count-tile list, the emit work
// items, and the cut of all of it into cost-balanced shards (one per GPU).  Pure host code: no HIP call, so the
// same planner serves lgmi_run_device() and the GPU-less lgmi_plan_shard().
//
// Reference shape being replaced: script/giremi.py:367-394 cuts the footprint list into chunks for Pool.map; pairs
// never cross a (footprint, strand) block (src/giremi/mismatch.py:387-391) and inside a block every pair is
// independent (src/giremi/mutual_information.py:12), so ANY cut of the ordered row list is a valid shard.
#pragma once
#include <atomic>
#include <memory>
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
// phases are a few hundred microseconds each: a spawn / join round per phase costs more than the work)
struct Team {
    unsigned T;
    std::atomic<unsigned> count{0}, gen{0};
    explicit Team(unsigned t) : T(t) {}
    void barrier() {
        if (T <= 1) return;
        const unsigned g = gen.load(std::memory_order_acquire);
        if (count.fetch_add(1, std::memory_order_acq_rel) + 1 == T) { count.store(0, std::memory_order_relaxed); gen.fetch_add(1, std::memory_order_acq_rel); }
        else while (gen.load(std::memory_order_acquire) == g) std::this_thread::yield();
    }
    template <class F> void run(F body) {                   // body(thread)
        std::vector<std::thread> th;
        for (unsigned t = 1; t < T; ++t) th.emplace_back(body, t);
        body(0u);
        for (auto& x : th) x.join();
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
    PodVec<SiteMap> smap;
    bool mfma_fp4 = true;           // every matrix-core block has fewer than 2^24 reads: f32 accumulation is exact
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
