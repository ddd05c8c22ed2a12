// plan.h — per-run planning on the host: slot-matrix layout of every block, the count-tile list, the emit work
// items, and the cut of all of it into cost-balanced shards (one per GPU).  Pure host code: no HIP call, so the
// same planner serves lgmi_run_device() and the GPU-less lgmi_plan_shard().
//
// Reference shape being replaced: script/giremi.py:367-394 cuts the footprint list into chunks for Pool.map; pairs
// never cross a (footprint, strand) block (src/giremi/mismatch.py:387-391) and inside a block every pair is
// independent (src/giremi/mutual_information.py:12), so ANY cut of the ordered row list is a valid shard.
#pragma once
#include <vector>

#include "lgmi_internal.h"

namespace lgmi {

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
    std::vector<uint32_t> xlist, ylist;
    std::vector<SiteMap> smap;
    bool mfma_fp4 = true;           // every matrix-core block has fewer than 2^24 reads: f32 accumulation is exact
    std::vector<Tile> tiles;        // 64 x 64 tiles for k_count (VALU popcount) — this shard's
    std::vector<Tile> mtiles;       // 128 x 128 tiles for the matrix-core count kernels — this shard's
    std::vector<OpGroup> op_groups; // FP4 matrix-core blocks: the 32-column operand groups this shard's tiles read
    uint64_t op_total = 0;          // uint4 entries of the re-laid operand buffer
    uint32_t op_max_steps = 0;
    std::vector<uint2> items;       // emit work items of the WHOLE batch: (site, segment of EMIT_SEG partners), in row order
    uint64_t item_begin = 0, item_end = 0;   // this shard's items
    std::vector<uint2> units;       // emit work units of this shard: (first item - item_begin, n items | kind << 16)
    uint64_t total_slots = 0, n_examined = 0, n_examined_total = 0, bytes_in = 0;
};

// count_kernel: 0 auto, 1 VALU popcount only, 2 matrix cores only, 3 matrix cores with int8 operands only;
// xg_override: 0 = default group of x-tile rows that sweep the y tiles together
// n_shuffles only prices the work items (a pair with a tri-allelic site costs n_shuffles table draws): it moves the
// shard boundaries, never the rows
void build_plan(const PlanInput& in, bool het_only, uint32_t shard_rank, uint32_t shard_world, int count_kernel,
                uint32_t xg_override, uint32_t n_shuffles, Plan& pl);

}  // namespace lgmi
