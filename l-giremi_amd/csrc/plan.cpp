// plan.cpp — host-side planner (see plan.h).  No HIP runtime call in this file.
//
// Layout of a block's slot matrix: rows = x sites in position order (all sites, or the het_snp ones when het_only —
// src/giremi/mismatch.py:392-396) and one pseudo row per tri x site: behind the last x site in an unsharded plan, RIGHT
// AFTER the site's own row in a sharded one (round 5: behind the last x site the pseudo rows sit in tile rows of their
// own that every shard of a multi-GPU run needs for nearly all of its columns — 4,000 of the 4,700 tiles eight shards of
// the north-star computed on top of the unsharded 29,000; next to its site's row a pseudo row lives in a tile row the
// shard computes anyway: 33,768 -> 31,669 tiles, slowest of eight shards 34.3 -> 33.2 ms on one box); columns = non-x sites,
// x sites (in x-rank order: this part of the y list is also the rank -> site table of the emit kernels), one pseudo column
// per tri site.  A row's number is therefore NOT its site's x rank: SiteMap::xrow / prow are rows, SiteMap::xnext counts ranks.  Rows of the RESULT are in the reference's order
// (itertools.combinations of the sorted positions, src/giremi/mutual_information.py:10-12): site i's row lists its
// partners j > i (x site: every later site; other site: the later x sites).  A work item is EMIT_SEG consecutive
// partners of one site.  A shard is a contiguous range of work items, so its rows are a contiguous range of the
// unsharded result, and it needs exactly the count tiles its items read slots from.
#include "plan.h"

#include <algorithm>
#include <cstdlib>

namespace lgmi {

namespace {

struct NeedMap {                       // which (x tile, y tile) of a block this shard reads slots from
    uint32_t ntx = 0, nty = 0, edge = 1;
    std::vector<uint8_t> bits;
    void reset(uint32_t ntx_, uint32_t nty_, uint32_t edge_) { ntx = ntx_; nty = nty_; edge = edge_; bits.assign((size_t)ntx * nty, 0); }
    void mark(uint32_t r0, uint32_t r1, uint32_t c0, uint32_t c1) {      // half-open slot ranges
        if (r0 >= r1 || c0 >= c1) return;
        const uint32_t tx0 = r0 / edge, tx1 = (r1 - 1) / edge, ty0 = c0 / edge, ty1 = (c1 - 1) / edge;
        for (uint32_t tx = tx0; tx <= tx1; ++tx)
            for (uint32_t ty = ty0; ty <= ty1; ++ty) bits[(size_t)tx * nty + ty] = 1;
    }
    bool get(uint32_t tx, uint32_t ty) const { return bits[(size_t)tx * nty + ty] != 0; }
};

}  // namespace

// Cost of a pair in units of one pair-word of the count kernel, from the north-star stage times:
// count 95 ms for 4.5e8 pairs x 3,125 words; emit + the 2 x 2 permutation path 63 ms for 4.5e8 pairs (15 ms without
// p-values); a pair whose table is larger than 2 x 2 — approximated here by "one of the two sites has a third class" (the
// tri flag) — costs n_shuffles table draws (212 ms for 1.74e10 of them) or, from ~200 shuffles on, the perimeter walk of
// the six-cell path whatever the shuffle count (k_perm_six + what it leaves to k_perm_general: 61 ms for 1.74e7 rows).
static const uint64_t COST_PAIR = 2000, COST_PAIR_NO_P = 500, COST_DRAW = 180, COST_SIX = 50000;
// A row of a site that is not an x site reads its slot by a walk down a column of the slot matrix, one row pitch (ny x 16
// bytes: 800 KB at north-star) per step: the further the walk reaches (x ranks xnext .. nxs), the more of it misses the TLB —
// and the more operand groups the shard's tiles make k_gather_ops lay out.  Eight shards of the north-star one after the
// other: k_emit<2> 3.56 -> 1.69 ms and k_gather_ops 1.62 -> 0.55 ms from the first shard to the last, for the same number
// of rows.  COST_WALK x (the walk's share of the x list) is added per such row (fitted to that profile), scaled down
// for blocks whose slot matrix is smaller than 4 GB.
static const uint64_t COST_WALK_DEFAULT = 2000;
static uint64_t cost_walk() { static const uint64_t v = [] { const char* e = getenv("LGMI_COST_WALK"); return e ? (uint64_t)atoll(e) : COST_WALK_DEFAULT; }(); return v; }
#define COST_WALK cost_walk()

// Blocks are independent of one another, so every per-block phase runs on several threads over contiguous block ranges
// (what a thread produces is laid down at offsets from a prefix over the blocks, or concatenated in thread = block order:
// the plan is the same whatever the thread count).  On 20,000 footprint-sized blocks the single-threaded planner was
// 29 ms of a 60 ms one-shot call (round 3); batches of a few blocks stay on the calling thread.
unsigned plan_threads(uint64_t n_blocks)
{
    static const unsigned cap = [] {
        const char* e = getenv("LGMI_PLAN_THREADS");
        const int v = e ? atoi(e) : 0;
        const unsigned hw = std::thread::hardware_concurrency();
        return (unsigned)(v > 0 ? v : std::min<unsigned>(hw ? hw : 1u, 16u));
    }();
    return (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(cap, n_blocks / 256));
}
void build_plan(const PlanInput& in, bool het_only, uint32_t shard_rank, uint32_t shard_world, int count_kernel,
                uint32_t xg_override, uint32_t n_shuffles, Plan& pl)
{
    if (shard_world == 0) { shard_world = 1; shard_rank = 0; }
    const bool sharded = shard_world > 1;
    // where a tri x site's pseudo row goes (see pass 1b); LGMI_PSEUDO_ROWS=end|next forces one (A/B runs, tests)
    static const int pr_env = [] { const char* e = getenv("LGMI_PSEUDO_ROWS"); return !e ? 0 : (e[0] == 'e' ? 1 : (e[0] == 'n' ? 2 : 0)); }();
    const bool interleave = pr_env ? pr_env == 2 : sharded;
    pl.rows_are_ranks = !interleave;
    if (count_kernel == 3) pl.mfma_fp4 = false;
    const uint64_t ns = in.n_sites, nb = in.n_blocks;
    Team team(plan_threads(nb));
    const unsigned T = team.T;                           // (what the system granted: plan.h)
    pl.smap.resize(ns);                                  // (PodVec: not initialised here — every site is written in pass 1b)
    pl.plans.resize(nb);

    // ---- pass 1a: sizes of every block's lists (x rows, y columns, work items)
    std::vector<uint64_t> blk_items(nb + 1, 0), blk_x(nb + 1, 0), blk_y(nb + 1, 0), blk_slots(nb + 1, 0);
    PodVec<uint64_t> item_cost;                           // partners x words of the block (+ the table draws)
    PodVec<uint32_t> item_ncand;
    std::vector<uint64_t> examined_part(T, 0);
    std::vector<std::vector<uint2>> part_units(T);
    std::vector<std::vector<Tile>> part_tiles(T), part_mtiles(T);
    std::vector<uint8_t> part_fp4(T, 1);
    uint64_t n_items = 0;
    team.run([&](unsigned t) {
    const uint64_t b0 = nb * t / T, b1 = nb * (t + 1) / T;
    {
        for (uint64_t b = b0; b < b1; ++b) {
            const uint32_t sb = (uint32_t)in.block_site_begin[b], se = (uint32_t)in.block_site_begin[b + 1];
            const uint32_t P = se - sb;
            uint32_t nxs = 0, ntri = 0, ntrix = 0;
            for (uint32_t s = sb; s < se; ++s) {
                const bool in_x = !het_only || in.type[s] == LGMI_TYPE_HET_SNP;
                nxs += in_x ? 1u : 0u;
                if (in.tri[s]) { ++ntri; if (in_x) ++ntrix; }
            }
            BlockPlan bp{};
            bp.site_begin = sb; bp.site_end = se;
            bp.nxs = nxs; bp.nx = nxs + ntrix; bp.ny = P + ntri;
            bp.ny_pad = (bp.ny + 3u) & ~3u;
            blk_x[b + 1] = bp.nx; blk_y[b + 1] = bp.ny;
            if (nxs == 0 || P < 2) bp.nx = 0;
            blk_slots[b + 1] = (uint64_t)bp.nx * bp.ny_pad;
            uint64_t n_it = 0;
            uint32_t xseen = 0;
            for (uint32_t s = sb; s < se; ++s) {
                const bool is_x = !het_only || in.type[s] == LGMI_TYPE_HET_SNP;
                if (is_x) ++xseen;
                const uint32_t ncand = is_x ? (se - 1 - s) : (nxs - xseen);
                n_it += ((uint64_t)ncand + (is_x ? EMIT_SEG : EMIT_SEG_Q) - 1) / (is_x ? EMIT_SEG : EMIT_SEG_Q);
            }
            blk_items[b + 1] = n_it;
            pl.plans[b] = bp;
        }
    }
    team.barrier();
    if (t == 0) {
    for (uint64_t b = 0; b < nb; ++b) {
        blk_items[b + 1] += blk_items[b]; blk_x[b + 1] += blk_x[b]; blk_y[b + 1] += blk_y[b]; blk_slots[b + 1] += blk_slots[b];
        pl.plans[b].slot_base = blk_slots[b];
        pl.plans[b].xl_off = (uint32_t)blk_x[b];
        pl.plans[b].yl_off = (uint32_t)blk_y[b];
    }
    pl.total_slots = blk_slots[nb];
    pl.xlist.resize(blk_x[nb]);
    pl.ylist.resize(blk_y[nb]);
    pl.xrows.resize(blk_y[nb]);
    n_items = blk_items[nb];
    pl.items.resize(n_items);
    item_cost.resize(sharded ? n_items : 0);
    item_ncand.resize(n_items);
    }
    team.barrier();

    // ---- pass 1b: slot-matrix layout of every block, the work items of the whole batch and their costs
    {
        std::vector<uint32_t> tri_pre, trix_pre;
        uint64_t examined = 0;
        for (uint64_t b = b0; b < b1; ++b) {
            const BlockPlan& bp = pl.plans[b];
            const uint32_t sb = bp.site_begin, se = bp.site_end, P = se - sb;
            uint32_t* const xl = pl.xlist.data() + bp.xl_off;
            uint32_t* const yl = pl.ylist.data() + bp.yl_off;
            for (uint32_t s = sb; s < se; ++s) pl.smap[s] = SiteMap{NONE, NONE, NONE, NONE, 0, 0};
            // x list: x sites in position order; the pseudo row of a tri x site right behind its own row in a SHARDED plan
            // (a shard then finds it in a tile row it computes anyway), behind the last x site otherwise (row == rank for
            // the real rows: the emit kernels' column walk measured 12 % faster that way, 13.8 against 15.5 ms at north-star)
            uint32_t nxs = 0, nx = 0;
            for (uint32_t s = sb; s < se; ++s) {
                const bool in_x = !het_only || in.type[s] == LGMI_TYPE_HET_SNP;
                if (in_x) {
                    pl.smap[s].xrow = nx; xl[nx++] = s; ++nxs;
                    if (interleave && in.tri[s]) { pl.smap[s].prow = nx; xl[nx++] = in.pseudo_of_site[s]; }
                }
                pl.smap[s].xnext = nxs;
                pl.smap[s].block = (uint32_t)b;
            }
            if (!interleave)
                for (uint32_t s = sb; s < se; ++s)
                    if (pl.smap[s].xrow != NONE && in.tri[s]) { pl.smap[s].prow = nx; xl[nx++] = in.pseudo_of_site[s]; }
            // y list: non-x sites, x sites (same order as the x list), pseudo cols of every tri site
            uint32_t ny = 0;
            for (uint32_t s = sb; s < se; ++s) if (pl.smap[s].xrow == NONE) { pl.smap[s].ycol = ny; yl[ny++] = s; }
            for (uint32_t s = sb; s < se; ++s) if (pl.smap[s].xrow != NONE) { pl.smap[s].ycol = ny; pl.xrows[bp.yl_off + ny] = pl.smap[s].xrow; yl[ny++] = s; }
            for (uint32_t s = sb; s < se; ++s) if (in.tri[s]) { pl.smap[s].pcol = ny; yl[ny++] = in.pseudo_of_site[s]; }
            // examined pairs: pairs a wave of the emit kernel will look at
            const uint64_t W = std::max<uint64_t>(1, ((uint64_t)in.block_n_reads[b] + 63u) / 64u);
            const uint64_t pair_cost = W + (n_shuffles ? COST_PAIR : COST_PAIR_NO_P);
            // (in proportion to the block's slot matrix up to 4 GB — 8.2 GB at north-star: a matrix of a few hundred
            //  megabytes stays within the TLB's reach whichever way it is walked)
            const uint64_t slot_bytes = (uint64_t)bp.nx * bp.ny_pad * 16ull;
            const uint64_t walk_cost = slot_bytes >= (1ull << 32) ? COST_WALK : COST_WALK * (slot_bytes >> 12) / (1ull << 20);
            // tri sites among the first k sites of the block / among the first k x sites: how many of an item's partners
            // bring the larger-than-2x2 paths with them
            if (sharded && n_shuffles) {
                tri_pre.assign(P + 1, 0); trix_pre.assign(nxs + 1, 0);
                uint32_t xr = 0;
                for (uint32_t k = 0; k < P; ++k) {
                    const bool tr = in.tri[sb + k] != 0;
                    tri_pre[k + 1] = tri_pre[k] + (tr ? 1u : 0u);
                    if (pl.smap[sb + k].xrow != NONE) { trix_pre[xr + 1] = trix_pre[xr] + (tr ? 1u : 0u); ++xr; }
                }
            }
            uint64_t it = blk_items[b];
            for (uint32_t s = sb; s < se; ++s) {
                const bool is_x = pl.smap[s].xrow != NONE;
                const uint32_t ncand = is_x ? (se - 1 - s) : (nxs - pl.smap[s].xnext);
                examined += ncand;
                // an x site's row is cut into segments of EMIT_SEG partners; another site's row (at most nxs partners: a
                // column walk of the slot matrix, done four sites at a time, emit.hip: emit_quad) into segments of EMIT_SEG_Q
                const uint32_t seg_len = is_x ? EMIT_SEG : EMIT_SEG_Q;
                for (uint32_t g = 0; (uint64_t)g * seg_len < ncand; ++g) {
                    pl.items[it] = make_uint2(s, g);
                    const uint32_t q_a = g * seg_len;
                    const uint32_t n_in_seg = std::min<uint32_t>(seg_len, ncand - q_a);
                    item_ncand[it] = n_in_seg;
                    if (sharded) {
                        uint64_t n_general = 0;
                        if (n_shuffles) {
                            if (in.tri[s]) n_general = n_in_seg;
                            else if (is_x) n_general = tri_pre[s + 1 + q_a + n_in_seg - sb] - tri_pre[s + 1 + q_a - sb];
                            else n_general = trix_pre[pl.smap[s].xnext + q_a + n_in_seg] - trix_pre[pl.smap[s].xnext + q_a];
                        }
                        item_cost[it] = (uint64_t)n_in_seg * pair_cost + n_general * std::min<uint64_t>((uint64_t)n_shuffles * COST_DRAW, COST_SIX);
                        if (!is_x && nxs) item_cost[it] += (uint64_t)n_in_seg * walk_cost * ncand / nxs;
                    }
                    ++it;
                }
            }
        }
        examined_part[t] = examined;
    }
    team.barrier();
    if (t == 0) {
    for (uint64_t e : examined_part) pl.n_examined_total += e;

    // ---- the shard: a contiguous, cost-balanced range of the work items
    pl.item_begin = 0;
    pl.item_end = n_items;
    if (sharded) {
        unsigned __int128 total = 0;
        for (uint64_t c : item_cost) total += c;
        const unsigned __int128 lo = total * shard_rank / shard_world, hi = total * (shard_rank + 1u) / shard_world;
        unsigned __int128 run = 0;
        uint64_t k = 0;
        while (k < n_items && run < lo) run += item_cost[k++];     // first item whose start cost is >= lo
        pl.item_begin = k;
        while (k < n_items && run < hi) run += item_cost[k++];
        pl.item_end = (shard_rank + 1u == shard_world) ? n_items : k;
        for (uint64_t q = pl.item_begin; q < pl.item_end; ++q) pl.n_examined += item_ncand[q];
    } else {
        pl.n_examined = pl.n_examined_total;
    }
    }
    team.barrier();
    // emit work units over the shard's items (indices relative to item_begin): an x site's item alone, or up to four
    // consecutive items of other sites whose slot-matrix columns share one aligned group of four (one 64-byte line).
    // A unit never spans two blocks, so the blocks' units are made side by side and concatenated in block order.
    {
        {
            const uint64_t a = std::max(blk_items[b0], pl.item_begin), e = std::min(blk_items[b1], pl.item_end);
            std::vector<uint2>& u = part_units[t];
            std::vector<uint8_t> taken((size_t)(e > a ? e - a : 0), 0);     // items a quad of an earlier site took along
            for (uint64_t k = a; k < e; ++k) {
                if (taken[(size_t)(k - a)]) continue;
                const uint32_t s0 = pl.items[k].x, g = pl.items[k].y;
                const SiteMap& m = pl.smap[s0];
                if (m.xrow != NONE) { u.push_back(make_uint2((uint32_t)(k - pl.item_begin), 1u)); continue; }
                // segment g of site s0 and of the up to three sites behind it in the same aligned group of four columns
                // that have the same partners (the same later x sites, so the same number of segments G): their items lie
                // G apart in the list (a site's segments are consecutive items)
                const BlockPlan& bp = pl.plans[m.block];
                const uint32_t G = (bp.nxs - m.xnext + EMIT_SEG_Q - 1u) / EMIT_SEG_Q;
                uint32_t n = 1;
                while (n < 4u && G < 4096u && k + (uint64_t)n * G < e) {
                    const uint2 it2 = pl.items[k + (uint64_t)n * G];
                    const SiteMap& m2 = pl.smap[it2.x];
                    if (it2.x != s0 + n || it2.y != g || m2.xrow != NONE || m2.block != m.block || m2.xnext != m.xnext ||
                        (m2.ycol >> 2) != (m.ycol >> 2) || m2.ycol != m.ycol + n) break;
                    ++n;
                }
                for (uint32_t j = 1; j < n; ++j) taken[(size_t)(k + (uint64_t)j * G - a)] = 1;
                u.push_back(make_uint2((uint32_t)(k - pl.item_begin), n | (1u << 16) | ((n > 1u ? G : 0u) << 20)));
            }
        }
    }

    // ---- pass 2: count tiles, block by block; a sharded run keeps the tiles its items read from
    std::vector<uint32_t> xmin, xmax, ymin, ymax, xbefore, tribefore, row_rank, tile_min_rank, trix_before;
    NeedMap need;
    for (uint64_t b = b0; b < b1; ++b) {
        const BlockPlan& bp = pl.plans[b];
        const uint32_t sb = bp.site_begin, se = bp.site_end, P = se - sb, nxs = bp.nxs;
        const uint64_t blk_item_begin = blk_items[b], blk_item_end = blk_items[b + 1];
        if (bp.nx == 0) continue;
        const uint32_t y_xpart = P - nxs;
        // which count kernel: the matrix-core kernel pays off on blocks with many columns and many reads
        // (its 128 x 128 tile has a 256-store epilogue per lane); small or shallow blocks keep the
        // VALU popcount kernel.  LGMI_COUNT_KERNEL=valu|mfma forces one of them (tests, A/B runs).
        const uint32_t block_words = (in.block_n_reads[b] + 63u) / 64u;
        bool use_mfma = bp.nx >= 96 && bp.ny >= 96 && block_words >= 32;
        if (count_kernel == 1) use_mfma = false;
        if (count_kernel >= 2) use_mfma = true;
        if (in.block_n_reads[b] >= (1u << 26)) use_mfma = false;   // the int8 kernel's accumulators hold 64 * count in 32 bits
        if (use_mfma && in.block_n_reads[b] >= (1u << 24)) part_fp4[t] = 0;
        const uint32_t edge = use_mfma ? 128u : (uint32_t)TILE;
        std::vector<Tile>& out_tiles = use_mfma ? part_mtiles[t] : part_tiles[t];
        const uint32_t ntx = (bp.nx + edge - 1) / edge, nty = (bp.ny + edge - 1) / edge;

        // x rank of every row (a pseudo row has its site's), then the smallest rank in every tile row
        row_rank.assign(bp.nx, 0);
        for (uint32_t s = sb; s < se; ++s) {
            const SiteMap& m = pl.smap[s];
            if (m.xrow != NONE) row_rank[m.xrow] = m.xnext - 1u;
            if (m.prow != NONE) row_rank[m.prow] = m.xnext - 1u;
        }
        tile_min_rank.assign(ntx, 0xFFFFFFFFu);
        for (uint32_t r = 0; r < bp.nx; ++r) tile_min_rank[r / edge] = std::min(tile_min_rank[r / edge], row_rank[r]);
        const uint32_t* const yl_x = pl.ylist.data() + bp.yl_off + y_xpart;      // x rank -> site
        if (sharded) {
            need.reset(ntx, nty, edge);
            const uint64_t a = std::max(blk_item_begin, pl.item_begin), e = std::min(blk_item_end, pl.item_end);
            if (a >= e) continue;                        // none of this block's rows belong to the shard
            xbefore.assign(P + 1, 0); tribefore.assign(P + 1, 0); trix_before.assign(nxs + 1, 0);
            for (uint32_t k = 0; k < P; ++k) {
                const bool is_x = pl.smap[sb + k].xrow != NONE;
                xbefore[k + 1] = xbefore[k] + (is_x ? 1u : 0u);
                tribefore[k + 1] = tribefore[k] + (in.tri[sb + k] ? 1u : 0u);
                if (is_x) trix_before[xbefore[k + 1]] = trix_before[xbefore[k]] + (in.tri[sb + k] ? 1u : 0u);
            }
            for (uint64_t it = a; it < e; ++it) {
                const uint32_t i = pl.items[it].x, g = pl.items[it].y;
                const SiteMap& mi = pl.smap[i];
                if (mi.xrow != NONE) {
                    const uint32_t ncand = se - 1u - i;
                    const uint32_t qa = g * EMIT_SEG, qb = std::min(ncand, (g + 1u) * EMIT_SEG);
                    const uint32_t ka = i + 1u + qa - sb, kb = i + 1u + qb - sb;       // partner sites, block-local
                    const uint32_t c0[3] = {ka - xbefore[ka], y_xpart + xbefore[ka], P + tribefore[ka]};
                    const uint32_t c1[3] = {kb - xbefore[kb], y_xpart + xbefore[kb], P + tribefore[kb]};
                    for (int part = 0; part < 3; ++part) {
                        need.mark(mi.xrow, mi.xrow + 1u, c0[part], c1[part]);
                        if (mi.prow != NONE) need.mark(mi.prow, mi.prow + 1u, c0[part], c1[part]);
                    }
                } else {
                    // its partners in this segment: the x sites of ranks [xa, xb); their rows and pseudo rows
                    const uint32_t xa = std::min(nxs, mi.xnext + g * EMIT_SEG_Q), xb = std::min(nxs, xa + EMIT_SEG_Q);
                    if (interleave) {                            // every row from the first of them to the row before the next rank's
                        const uint32_t ra = xa < nxs ? pl.smap[yl_x[xa]].xrow : bp.nx, rb = xb < nxs ? pl.smap[yl_x[xb]].xrow : bp.nx;
                        need.mark(ra, rb, mi.ycol, mi.ycol + 1u);
                        if (mi.pcol != NONE) need.mark(ra, rb, mi.pcol, mi.pcol + 1u);
                    } else {                                     // real rows [xa, xb), and the pseudo rows behind the last x site of the tri ones
                        const uint32_t pa = nxs + trix_before[xa], pb = nxs + trix_before[xb];
                        need.mark(xa, xb, mi.ycol, mi.ycol + 1u);
                        need.mark(pa, pb, mi.ycol, mi.ycol + 1u);
                        if (mi.pcol != NONE) { need.mark(xa, xb, mi.pcol, mi.pcol + 1u); need.mark(pa, pb, mi.pcol, mi.pcol + 1u); }
                    }
                }
            }
        }

        // tiles: union band per `edge`-column group of each list
        xmin.assign(ntx, 0xFFFFFFFFu); xmax.assign(ntx, 0); ymin.assign(nty, 0xFFFFFFFFu); ymax.assign(nty, 0);
        for (uint32_t r = 0; r < bp.nx; ++r) {
            const Col& c = in.cols[pl.xlist[bp.xl_off + r]];
            if (!c.nw) continue;
            xmin[r / edge] = std::min(xmin[r / edge], c.w0); xmax[r / edge] = std::max(xmax[r / edge], c.w0 + c.nw);
        }
        for (uint32_t q = 0; q < bp.ny; ++q) {
            const Col& c = in.cols[pl.ylist[bp.yl_off + q]];
            if (!c.nw) continue;
            ymin[q / edge] = std::min(ymin[q / edge], c.w0); ymax[q / edge] = std::max(ymax[q / edge], c.w0 + c.nw);
        }
        // tile order: groups of XG x-tile rows sweep the y tiles together, so that the XG tiles that
        // run side by side on an XCD (xcd_remap in the kernels) share one y tile in L2 and every y column is
        // fetched from HBM once per group instead of once per x-tile row
        uint32_t XG = use_mfma ? 4 : 8;
        if (xg_override) XG = xg_override;
        for (uint32_t tg = 0; tg < ntx; tg += XG) {
            for (uint32_t ty = 0; ty < nty; ++ty) {
                if (ymin[ty] >= ymax[ty]) continue;
                const uint32_t y0 = ty * edge, y1 = std::min(y0 + edge, bp.ny);
                for (uint32_t tx = tg; tx < std::min(tg + XG, ntx); ++tx) {
                    if (xmin[tx] >= xmax[tx]) continue;
                    const uint32_t x0 = tx * edge, x1 = std::min(x0 + edge, bp.nx);
                    // rows (an x site's and its pseudo row alike) against x site cols: only row rank < col rank is ever read
                    (void)x1;
                    if (y0 >= y_xpart && y1 <= y_xpart + nxs && tile_min_rank[tx] >= (y1 - 1 - y_xpart)) continue;
                    const uint32_t k0 = std::max(xmin[tx], ymin[ty]), k1 = std::min(xmax[tx], ymax[ty]);
                    if (k0 >= k1) continue;
                    if (sharded && !need.get(tx, ty)) continue;
                    out_tiles.push_back(Tile{(uint32_t)b, x0, y0, k0, k1});
                }
            }
        }
    }
    });
    {
        size_t tot = 0;
        for (auto& u : part_units) tot += u.size();
        pl.units.reserve(tot);
        for (auto& u : part_units) pl.units.insert(pl.units.end(), u.begin(), u.end());
    }
    for (unsigned t = 0; t < T; ++t) {
        pl.tiles.insert(pl.tiles.end(), part_tiles[t].begin(), part_tiles[t].end());
        pl.mtiles.insert(pl.mtiles.end(), part_mtiles[t].begin(), part_mtiles[t].end());
        if (!part_fp4[t]) pl.mfma_fp4 = false;
    }
    for (uint64_t s = 0; s < ns; ++s) pl.bytes_in += 16ull * in.cols[s].nw + 17ull;

    // ---- FP4 matrix-core blocks: operand groups the shard's tiles read, and where each block's groups live
    if (pl.mfma_fp4 && !pl.mtiles.empty()) {
        std::vector<uint8_t> needx, needy;
        size_t k = 0;
        while (k < pl.mtiles.size()) {
            const uint32_t b = pl.mtiles[k].block;
            BlockPlan& bp = pl.plans[b];
            const uint32_t gx = (bp.nx + 31u) / 32u, gy = (bp.ny + 31u) / 32u;
            const uint32_t words = (in.block_n_reads[b] + 63u) / 64u;
            bp.op_steps = (words + 3u) / 4u + 12u;         // + the software pipeline's read-ahead past the last step (< 8 steps)
            needx.assign(gx, 0); needy.assign(gy, 0);
            for (; k < pl.mtiles.size() && pl.mtiles[k].block == b; ++k) {
                const Tile& t = pl.mtiles[k];
                for (uint32_t g = t.x0 / 32u; g < std::min(gx, t.x0 / 32u + 4u); ++g) needx[g] = 1;
                for (uint32_t g = t.y0 / 32u; g < std::min(gy, t.y0 / 32u + 4u); ++g) needy[g] = 1;
            }
            // every group of the block has its place (a tile addresses its groups by index); only the needed ones are filled
            bp.xop_off = pl.op_total;
            pl.op_total += (uint64_t)gx * bp.op_steps * 128u;
            bp.yop_off = pl.op_total;
            pl.op_total += (uint64_t)gy * bp.op_steps * 128u;
            pl.op_max_steps = std::max(pl.op_max_steps, bp.op_steps);
            for (uint32_t g = 0; g < gx; ++g) if (needx[g]) pl.op_groups.push_back(OpGroup{b, 0u, g, 0u});
            for (uint32_t g = 0; g < gy; ++g) if (needy[g]) pl.op_groups.push_back(OpGroup{b, 1u, g, 0u});
        }
    }
}

}  // namespace lgmi
