// plan_fuzz.cpp — the host planner (l-giremi_amd/csrc/plan.cpp) on random batches under AddressSanitizer / UBSan (or
// ThreadSanitizer): built and run by tools/asan_plan.sh, CPU only.  Besides memory safety it checks what the run relies
// on: a shard's units cover its items exactly once, every tile lies inside its block's slot matrix, the shards' item
// ranges tile the whole list, and every slot a shard's items read lies in one of its tiles (brute force).
#include <cstdio>
#include <cstdlib>
#include <random>
#include <set>

#include "../../l-giremi_amd/csrc/plan.h"

using namespace lgmi;

struct Batch {
    std::vector<uint64_t> bsb;
    std::vector<uint32_t> reads;
    std::vector<uint8_t> type, tri;
    std::vector<Col> cols;
    std::vector<uint32_t> pseudo;
    PlanInput in() const {
        PlanInput p;
        p.n_blocks = reads.size(); p.n_sites = type.size(); p.block_site_begin = bsb.data(); p.block_n_reads = reads.data();
        p.type = type.data(); p.tri = tri.data(); p.cols = cols.data(); p.pseudo_of_site = pseudo.data();
        return p;
    }
};

static Batch make(std::mt19937_64& g, int shape) {
    Batch b;
    auto U = [&](uint64_t lo, uint64_t hi) { return lo + g() % (hi - lo + 1); };
    uint64_t nb = shape == 0 ? U(1, 3) : shape == 1 ? U(200, 900) : U(1, 40);
    if (shape == 3) nb = 1;
    b.bsb.push_back(0);
    const unsigned het_pct = (unsigned)U(0, 100), tri_pm = (unsigned)U(0, 300);
    uint64_t off = 0;
    for (uint64_t k = 0; k < nb; ++k) {
        uint64_t P = shape == 0 ? U(0, 2600) : shape == 1 ? U(0, 40) : U(0, 400);
        uint32_t R = (uint32_t)(shape == 1 ? U(1, 300) : U(1, 20000));
        if (shape == 3) { P = 1500; R = 9000; }                    // tests/test_gpu_shard.py: the split-under-a-budget case
        const uint32_t W = (R + 63) / 64;
        for (uint64_t s = 0; s < P; ++s) {
            b.type.push_back(U(0, 99) < het_pct ? (uint8_t)LGMI_TYPE_HET_SNP : (uint8_t)U(0, 3));
            b.tri.push_back(U(0, 999) < tri_pm);
            const uint32_t w0 = shape == 2 ? (uint32_t)U(0, W - 1) : 0, nw = shape == 2 ? (uint32_t)U(0, W - w0) : W;
            b.cols.push_back(Col{off, w0, nw});
            off += nw;
        }
        b.reads.push_back(R);
        b.bsb.push_back(b.type.size());
    }
    b.pseudo.assign(b.type.size(), NONE);
    for (size_t s = 0; s < b.type.size(); ++s)
        if (b.tri[s]) { b.pseudo[s] = (uint32_t)b.cols.size(); b.cols.push_back(Col{off, b.cols[s].w0, b.cols[s].nw}); off += b.cols[s].nw; }
    return b;
}

#define CHECK(c, ...) do { if (!(c)) { fprintf(stderr, "plan_fuzz: %s:%d: %s — ", __FILE__, __LINE__, #c); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); abort(); } } while (0)

static void check_plan(const Batch& b, const Plan& pl, bool het_only, bool sharded, bool brute) {
    const uint64_t n_mine = pl.item_end - pl.item_begin;
    CHECK(pl.item_begin <= pl.item_end && pl.item_end <= pl.items.size(), "item range");
    std::vector<uint8_t> seen(n_mine, 0);
    for (const uint2& u : pl.units) {
        const uint32_t n = u.y & 0xFFFFu, kind = (u.y >> 16) & 0xFu, G = u.y >> 20;
        CHECK(n >= 1 && (kind == 0 ? n == 1 : n <= 4), "unit of %u items, kind %u", n, kind);
        for (uint32_t j = 0; j < n; ++j) {
            const uint64_t k = (uint64_t)u.x + (uint64_t)j * (kind ? (n > 1 ? G : 0) : 1);
            CHECK(k < n_mine, "unit item %llu of %llu", (unsigned long long)k, (unsigned long long)n_mine);
            CHECK(!seen[k], "item in two units");
            seen[k] = 1;
        }
    }
    for (uint64_t k = 0; k < n_mine; ++k) CHECK(seen[k], "item %llu in no unit", (unsigned long long)k);
    std::vector<std::set<std::pair<uint32_t, uint32_t>>> have(brute ? pl.plans.size() : 0);
    for (int kind = 0; kind < 2; ++kind)
        for (const Tile& t : kind ? pl.mtiles : pl.tiles) {
            CHECK(t.block < pl.plans.size(), "tile block");
            const BlockPlan& bp = pl.plans[t.block];
            const uint32_t edge = kind ? 128u : (uint32_t)TILE;
            CHECK(t.x0 < bp.nx && t.y0 < bp.ny && t.x0 % edge == 0 && t.y0 % edge == 0 && t.k0 < t.k1 && t.k1 <= (b.reads[t.block] + 63u) / 64u,
                  "tile (%u, %u) of a %u x %u block", t.x0, t.y0, bp.nx, bp.ny);
            if (brute) for (uint32_t dx = 0; dx < edge; dx += 64) for (uint32_t dy = 0; dy < edge; dy += 64) have[t.block].insert({(t.x0 + dx) / 64, (t.y0 + dy) / 64});
        }
    for (const OpGroup& og : pl.op_groups) CHECK(og.block < pl.plans.size(), "operand group block");
    if (!brute) return;
    // every slot the shard's items read (rows xrow / prow of the x partner, columns ycol / pcol of the other) is in a tile,
    // unless the band intersection is empty (no tile is made then: the slot's count is zero by construction)
    for (uint64_t it = pl.item_begin; it < pl.item_end; ++it) {
        const uint32_t i = pl.items[it].x, g = pl.items[it].y;
        const SiteMap& mi = pl.smap[i];
        const BlockPlan& bp = pl.plans[mi.block];
        const bool is_x = mi.xrow != NONE;
        const uint32_t seg = is_x ? EMIT_SEG : EMIT_SEG_Q;
        const uint32_t ncand = is_x ? bp.site_end - 1 - i : bp.nxs - mi.xnext;
        for (uint32_t q = g * seg; q < std::min(ncand, (g + 1) * seg); ++q) {
            uint32_t j;
            if (is_x) j = i + 1 + q;
            else j = pl.ylist[bp.yl_off + (bp.site_end - bp.site_begin - bp.nxs) + mi.xnext + q];
            const SiteMap& mj = pl.smap[j];
            // the x site of the pair gives the rows; when both are x sites the earlier one does
            const SiteMap& mx = is_x ? mi : mj;
            const SiteMap& my = is_x ? mj : mi;
            const uint32_t rows[2] = {mx.xrow, mx.prow}, colsv[2] = {my.ycol, my.pcol};
            const Col &cx = b.cols[is_x ? i : j], &cy = b.cols[is_x ? j : i];
            const bool overlap = std::max(cx.w0, cy.w0) < std::min(cx.w0 + cx.nw, cy.w0 + cy.nw);
            if (!overlap) continue;
            for (uint32_t r : rows) for (uint32_t c : colsv) {
                if (r == NONE || c == NONE) continue;
                CHECK(r < bp.nx && c < bp.ny, "slot (%u, %u) outside %u x %u", r, c, bp.nx, bp.ny);
                CHECK(have[mi.block].count({r / 64, c / 64}), "slot (%u, %u) of pair (%u, %u) in no tile (sharded %d, het_only %d)", r, c, i, j, (int)sharded, (int)het_only);
            }
        }
    }
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 60;
    const uint64_t seed0 = argc > 2 ? strtoull(argv[2], nullptr, 10) : 1;
    uint64_t n_plans = 0;
    for (int r = 0; r < rounds; ++r) {
        std::mt19937_64 g(seed0 * 7919 + r);
        const int shape = r % 4;
        const Batch b = make(g, shape);
        const PlanInput in = b.in();
        for (int het = 0; het < 2; ++het)
            for (uint32_t world : {1u, 2u, 3u, 5u, 8u}) {
                const uint32_t ns = (g() & 1) ? 0u : (uint32_t)(g() % 2000);
                const int ck = shape == 2 ? (int)(g() % 3) : 0;
                uint64_t covered = 0, examined = 0, total = 0;
                for (uint32_t rank = 0; rank < world; ++rank) {
                    Plan pl;
                    build_plan(in, het != 0, rank, world, ck, 0, ns, pl);
                    ++n_plans;
                    CHECK(pl.item_begin == covered, "shard %u of %u starts at %llu, the last one ended at %llu", rank, world,
                          (unsigned long long)pl.item_begin, (unsigned long long)covered);
                    covered = pl.item_end;
                    examined += pl.n_examined; total = pl.n_examined_total;
                    check_plan(b, pl, het != 0, world > 1, b.type.size() <= 3000);
                    if (rank + 1 == world) CHECK(pl.item_end == pl.items.size(), "last shard ends early");
                }
                CHECK(examined == total, "shards examine %llu pairs of %llu", (unsigned long long)examined, (unsigned long long)total);
            }
    }
    printf("plan_fuzz: %d batches, %llu plans: ok\n", rounds, (unsigned long long)n_plans);
    return 0;
}
