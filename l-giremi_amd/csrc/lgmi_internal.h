// lgmi_internal.h — device data layout and kernel launchers shared by the HIP
// translation units of liblgmi.so.  Not part of the public ABI (include/lgmi.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "../../include/lgmi.h"

namespace lgmi {

// ---------------------------------------------------------------- layout in HBM
//
// A *column* is what the count kernel pairs up: a real site (coverage plane C,
// allele plane A = reads of class 2) or, for a site that has class-0 reads
// ("tri" site), an extra pseudo column (C, A = reads of class 1).  Real site s
// is column s; pseudo columns follow at n_sites .. n_cols-1.
// cplanes holds, per column, the band's words interleaved as (C_k, A_k) pairs so
// that one 16-byte load/LDS read brings both planes of 64 reads.
struct Col {
    uint64_t off;   // index (in 16-byte pairs) of the band's first word in cplanes
    uint32_t w0;    // first 64-read word of the band
    uint32_t nw;    // words in the band
};

// per-block plan entry (built on the host for every run, depends on het_only)
struct BlockPlan {
    uint64_t slot_base;  // first slot of the block's nx x ny_pad slot matrix
    uint32_t xl_off;     // offset of the block's x list in xlist[]
    uint32_t yl_off;     // offset of the block's y list in ylist[]
    uint32_t nx;         // rows: x sites in position order, a tri x site's pseudo row right behind its own
    uint32_t ny;         // cols: non-x sites, x sites, pseudo cols of tri sites
    uint32_t ny_pad;     // row stride of the slot matrix (multiple of 4)
    uint32_t nxs;        // number of real x sites (x ranks 0 .. nxs - 1)
    uint32_t site_begin; // first global site of the block
    uint32_t site_end;
    // matrix-core blocks (FP4 kernel): the operands re-laid in the order the waves load them (k_gather_ops):
    // group g of 32 consecutive x rows (y columns), step t of 4 words -> 128 uint4 at op_off + (g * op_steps + t) * 128:
    // [plane C | plane A][lane], lane = column (lane & 31) + 32 * half, the half's two words of the plane
    uint64_t xop_off;    // in uint4 units; valid when op_steps > 0
    uint64_t yop_off;
    uint32_t op_steps;   // steps laid out per group (covers the block's words + the kernel's prefetch overrun, zeros beyond)
    uint32_t op_pad;
};

// one 64 x 64 tile of a block's slot matrix and the word range it must sweep
struct Tile {
    uint32_t block;
    uint32_t x0;   // first row (index into the block's x list)
    uint32_t y0;   // first col (index into the block's y list)
    uint32_t k0;   // word range [k0, k1) — intersection of the two union bands
    uint32_t k1;
};

// per-site lookup used by the emit kernels
struct SiteMap {
    uint32_t xrow;   // row of the site in its block's slot matrix, or NONE
    uint32_t ycol;   // col of the site
    uint32_t prow;   // pseudo row (tri site that is an x site), or NONE
    uint32_t pcol;   // pseudo col (tri site), or NONE
    uint32_t xnext;  // number of real x sites of the block with index <= this site
    uint32_t block;
};
static const uint32_t NONE = 0xFFFFFFFFu;

// what lgmi_result.owner_ points to (api.cpp, comm.cpp); released by lgmi_result_free()
struct ResultOwner { virtual ~ResultOwner() {} };

// device arrays of a resident result, as comm.cpp sees them (lgmi_dresult stays private to api.cpp)
struct DResultView {
    uint64_t n_rows = 0, n_sites = 0;
    const uint32_t* i = nullptr; const uint32_t* j = nullptr; const double* mi = nullptr;
    const double* p = nullptr; const uint32_t* exceed = nullptr; const uint32_t* counts = nullptr;
    const double* mean = nullptr; const uint32_t* npairs = nullptr; const unsigned long long* sum = nullptr;
    uint32_t n_shuffles = 0;      // with p_from_exceed: row_p == (1 + row_exceed) / (n_shuffles + 1) for every row
    bool p_from_exceed = false;
    lgmi_run_info info = {};
    // the compact gather (same batch on every rank): rows by first site, candidate counts, the shard's first work item, and
    // what names a site's candidates (host tables: x sites of every block by rank, per site its first candidate)
    const uint32_t* nfirst = nullptr; const uint32_t* ncand = nullptr;
    uint32_t first_site = 0xFFFFFFFFu, first_seg = 0;
    const std::vector<uint64_t>* block_site_begin = nullptr; const std::vector<uint8_t>* site_type = nullptr; bool het_only = false;
    const void* table_owner = nullptr;      // shared_ptr<const SiteTable>* of the source result, for the gathered one to share
};

static const int TILE = 64;      // tile edge in columns
static const int KC = 8;         // 64-bit words staged per LDS stage
static const double MEAN_SCALE = 1099511627776.0;  // 2^40 fixed point for mean MI

struct DevBatch {  // what lgmi_dbatch owns on the device
    uint64_t n_blocks = 0, n_sites = 0, n_cols = 0, n_pairs16 = 0;
    int64_t*  d_pos = nullptr;
    uint8_t*  d_type = nullptr;
    uint8_t*  d_tri = nullptr;       // [n_sites] 1 when the site has class-0 reads
    Col*      d_cols = nullptr;      // [n_cols]
    ulonglong2* d_cplanes = nullptr; // [n_pairs16]
};

// ---------------------------------------------------------------- launchers
// count.hip
void launch_count(hipStream_t st, uint32_t n_tiles, const Tile* tiles, const BlockPlan* plans,
                  const uint32_t* xlist, const uint32_t* ylist, const Col* cols,
                  const ulonglong2* cplanes, uint4* slots);

// count_mfma.hip (128 x 128 tiles on the int8 matrix cores; same slot planes)
void launch_count_mfma(hipStream_t st, uint32_t n_tiles, const Tile* tiles, const BlockPlan* plans,
                       const uint32_t* xlist, const uint32_t* ylist, const Col* cols,
                       const ulonglong2* cplanes, const ulonglong2* zero_entry, uint4* slots);
void launch_count_mfma_fp4(hipStream_t st, uint32_t n_tiles, const Tile* tiles, const BlockPlan* plans,
                           const uint4* ops, uint4* slots);
// one entry per 32-column group whose operands a run needs
struct OpGroup {
    uint32_t block;
    uint32_t is_y;       // 0: group of the x list, 1: of the y list
    uint32_t group;      // rows / columns [32 * group, 32 * group + 32) of the list
    uint32_t pad;
};
void launch_gather_ops(hipStream_t st, uint32_t n_groups, uint32_t max_steps, const OpGroup* groups, const BlockPlan* plans,
                       const uint32_t* xlist, const uint32_t* ylist, const Col* cols, const ulonglong2* cplanes,
                       uint4* ops);

// emit.hip
struct EmitArgs {
    uint32_t n_sites;
    uint32_t min_common;
    int het_only;
    const BlockPlan* plans;
    const SiteMap* smap;
    const uint32_t* xsites;   // the y list (its x part, [yl_off + n_sites_of_block - nxs, + nxs), lists the x sites by rank)
    const uint32_t* xrows;    // parallel to it: the slot-matrix row of the x site of that rank
    int rows_are_ranks;       // 1: pseudo rows sit behind the last x site (unsharded plan): row of rank r == r
    const Col* cols;
    const uint8_t* type;
    const uint8_t* tri;
    const uint4* slots;       // one (N, R, C, A) per slot: common reads, class-2 reads of x, of y, of both
    // work items of the two emit passes: a site row is cut into segments of EMIT_SEG partners so that the long
    // rows of x sites (up to the whole block) spread over many waves; items are in (site, segment) order
    uint32_t n_items;
    const uint2* items;       // [n_items] (site, segment)
    uint32_t n_units;         // one wave per unit: an x site's item, or up to four other sites sharing a line of slots
    const uint2* units;       // [n_units] (first item, n items | kind << 16 | stride << 20); kind 1 = quad: segment g of up to four
                              //           sites whose items lie `stride` apart (each site's segments are consecutive items)
    uint32_t* row_cnt;        // [n_items]   pass 1 out
    const uint64_t* row_start;// [n_items+1] pass 2 in
    uint32_t* out_i; uint32_t* out_j; double* out_mi; uint32_t* out_counts; // pass 2 out
    // what the permutation stage needs of a row, 16 bytes instead of its 36-byte table (NULL: no p-values):
    //   (N, K, n, kind << 30 | k_obs)   kind 1: at most one non-empty class on a side (p = 1)
    //                                   kind 2: 2 x 2 non-empty, the hypergeometric (N, K, n) and the observed count
    //                                   kind 3: larger — the only rows whose table is written when counts_sparse
    uint4* out_rec;
    int counts_sparse;        // out_counts is scratch for the permutation stage only: tables of kind-3 rows, nothing else
    unsigned long long* site_sum; uint32_t* site_cnt;  // [n_sites] fixed-point sums
    int* err_flag;            // set to 1 when a pair with N == 0 reaches the MI
    unsigned long long* unit_words;  // [n_units] pass 1: per unit, the sum over its examined pairs of overlapping words
};
static const uint32_t EMIT_SEG = LGMI_EMIT_SEG;   // multiple of 64
static const uint32_t EMIT_SEG_Q = LGMI_EMIT_SEG_Q;   // multiple of 16 (a quad trip covers 16 x ranks)
void launch_emit_count(hipStream_t st, const EmitArgs& a);
size_t scan_tmp_words(uint32_t n);
void launch_scan(hipStream_t st, const uint32_t* cnt, uint64_t* start, uint32_t n, uint64_t* tmp);
void launch_sum_u64(hipStream_t st, const unsigned long long* v, uint32_t n, unsigned long long* out);
void launch_emit_write(hipStream_t st, const EmitArgs& a);
void launch_site_mean(hipStream_t st, uint32_t n_sites, const unsigned long long* sum,
                      const uint32_t* cnt, double* mean);
void launch_sites_add(hipStream_t st, uint32_t n_sites, unsigned long long* sum, const unsigned long long* sum_part,
                      uint32_t* cnt, const uint32_t* cnt_part);
void launch_rows_mean(hipStream_t st, uint64_t n_rows, const uint32_t* ri, const uint32_t* rj,
                      const double* mi, unsigned long long* sum, uint32_t* cnt);
// the compact row form (ABI 6): per-site row counts by first site / candidate counts, full flags + offsets, listed partners
void launch_site_rows(hipStream_t st, uint32_t n_items, const uint2* items, const uint32_t* row_cnt, uint32_t n_sites,
                      const SiteMap* smap, const BlockPlan* plans, uint32_t* n_first, uint32_t* n_cand);
void launch_compact_sites(hipStream_t st, uint32_t n_sites, const uint32_t* n_first, const uint32_t* n_cand, uint8_t* full,
                          uint32_t* listed_tmp, uint64_t* row_begin, uint64_t* list_begin, uint64_t* scan_tmp);
void launch_list_partners(hipStream_t st, uint32_t n_sites, const uint8_t* full, const uint64_t* row_begin, const uint64_t* list_begin,
                          const uint32_t* row_j, uint32_t* out);
void launch_narrow_u16(hipStream_t st, uint64_t n, const uint32_t* in, uint16_t* out);
void launch_add_u32(hipStream_t st, uint32_t n, uint32_t* acc, const uint32_t* part);

// perm.hip
struct PermArgs {
    const uint64_t* n_rows_dev;   // number of rows, on the device (the host may not know it yet)
    uint64_t max_rows;            // upper bound the grids are sized by
    const uint32_t* row_i; const uint32_t* row_j; const uint32_t* counts;
    const uint4* rec;             // per row (N, K, n, kind << 30 | k_obs), written by k_emit<2> (EmitArgs::out_rec)
    const long long* G; const double* LF;
    uint32_t n_shuffles; uint64_t seed;
    uint32_t site_base;           // added to row_i / row_j in the Philox counters (lgmi_params.stream_site_base)
    int exact_2x2;                // rows with at most 2 x 2 non-empty classes get the exact p, not a binomial draw
    uint32_t enum_max;            // larger tables with at most this many candidate tables: exact mass by enumeration (set by launch_perm_general)
    uint32_t six_pts;             // six-cell tables: lattice points per shuffle a row may cost on the exact path, 0 = off (set by launch_perm_general)
    double* out_p; uint32_t* out_exceed;   // out_p may be NULL (lgmi_params.no_row_p: p is a function of exceed)
    uint32_t* gen_list; unsigned int* gen_count;   // gen_count[0] rows queued by k_perm_fast, [1] next row of k_perm_general, [2] rows k_perm_enum
                                                   // leaves to k_perm_general, listed at gen_list + gen_count[0], [3] some row is enumerable,
                                                   // [4] rows k_perm_six leaves (third list), [5] its next row, [6] rows it finished (all zero at launch)
};
void launch_perm_fast(hipStream_t st, const PermArgs& a);
void launch_perm_general(hipStream_t st, const PermArgs& a, hipEvent_t after_exact = nullptr);   // k_perm_enum, k_perm_six, [event], k_perm_general
void launch_selftest_log(hipStream_t st, uint64_t n, const double* x, double* out);
void launch_selftest_le_exp(hipStream_t st, uint64_t n, const double* x2, const double* t, uint8_t* fast, uint8_t* det,
                            double* e_hw, double* e_det);

// ecdf.hip
size_t ecdf_sort_temp_bytes(uint32_t n);
hipError_t launch_ecdf(hipStream_t st, uint32_t n_ref, const double* ref, double* sorted, void* temp,
                       size_t temp_bytes, uint64_t n_query, const double* query, double* out);

// synth.hip: layout prep for uploaded batches and the dense synthetic generator
void launch_tri_flags(hipStream_t st, uint32_t n_sites, const uint32_t* site_nw,
                      const uint64_t* site_plane_off, const uint64_t* planes, uint8_t* tri);
// columns [c0, c0 + n_cols) (real: site c; pseudo: col_site[c - n_sites]) <- (C, A) pairs from lo/hi planes
void launch_prep_cols(hipStream_t st, uint32_t c0, uint32_t n_cols, uint32_t n_sites, const Col* cols,
                      const uint32_t* pseudo_site, const uint64_t* site_plane_off,
                      const uint64_t* planes, ulonglong2* cplanes);
void launch_flags_differ(hipStream_t st, uint32_t n, const uint8_t* a, const uint8_t* b, int* out);
void launch_synth_depth(hipStream_t st, const lgmi_synth_spec& sp, uint32_t W, uint32_t* depth3);
void launch_synth_write(hipStream_t st, const lgmi_synth_spec& sp, uint32_t W, const uint32_t* depth3,
                        const uint32_t* pseudo_of_site, ulonglong2* cplanes, uint32_t site_base);

}  // namespace lgmi
