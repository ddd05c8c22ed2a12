/*
 * lgmi.h — C ABI of liblgmi.so: the MI355X (gfx950) pairwise mutual-information
 * engine that replaces L-GIREMI's site-pair MI step.
 *
 * What each entry point replaces in the reference (gxiaolab/L-GIREMI v0.2.4,
 * pure Python, paths relative to the reference root):
 *
 *   lgmi_run / lgmi_run_device
 *       src/giremi/mutual_information.py:6-45   mismatch_pair_mutual_info()
 *           (pair enumeration :10-12, common reads :15-20, allele->class :25-40,
 *            sklearn.metrics.mutual_info_score :41, row assembly :42-45)
 *       src/giremi/mismatch.py:392-396          het_snp pair filter (het_only=1)
 *       src/giremi/mutual_information.py:48-60  mean_mismatch_pair_mutual_info()
 *           (per-site mean of the kept rows; site_mean_mi / site_n_pairs)
 *   lgmi_site_mean
 *       src/giremi/mutual_information.py:48-60  for caller-supplied rows
 *   lgmi_ecdf
 *       src/giremi/stat.py:7-29 (ecdf), as used for `mip` at script/giremi.py:415-429
 *   permutation p-value (row_p / row_exceed)
 *       no reference counterpart (BASELINE.json north_star asks for it; the
 *       reference has no permutation test) — specified in DESIGN.md §5 and
 *       restated on the CPU in oracle/lgmi_oracle.c.
 *
 * The reference has no FFI of its own (it is pure Python); the binding a
 * maintainer would add is the ctypes stub shown in INTEGRATION.md.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes only; no C++/torch types.
 *   - every function returns 0 on success or a negative LGMI_E_* code; the
 *     message is available from lgmi_last_error() (thread-local).
 *   - inputs are borrowed (never written, must stay alive for the call);
 *     results are library-owned and released by the matching *_free().
 *   - the library never initialises HIP at load time (dlopen may precede a
 *     fork); HIP is first touched in lgmi_device_count()/lgmi_ctx_create().
 *   - there is no CPU fallback: without a usable gfx950 device
 *     lgmi_ctx_create() fails with LGMI_E_NODEV.
 */
#ifndef LGMI_H
#define LGMI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LGMI_ABI_VERSION 6

/* error codes */
#define LGMI_OK        0
#define LGMI_E_ARG    -1   /* bad argument / malformed batch            */
#define LGMI_E_OOM    -2   /* host or device allocation failed          */
#define LGMI_E_HIP    -3   /* HIP runtime error                         */
#define LGMI_E_RCCL   -4   /* RCCL error                                */
#define LGMI_E_NODEV  -5   /* no usable GPU                             */
#define LGMI_E_STATE  -6   /* call made in the wrong state              */
#define LGMI_E_DOMAIN -7   /* a pair with 0 common reads reached the MI
                              (min_common == 0): the reference raises
                              ValueError('math domain error') there     */

/* site types (mismatch.py:29-44, :326-340) */
#define LGMI_TYPE_MISMATCH 0
#define LGMI_TYPE_SNP      1
#define LGMI_TYPE_HET_SNP  2

/* 2-bit allele-class code of one read at one site, stored as two bit planes
 * (bit r%64 of word r/64):   hi lo
 *    not covered              0  0
 *    class 1 (minor allele)   0  1      mutual_information.py:35,38
 *    class 2 (major allele)   1  0      mutual_information.py:36,39
 *    class 0 (any other)      1  1      defaultdict(int), :34,:37          */

/*
 * lgmi_batch — the packed read x site allele matrix ("LGB" blocks).
 * A block is one (footprint, strand): pairs never cross blocks
 * (mismatch.py:387-391). Sites of a block are sorted by position; reads of a
 * block are numbered 0..block_n_reads-1; each site stores only the band of
 * 64-read words [word_off, word_off+n_words) that can contain covered reads.
 * planes[plane_off .. plane_off+n_words)            = lo plane of the band
 * planes[plane_off+n_words .. plane_off+2*n_words)  = hi plane of the band
 * Bits beyond block_n_reads in the last word must be zero.
 */
typedef struct lgmi_batch {
    uint64_t n_blocks;
    uint64_t n_sites;                 /* == block_site_begin[n_blocks]        */
    uint64_t n_plane_words;           /* length of planes[] in 64-bit words   */
    const uint64_t* block_site_begin; /* [n_blocks+1] first site of each block */
    const uint32_t* block_n_reads;    /* [n_blocks]                            */
    const int64_t*  site_pos;         /* [n_sites] strictly increasing / block */
    const uint8_t*  site_type;        /* [n_sites] LGMI_TYPE_*                 */
    const uint32_t* site_word_off;    /* [n_sites]                             */
    const uint32_t* site_n_words;     /* [n_sites]                             */
    const uint64_t* site_plane_off;   /* [n_sites]                             */
    const uint64_t* planes;           /* [n_plane_words]                       */
    /* ABI 6, optional (NULL: the library finds out on the device, after the upload): site_tri[s] = 1 when site s has a
     * read of class 0 (some bit set in lo & hi), else 0 — a packer knows this for free.  With it lgmi_run() can lay out
     * and plan the run before the planes have moved, and uploads them in pieces of sites with the count kernels of the
     * pairs whose both sites have arrived running underneath (the planes must then be packed in site order:
     * site_plane_off[s + 1] == site_plane_off[s] + 2 * site_n_words[s]; otherwise the plain upload is used).  The flags
     * are checked against the planes on the device: a wrong one is LGMI_E_ARG, never a wrong result. */
    const uint8_t*  site_tri;         /* [n_sites] or NULL                     */
} lgmi_batch;

typedef struct lgmi_params {
    uint32_t min_common;   /* mutual_information.py:19 (library default 5, CLI 6) */
    uint32_t n_shuffles;   /* S <= 2^24; 0 = no permutation p (row_p = NaN)       */
    uint64_t seed;         /* Philox key for the permutation draws                */
    uint8_t  het_only;     /* 1: only pairs with >=1 het_snp side (mismatch.py:392-396);
                              0: all P(P-1)/2 pairs (mutual_information.py:12)   */
    uint8_t  emit_counts;  /* 1: fill row_counts (3x3 table per row)              */
    uint8_t  exact_2x2;    /* 1: rows whose permutation p has an exact form within reach get the EXACT p (the
                              hypergeometric mass of the tables with MI >= observed) instead of a Monte-Carlo
                              estimate; their row_exceed is LGMI_EXCEED_EXACT.  These are: tables with at most 2
                              non-empty classes per site (summed from log-factorials); since ABI 5 also larger
                              tables with few candidate tables (enumeration) and 3 x 2 / 2 x 3 tables whose region
                              {MI < observed} has at most 2^20 chords (perimeter walk).  The other
                              larger tables keep the n_shuffles estimate (row_p NaN, row_exceed LGMI_EXCEED_EXACT
                              when n_shuffles == 0).
                              ACCURACY of an "exact" p: masses are summed in 2^-62 fixed point from a log-factorial
                              table, |dp| <= 1e-9 up to 2e5 reads (5e-8 at 3e6).  A tail that is summed directly (2 x 2
                              tail form, enumeration) keeps that as a RELATIVE figure down to ~1e-18; the 2 x 2 centre form
                              and the 3 x 2 / 2 x 3 walk return 1 - (mass inside), an ABSOLUTE 1e-9: read a p below ~1e-8
                              from them as "< 1e-8".  A 3 x 2 / 2 x 3 row whose whole tail is bounded below 2^-33 before any
                              sum is made returns that BOUND (an upper bound of its p, < 9.3e-11, at least 2.2e-308) — an
                              exact p is never 0.0: the observed table is in its own tail */
    uint8_t  no_row_p;     /* 1: when every row_p is a function of its row_exceed — Monte-Carlo estimates,
                              p = (1 + row_exceed) / (n_shuffles + 1), i.e. n_shuffles > 0 without exact_2x2 — the
                              p array is not made at all: lgmi_result.row_p is NULL, row_p_derived is 1 and the
                              caller derives what it needs (8 of 28 bytes per row less on the way to the host, no
                              8-byte store per row in the permutation kernels).  0: row_p is always an array */
    /* tile-level sharding of ONE batch over several GPUs (the reference's analogue is the chunked
     * Pool.map of script/giremi.py:367-394): the result rows, in reference order, are cut into shard_world
     * contiguous, cost-balanced ranges (cost model: csrc/plan.cpp); this call computes range shard_rank only (count tiles that feed it,
     * its rows, their p-values, its share of the per-site sums).  Concatenating the shards' rows in rank
     * order gives exactly the unsharded rows.  0/0 or x/1 = unsharded. */
    uint16_t shard_rank;
    uint16_t shard_world;
    /* added to both site indices of a pair in the Philox counters of its permutation draws (the counters are keyed by
     * the pair, so that shards and launch geometry never change a draw).  A multi-GPU host whose ranks run DIFFERENT
     * batches (footprints dealt to ranks, lgmi.cli --gpus) passes each rank's site base — the same number it passes to
     * lgmi_comm_gather — and gets, pair for pair, the draws of the single batch that holds all the footprints.  0 else. */
    uint32_t stream_site_base;
    uint8_t  compact_rows; /* ABI 6.  1: lgmi_run() returns the rows in the COMPACT form described at lgmi_result (row_i and
                              row_j are not shipped: 10 instead of 20 bytes per row of a dense block cross PCIe, and the
                              permutation counts travel in pieces while the permutation stage is still running).  Ignored by
                              lgmi_run_device(): a resident result holds both forms, lgmi_dresult_fetch() /
                              lgmi_dresult_fetch_compact() choose */
    uint8_t  reserved1[3]; /* must be 0 */
} lgmi_params;
#define LGMI_EXCEED_EXACT 0xFFFFFFFFu

/*
 * lgmi_result — rows in reference order: block, then combinations() order of
 * the sorted positions (i < j). Site indices are GLOBAL (into the batch's site
 * arrays). row_counts[9*r + 3*a + b] = #common reads with class a at site i
 * and class b at site j (a, b in 0,1,2). site_mean_mi[s] is NaN and
 * site_n_pairs[s] 0 for a site that appears in no row.  In a sharded run (shard_world > 1) the per-site
 * figures cover the shard's own rows only; lgmi_comm_gather() adds the shards' integer sums up.
 */
typedef struct lgmi_result {
    uint64_t n_rows;
    uint64_t n_sites;
    const uint32_t* row_i;
    const uint32_t* row_j;
    const double*   row_mi;
    const double*   row_p;        /* NULL when n_shuffles == 0 (and no exact_2x2), or when row_p_derived */
    const uint32_t* row_exceed;   /* NULL when n_shuffles == 0 */
    const uint32_t* row_counts;   /* NULL unless emit_counts   */
    const double*   site_mean_mi; /* [n_sites] */
    const uint32_t* site_n_pairs; /* [n_sites] */
    void* owner_;                 /* private */
    uint32_t n_shuffles;          /* of the run that made the rows */
    uint32_t row_p_derived;       /* 1: row_p is NULL because lgmi_params.no_row_p asked for that:
                                     p[r] = (1 + row_exceed[r]) / (n_shuffles + 1) */
    /* ---- ABI 6: the COMPACT row form (lgmi_params.compact_rows with lgmi_run(), or lgmi_dresult_fetch_compact()).
     * Rows are in reference order (mutual_information.py:10-12, :42-45): all rows of first site s are consecutive, so
     * row_i is a run-length code — rows [row_begin[s], row_begin[s + 1]) have row_i == s — and the partners of a site
     * whose every candidate pair was emitted are the candidates themselves, in order:
     *     candidates of s = the later sites of its block: all of them when het_only == 0 or s is a het_snp site, else the
     *     later het_snp sites (mismatch.py:392-396).
     * compact == 1: row_i == row_j == NULL;
     *     site_row_full[s] == 1: the k-th row of s has row_j = the k-th candidate of s (nothing is stored);
     *     site_row_full[s] == 0: its row_j are listed, in row order, in row_j_listed (sites in increasing s);
     *     row_exceed16 (row_exceed == NULL) when every count fits 16 bits: n_shuffles <= 65535 and no exact p.
     * lgmi_result_expand_rows() writes the plain row_i / row_j arrays from these. */
    uint32_t compact;
    uint32_t reserved3;               /* 0 */
    const uint64_t* row_begin;        /* [n_sites + 1] */
    const uint8_t*  site_row_full;    /* [n_sites] */
    const uint32_t* row_j_listed;     /* [n_row_j_listed] */
    uint64_t        n_row_j_listed;
    const uint16_t* row_exceed16;     /* [n_rows] or NULL */
} lgmi_result;

/* figures of one run: work done and HIP-event time of each stage (ms), taken
 * on the stream the kernels were launched on */
typedef struct lgmi_run_info {
    uint64_t n_rows;        /* emitted pairs M                                     */
    uint64_t n_examined;    /* examined pairs E (het-involved, or all)   */
    uint64_t n_tile_pairs;  /* pairs inside the tiles the count kernel computed    */
    uint64_t word_pairs;    /* sum over examined pairs of overlapping 64-bit words */
    uint64_t bytes_in;      /* algorithmic input bytes (planes once + site meta)   */
    uint64_t bytes_out;     /* algorithmic output bytes (rows)                     */
    float ms_total;
    float ms_prep;          /* band/class prep + tile list                         */
    float ms_count;         /* pair co-occurrence count kernel                     */
    float ms_emit;          /* validity scan + MI + ordered row emission           */
    float ms_perm;          /* permutation p-values                                */
    float ms_mean;          /* per-site mean MI                                    */
    uint32_t n_count_launches;
    uint32_t n_mfma_tiles;  /* 128 x 128 tiles computed on the matrix cores (0: VALU popcount only)      */
    uint32_t mfma_dtype;    /* operand type of those tiles: 0 none, 1 int8, 2 fp4 (e2m1)                 */
    uint32_t n_six_rows;    /* of n_general_rows: 3 x 2 / 2 x 3 rows whose exact tail mass came from the perimeter walk
                               (k_perm_six) — one binomial variate each instead of n_shuffles table draws  */
    uint64_t n_examined_total; /* examined pairs of the whole batch (== n_examined when unsharded)         */
    uint64_t n_general_rows;   /* rows whose table is larger than 2 x 2: these get n_shuffles real table
                                  draws each; 2 x 2 rows get ONE binomial variate            */
    float ms_plan_host;        /* host wall time of the per-run planning (tile list, work items), ms       */
    float ms_perm_fast;        /* k_perm_fast: classification + exact 2 x 2 tails + binomial draws         */
    float ms_perm_general;     /* k_perm_general: Monte-Carlo table draws of the larger tables             */
    uint32_t n_seq_shards;     /* > 1: lgmi_run_device ran the batch as that many sequential shards because one
                                  launch sequence would not fit the memory budget (LGMI_MEM_BUDGET_MB, default 70 % of
                                  the device memory) or p-values were asked for 2^32 candidate rows or more; the
                                  stage times are sums over the shards.  0 / 1: one sequence                  */
    float ms_perm_exact;       /* of ms_perm_general: k_perm_enum + k_perm_six, the larger-than-2x2 rows whose tail mass is exact
                                  (enumeration; perimeter walk of 3 x 2 / 2 x 3 tables); the rest is k_perm_general's sampling    */
    uint32_t reserved2;
} lgmi_run_info;

/* device-side synthetic chromosome generator (SURVEY 8d "dense" regime):
 * one block of n_sites x n_reads, every read spans every site, per-(site,read)
 * dropout, haplotype-linked het SNPs every het_every-th site, independent
 * mismatch sites elsewhere, a third allele on tri_per_1024/1024 of the sites. */
typedef struct lgmi_synth_spec {
    uint64_t seed;
    uint32_t n_sites;
    uint32_t n_reads;
    uint32_t het_every;      /* site s is het_snp iff s % het_every == 0 (5)      */
    uint32_t dropout_u16;    /* P(read does not cover site) * 65536   (6554)       */
    uint32_t het_noise_u16;  /* P(het allele != haplotype) * 65536    (1311)       */
    uint32_t tri_per_1024;   /* sites with a third allele, per 1024   (20)         */
    uint32_t tri_frac_u16;   /* P(third allele | covered) * 65536     (3277)       */
    uint32_t snp_per_1024;   /* non-het sites typed 'snp', per 1024   (10)         */
    uint32_t n_blocks;       /* 0 or 1: one chromosome; k > 1: k chromosomes (blocks) of n_sites x n_reads each in one
                                batch, block c drawn with seed + c (BASELINE.json configs[2]: 22 chromosomes)          */
    uint32_t reserved;       /* must be 0 */
} lgmi_synth_spec;

typedef struct lgmi_ctx     lgmi_ctx;     /* one device, one stream, workspace */
typedef struct lgmi_dbatch  lgmi_dbatch;  /* a batch resident in HBM           */
typedef struct lgmi_dresult lgmi_dresult; /* result rows resident in HBM       */

int         lgmi_abi_version(void);
const char* lgmi_last_error(void);
/* sizeof the library's own lgmi_batch (which = 0), lgmi_params (1), lgmi_result (2), lgmi_run_info (3), lgmi_synth_spec (4),
 * lgmi_shard_plan (5), lgmi_gather_opts (6), lgmi_comm_info_t (7); 0 for any other `which`.  A binding checks its struct
 * declarations against these before its first call: a caller built against an older header would otherwise have the
 * library read or write past its structs (ABI 5; INTEGRATION.md 4 shows the check). */
size_t      lgmi_struct_size(int which);
int         lgmi_device_count(int* out_count);

int  lgmi_ctx_create(int device_id, lgmi_ctx** out);
void lgmi_ctx_destroy(lgmi_ctx* ctx);

/* host -> HBM (validates the batch) */
int  lgmi_batch_upload(lgmi_ctx* ctx, const lgmi_batch* batch, lgmi_dbatch** out);
/* synthetic dense chromosome generated directly in HBM */
int  lgmi_synth_dense(lgmi_ctx* ctx, const lgmi_synth_spec* spec, lgmi_dbatch** out);
/* HBM -> host copy of a resident batch; arrays are owned by the dbatch and
 * stay valid until lgmi_dbatch_free() */
int  lgmi_dbatch_download(lgmi_dbatch* db, lgmi_batch* out);
void lgmi_dbatch_free(lgmi_dbatch* db);

/* the hot path on a resident batch; rows stay in HBM */
int  lgmi_run_device(lgmi_ctx* ctx, const lgmi_dbatch* db, const lgmi_params* prm,
                     lgmi_dresult** out);
/* the same in two calls: lgmi_run_device_rows() stops when the rows (i, j, mi, tables, per-site means) are final,
 * lgmi_dresult_permute() runs the permutation stage on them (row_p / row_exceed are undefined in between).  A
 * multi-GPU host starts gathering the rows between the two (lgmi_comm_gather_begin) so that the transfer runs
 * under the permutation kernels. */
int  lgmi_run_device_rows(lgmi_ctx* ctx, const lgmi_dbatch* db, const lgmi_params* prm, lgmi_dresult** out);
int  lgmi_dresult_permute(lgmi_ctx* ctx, lgmi_dresult* dr);
int  lgmi_dresult_info(const lgmi_dresult* dr, lgmi_run_info* out);
/* raw device pointers of a resident result (for RCCL / zero-copy consumers) */
int  lgmi_dresult_device_ptrs(const lgmi_dresult* dr, lgmi_result* out_device_view);
/* HBM -> host; the host arrays live until lgmi_result_free(out) */
int  lgmi_dresult_fetch(lgmi_dresult* dr, lgmi_result* out);
/* the same in the compact row form (lgmi_result.compact == 1): what crosses PCIe is row_mi, the 16- or 32-bit permutation
 * counts, the listed partners of the sites that are not full, and three per-site arrays.  Not for results gathered from
 * ranks that ran different batches (LGMI_E_STATE): the root does not hold the other ranks' site tables */
int  lgmi_dresult_fetch_compact(lgmi_dresult* dr, lgmi_result* out);
/* compact form -> the plain arrays: row_i_out / row_j_out [n_rows] (either may be NULL), written on the host by the
 * library's thread team from row_begin, site_row_full, row_j_listed and the site table the result kept.  A result that
 * is not compact is copied.  (The reference's rows carry both positions: mutual_information.py:42-45.) */
int  lgmi_result_expand_rows(const lgmi_result* res, uint32_t* row_i_out, uint32_t* row_j_out);
void lgmi_dresult_free(lgmi_dresult* dr);

/* upload + run + fetch in one call (what the Python drop-ins use) */
int  lgmi_run(lgmi_ctx* ctx, const lgmi_batch* batch, const lgmi_params* prm,
              lgmi_result* out, lgmi_run_info* info_or_null);
void lgmi_result_free(lgmi_result* res);

/* per-site mean of caller-supplied rows (mutual_information.py:48-60):
 * mean_out[s] = mean of mi[r] over rows with row_i[r]==s or row_j[r]==s,
 * NaN / n_out 0 when the site is in no row. */
int  lgmi_site_mean(lgmi_ctx* ctx, uint64_t n_rows, const uint32_t* row_i,
                    const uint32_t* row_j, const double* row_mi, uint64_t n_sites,
                    double* mean_out, uint32_t* n_out);

/* empirical CDF of a reference sample evaluated at query values — stat.ecdf
 * (src/giremi/stat.py:7-29; the `mip` column, script/giremi.py:415-429):
 * out[q] = #{ref < query[q]} / n_ref  (strict, searchsorted side='left'); NaN queries give
 * NaN.  ref must not contain NaN; n_ref == 0 is LGMI_E_ARG (the reference divides by 0). */
int  lgmi_ecdf(lgmi_ctx* ctx, uint64_t n_ref, const double* ref, uint64_t n_query,
               const double* query, double* out);

/* ---- sharding plan, host only (no GPU touched): what lgmi_run_device() would do for (shard_rank, shard_world).
 * Lets a launcher balance / inspect shards and lets CPU-only tests check that every row of a shard is fed by
 * one of the shard's count tiles.  A work item is (site i, segment g): the partners number
 * [g * LGMI_EMIT_SEG, (g+1) * LGMI_EMIT_SEG) of site i's row, rows being in reference order.  Rows of x sites (every
 * site, or the het_snp sites when het_only) are cut into such segments; the row of any other site (its partners are
 * the later x sites) into segments of LGMI_EMIT_SEG_Q partners. */
#define LGMI_EMIT_SEG 8192u
/* ... and the row of any other site (its partners: the later x sites, in x-rank order) into segments of
 * LGMI_EMIT_SEG_Q partners (round 5; one item per such site until then): the kernels reach those slots by a walk
 * down a column of the slot matrix, a chain of dependent look-ups per 16 partners — a walk of 10,000 partners was the
 * longest thing a shard of a multi-GPU run waited for */
#define LGMI_EMIT_SEG_Q 1024u
typedef struct lgmi_shard_plan {
    uint64_t n_items_total;      /* work items of the whole batch                                  */
    uint64_t item_begin;         /* this shard's items are [item_begin, item_end)                  */
    uint64_t item_end;
    uint64_t n_examined_total;
    uint64_t n_examined;         /* examined pairs inside this shard's items                       */
    uint64_t n_tiles_total;      /* count tiles of the unsharded plan                              */
    uint64_t n_tiles;            /* count tiles this shard computes                                */
    const uint32_t* item_site;   /* [n_items_total]                                                */
    const uint32_t* item_seg;    /* [n_items_total]                                                */
    const uint32_t* tile_block;  /* [n_tiles] block of the tile                                    */
    const uint32_t* tile_x0;     /* [n_tiles] first slot-matrix row                                */
    const uint32_t* tile_y0;     /* [n_tiles] first slot-matrix column                             */
    const uint32_t* tile_edge;   /* [n_tiles] 64 (VALU popcount kernel) or 128 (matrix cores)      */
    const uint32_t* site_xrow;   /* [n_sites] slot-matrix row of the site in its block, or 0xFFFFFFFF */
    const uint32_t* site_ycol;   /* [n_sites] slot-matrix column                                   */
    const uint32_t* site_prow;   /* [n_sites] pseudo row (third-class plane) or 0xFFFFFFFF         */
    const uint32_t* site_pcol;   /* [n_sites] pseudo column or 0xFFFFFFFF                          */
    const uint32_t* site_xnext;  /* [n_sites] number of x sites of the block at or before the site */
    void* owner_;
} lgmi_shard_plan;
/* n_shuffles is the lgmi_params value the run will use: it only prices the work items (a pair with a tri-allelic
 * site costs n_shuffles table draws), i.e. it moves the shard boundaries, never the rows */
int  lgmi_plan_shard(const lgmi_batch* batch, int het_only, uint32_t n_shuffles, uint32_t shard_rank,
                     uint32_t shard_world, lgmi_shard_plan* out);
void lgmi_shard_plan_free(lgmi_shard_plan* plan);

/* device self-test of the one place where the permutation kernels use the hardware's f32 exp (perm.hip: le_exp):
 * for every k: fast[k] = le_exp(x2[k], t[k]), det[k] = (x2[k] <= det_exp(t[k])), e_hw[k] = (double)__expf((float)t[k]),
 * e_det[k] = det_exp(t[k]).  Host arrays of length n. */
int  lgmi_selftest_le_exp(lgmi_ctx* ctx, uint64_t n, const double* x2, const double* t, uint8_t* fast, uint8_t* det,
                          double* e_hw, double* e_det);

/* device self-test of the logarithm the MI of a row is made of (emit.hip: mi_log — table-driven, for doubles that hold
 * an integer >= 1): out[k] = mi_log(x[k]).  Host arrays of length n.  The reference's counterpart is numpy's log inside
 * scikit-learn's mutual_info_score (sklearn/metrics/cluster/_supervised.py:911-921). */
int  lgmi_selftest_log(lgmi_ctx* ctx, uint64_t n, const double* x, double* out);

/* wait for everything queued on the context's stream (the bench's device sync) */
int  lgmi_ctx_synchronize(lgmi_ctx* ctx);

/* ---- multi-GPU: one process per GPU, RCCL used only for the final gather ---- */
#define LGMI_UNIQUE_ID_BYTES 128
int  lgmi_comm_unique_id(void* out128);                 /* rank 0 creates, host broadcasts */
int  lgmi_comm_init(lgmi_ctx* ctx, const void* id128, int rank, int world);
/* What was bound and what the communicator itself says (ncclGetVersion, ncclCommCount, ncclCommUserRank, dladdr of the
 * library): a multi-GPU bench line carries this so that "RCCL saw N ranks" is a recorded fact, not an assumption.
 * stand_in = 1: LGMI_RCCL_LIB pointed the library at something else than librccl — the tests' file-based stand-in,
 * which is refused unless LGMI_ALLOW_RCCL_STANDIN=1.  -1 = the library has no such entry point.  (No reference
 * counterpart: the reference's parallelism is multiprocessing.Pool, src/giremi/script/giremi.py:375-380.) */
typedef struct lgmi_comm_info_t {
    int32_t rccl_version;     /* ncclGetVersion: major * 10000 + minor * 100 + patch */
    int32_t nranks, rank;     /* ncclCommCount / ncclCommUserRank of the context's communicator */
    int32_t world_given, rank_given;   /* what lgmi_comm_init() was called with */
    int32_t initialised;      /* the context has a communicator */
    int32_t stand_in;
    char    lib_path[512];
} lgmi_comm_info_t;
int  lgmi_comm_info(lgmi_ctx* ctx, lgmi_comm_info_t* out);
/* all-gather of one u64 per rank (row counts) over RCCL */
int  lgmi_comm_allgather_u64(lgmi_ctx* ctx, uint64_t mine, uint64_t* out_world);
/* all-gather of n u64 per rank: out_world[r * n + k] = rank r's mine[k] */
int  lgmi_comm_allgather_u64v(lgmi_ctx* ctx, const uint64_t* mine, uint32_t n, uint64_t* out_world);
/*
 * The final gather, HBM to HBM over xGMI: every rank's rows (row_i, row_j, row_mi [, row_p, row_exceed]
 * [, row_counts]) land on rank `root` in rank order as a new resident result (*out; NULL elsewhere) that
 * lgmi_dresult_fetch() brings to the host.  Rows stay attributable:
 *   - site_base is added to the rank's row_i / row_j, so that ranks that ran DIFFERENT batches (blocks dealt to
 *     ranks) report indices into one global site numbering; the gathered per-site arrays have
 *     max(site_base + n_sites) entries, each rank's per-site means stored at its base;
 *   - rank_row_begin (NULL or [world + 1], filled on every rank) tells which rows came from which rank;
 *   - same_batch = 1 (every rank ran a shard of the SAME batch, lgmi_params.shard_*): site_base must be 0
 *     everywhere, and the shards' per-site integer sums are added (ncclReduce) before the mean is taken —
 *     integer sums, so the result equals the unsharded run's bit for bit.
 * Every rank must call it; an error on one rank (allocation, mismatched flags) is agreed on before any
 * send / receive is posted, so no rank is left blocked.
 */
typedef struct lgmi_gather_opts {
    uint32_t site_base;
    uint8_t  same_batch;
    uint8_t  reserved[3];     /* must be 0 */
} lgmi_gather_opts;
int  lgmi_comm_gather(lgmi_ctx* ctx, const lgmi_dresult* mine, int root, const lgmi_gather_opts* opts_or_null,
                      lgmi_dresult** out, uint64_t* rank_row_begin_or_null);
/* The same gather in two halves, on a communication stream of its own: _begin exchanges the sizes, allocates on the
 * root and posts the transfer of (row_i, row_j, row_mi [, row_counts]) — it returns while that transfer is in flight,
 * so a result made by lgmi_run_device_rows() can run lgmi_dresult_permute() meanwhile; _finish posts what the
 * permutation stage produced (row_exceed [, row_p]), the per-site reduction, waits for everything and builds the
 * gathered result.  lgmi_comm_gather() is _begin followed by _finish.  `mine` must stay alive until _finish returns;
 * _finish always releases the handle. */
typedef struct lgmi_gather lgmi_gather;
int  lgmi_comm_gather_begin(lgmi_ctx* ctx, const lgmi_dresult* mine, int root, const lgmi_gather_opts* opts_or_null,
                            lgmi_gather** handle);
int  lgmi_comm_gather_finish(lgmi_gather* handle, lgmi_dresult** out, uint64_t* rank_row_begin_or_null);
/* lgmi_comm_gather() with default options followed by a fetch on the root: `out` is a host-resident
 * concatenation there, n_rows = 0 elsewhere */
int  lgmi_comm_gather_rows(lgmi_ctx* ctx, const lgmi_dresult* mine, int root,
                           lgmi_result* out);
void lgmi_comm_destroy(lgmi_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* LGMI_H */
