/*
 * lgmi_io.h — C ABI of liblgmi_io.so: streaming, indexed BGZF/BAM access for the l-giremi-compatible CLI
 * (SURVEY §8f N2).  Host only (zlib), no GPU.
 *
 * What it replaces in the reference (gxiaolab/L-GIREMI v0.2.4): the pysam.AlignmentFile the CLI opens at
 * src/giremi/script/giremi.py:21-24 and the three things the MI path does with it —
 *     sam.fetch(chromosome)                          src/giremi/footprint.py:6-28 (read intervals -> footprints)
 *     sam.fetch(chromosome, start, end)              src/giremi/mismatch.py:69-149 (cs-tag walk per read)
 *     sam.pileup(chromosome, start, end)             src/giremi/mismatch.py:160-190 (reference-allele read names)
 * pysam/htslib cannot be installed here, so the format is read from its specification (SAM/BAM v1, sections 4.1
 * BGZF, 4.2 BAM, 5.2 BAI).  A query reads only the BGZF blocks the index points at: memory is proportional to the
 * region, not to the file.  Without a .bai next to the BAM the index is built in memory by one streaming pass
 * (lgio_bam_build_index writes a standard .bai that samtools / pysam accept as well).
 *
 * Conventions: 0 on success, negative LGIO_E_* otherwise, message from lgio_last_error() (thread-local); results
 * are library-owned and released by the matching *_free(); coordinates are 0-based half-open like pysam's.
 */
#ifndef LGMI_IO_H
#define LGMI_IO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LGIO_ABI_VERSION 4
#define LGIO_OK        0
#define LGIO_E_ARG    -1
#define LGIO_E_IO     -2   /* open / read / seek failed                     */
#define LGIO_E_FORMAT -3   /* not BGZF / BAM / BAI, or a truncated record   */
#define LGIO_E_OOM    -4

/* what lgio_bam_fetch() materialises per read, beyond the fixed columns */
#define LGIO_NAMES  1u
#define LGIO_CIGAR  2u
#define LGIO_SEQ    4u   /* sequence (ASCII) and qualities */
#define LGIO_CS     8u   /* text of the cs:Z tag (minimap2 --cs) */
#define LGIO_AUX   16u   /* the raw auxiliary bytes (every tag)  */
#define LGIO_ALL   31u

typedef struct lgio_bam lgio_bam;

/* reads overlapping a region, in file (coordinate) order; variable-length fields are CSR-packed:
 * field k of read r is bytes [off[r], off[r+1]) of the pool */
typedef struct lgio_reads {
    uint64_t n;
    const int32_t*  tid;          /* [n] reference id                          */
    const int64_t*  start;        /* [n] 0-based leftmost position             */
    const int64_t*  end;          /* [n] one past the last aligned position    */
    const uint16_t* flag;         /* [n]                                       */
    const uint8_t*  mapq;         /* [n]                                       */
    const uint64_t* name_off;     /* [n+1] into names (no terminator)          */
    const char*     names;
    const uint64_t* cigar_off;    /* [n+1] into cigar, in operations           */
    const uint32_t* cigar;        /* BAM encoding: length << 4 | op (MIDNSHP=X)*/
    const uint64_t* seq_off;      /* [n+1] into seq and qual, in bases         */
    const char*     seq;          /* ASCII, =ACMGRSVTWYHKDBN                   */
    const uint8_t*  qual;         /* phred, 0xFF when absent                   */
    const uint64_t* cs_off;       /* [n+1] into cs                             */
    const char*     cs;
    const uint8_t*  has_cs;       /* [n]                                       */
    const uint64_t* aux_off;      /* [n+1] into aux                            */
    const uint8_t*  aux;
    void* owner_;
} lgio_reads;

/* pile-up columns with pysam's defaults (stepper 'samtools': unmapped, secondary, QC-failed and duplicate reads
 * and orphans of paired reads are skipped; bases below min_base_quality are dropped; at most max_depth reads per
 * column; columns are NOT truncated to the requested interval; a read that has a deletion or a reference skip at
 * the column contributes base 0 — pysam's empty string).  read[k] indexes `reads`. */
typedef struct lgio_pileup {
    uint64_t n_cols;
    const int64_t*  pos;          /* [n_cols] increasing                       */
    const uint64_t* col_off;      /* [n_cols+1] into read / base               */
    const uint32_t* read;
    const char*     base;
    lgio_reads reads;             /* names only                                */
    void* owner_;
} lgio_pileup;

int         lgio_abi_version(void);
const char* lgio_last_error(void);

int  lgio_bam_open(const char* path, lgio_bam** out);    /* loads <path>.bai (or <stem>.bai) when present */
void lgio_bam_close(lgio_bam* bam);
int  lgio_bam_n_refs(const lgio_bam* bam);
const char* lgio_bam_ref_name(const lgio_bam* bam, int tid);
int64_t     lgio_bam_ref_length(const lgio_bam* bam, int tid);
const char* lgio_bam_header_text(const lgio_bam* bam);
int  lgio_bam_has_index_file(const lgio_bam* bam);       /* 1: a .bai was loaded, 0: the index was built in memory */

/* one streaming pass over bam_path -> standard BAI at bai_path (NULL: <bam_path>.bai) */
int  lgio_bam_build_index(const char* bam_path, const char* bai_path_or_null);

/* reads overlapping [start, end) of reference tid (start < 0: the whole reference), unmapped reads skipped;
 * `what` = LGIO_* bits */
int  lgio_bam_fetch(lgio_bam* bam, int tid, int64_t start, int64_t end, uint32_t what, lgio_reads* out);
void lgio_reads_free(lgio_reads* reads);

/* Where this pile-up differs from pysam's AlignmentFile.pileup(contig, start, stop) with its defaults, which is what
 * the reference calls (src/giremi/mismatch.py:161) — none of it can show on the single-end long reads L-GIREMI is made
 * for, all of it would on paired short reads:
 *   - overlapping mates: htslib (ignore_overlaps=True) zeroes one mate's base quality where the two overlap, so that
 *     base is dropped; here both mates are emitted;
 *   - deletion / reference-skip entries are not subject to min_base_quality (htslib tests the quality at the query
 *     position next to them);
 *   - max_depth caps each column after filtering; htslib caps the reads entering the pile-up.
 * No entry point lets an exception out: allocation failures come back as LGIO_E_OOM. */
int  lgio_bam_pileup(lgio_bam* bam, int tid, int64_t start, int64_t end, int min_base_quality, int max_depth,
                     lgio_pileup* out);
void lgio_pileup_free(lgio_pileup* pile);


/* ---- round 3: the site extraction of one footprint, natively -------------------------------------------------
 * What get_region_mismatches_with_filters does with the alignment file (src/giremi/mismatch.py:29-290): the cs-tag
 * walk of every read (:69-149), the pile-up's reference-allele reads (:160-190), allele depths (:203-208), the
 * window filter (:211-240), the allele depth / ratio filters (:243-266) and the site depth / allele-count filters
 * (:268-290) — everything that needs only the BAM.  The homopolymer, simple-repeat and SNP steps (:292-340) need the
 * genome and the caller's lists and stay with the caller; they see only the few surviving sites.  Same quirks as the
 * Python path in lgmi/region.py, which remains the specification (tests/test_region_fast.py compares the two):
 * every covered position becomes an (empty, later removed) site, a site removed by the window filter comes back
 * empty when a later window looks at it, allele and site order are first-seen order.
 * Inputs this routine does not cover come back with fallback = 1 and nothing else filled (the caller then runs the
 * Python path, which treats them — or raises — the way the reference does): a read without a cs tag, a cs string
 * that is not minimap2's short form, a substitution that involves a base other than a/c/g/t. */
typedef struct lgio_site_params {
    int32_t keep_non_spliced_read;      /* mismatch.py:11 keep_non_spliced_read                                   */
    int32_t min_base_quality;           /* pile-up: pysam's default 13                                            */
    int32_t max_depth;                  /* pile-up: pysam's default 8000                                          */
    int32_t reserved;
    int64_t min_dist_from_splice;
    int64_t half_window;                /* round(mismatch_window_size / 2), rounded by the caller                 */
    double  min_allele_depth, min_allele_ratio, min_total_depth;
    double  max_window_mismatch, max_window_mismatch_type;
} lgio_site_params;

#define LGIO_REMOVED_WINDOW   0   /* 'too many window mismatches'           */
#define LGIO_REMOVED_DEPTH    1   /* 'too few usable reads after filters'   */
#define LGIO_REMOVED_ALLELES  2   /* 'not enough allele after filters'      */

typedef struct lgio_sites {
    int32_t fallback;                   /* 1: not covered here (see above)                                        */
    int32_t reserved;
    uint64_t n_sites;                   /* surviving sites: '+' strand first, each strand in first-seen order     */
    const uint8_t*  strand;             /* [n_sites] 0 '+', 1 '-'                                                 */
    const int64_t*  pos;                /* [n_sites]                                                              */
    const char*     ref;                /* [n_sites] upper case                                                   */
    const uint32_t* neighbor;           /* [n_sites * 16] window counts by change, index 4*ref + alt over "ACGT"
                                           (the bases as aligned: the caller complements them for '-')           */
    const uint64_t* allele_off;         /* [n_sites + 1] into allele_nt / reads_off                               */
    const char*     allele_nt;          /* surviving alleles, in first-seen order (reference allele last)         */
    const uint64_t* reads_off;          /* [n_alleles + 1] into reads                                             */
    const uint32_t* reads;              /* read ids: indexes of name_off                                          */
    uint64_t n_removed[2];              /* removed sites per strand, in the order the reference's dict holds them */
    const int64_t*  removed_pos[2];
    const uint8_t*  removed_code[2];    /* LGIO_REMOVED_*                                                         */
    uint64_t n_reads;                   /* every read fetch(start, end) returns, in file order                    */
    const uint64_t* name_off;           /* [n_reads + 1] into names                                               */
    const char*     names;
    void* owner_;
    /* ABI 3: read_uid[r] = the first read of the footprint with r's NAME (r itself for a name seen once).  The reference
     * keys its read -> allele maps by name (mutual_information.py:15-16), so two records of one name are ONE read to it;
     * a caller that packs the footprint from read ids (no name strings made) uses these instead of the names. */
    const uint32_t* read_uid;           /* [n_reads]                                                              */
} lgio_sites;

int  lgio_bam_region_sites(lgio_bam* bam, int tid, int64_t start, int64_t end, const lgio_site_params* params,
                           lgio_sites* out);
void lgio_sites_free(lgio_sites* sites);

/* ---- round 3: (start, end) of every mapped read of a reference, the footprint scan of src/giremi/footprint.py:6-28,
 * with the BGZF blocks inflated by `threads` threads (<= 1: the calling thread).  Same reads, same order, same values as
 * lgio_bam_fetch(bam, tid, -1, 0, 0, ...)'s start / end columns; needs a coordinate-sorted file (which the index
 * already presumes).  The scan is the one stage of a run whose cost is the size of the BAM. */
typedef struct lgio_intervals {
    uint64_t n;
    const int64_t* start;
    const int64_t* end;
    void* owner_;
} lgio_intervals;
int  lgio_bam_ref_intervals(lgio_bam* bam, int tid, int threads, lgio_intervals* out);
void lgio_intervals_free(lgio_intervals* iv);

/* ---- round 5 (ABI 3): the removed-site table, written natively.  One row per covered position of every footprint — 15
 * million rows for 8,000 genes — as pandas' to_csv(sep='\t', index=False) writes them (src/giremi/script/giremi.py:403-409:
 * chromosome, strand, pos, removed), from dictionary codes: row k = chrom_names[chrom_code[k]] TAB '+' or '-' (strand[k] 0
 * / 1) TAB pos[k] TAB reason_names[reason_code[k]] NEWLINE.  header != 0 writes the column line first; append != 0 appends
 * to the file.  The rows are formatted by `threads` threads into buffers written in order.  (Names must not hold a tab,
 * a quote or a newline — pandas would quote them: such a table is refused, LGIO_E_ARG, and the caller takes pandas.) */
int  lgio_write_removed_table(const char* path, int append, int header, uint64_t n, const int32_t* chrom_code,
                              const int8_t* strand, const int64_t* pos, const int8_t* reason_code,
                              const char* const* chrom_names, uint32_t n_chrom, const char* const* reason_names,
                              uint32_t n_reasons, int threads);

/* ---- round 5 (ABI 4): any table of integers, floats and dictionary strings, written the way
 * pandas.DataFrame.to_csv(sep='\t', index=False) writes it — PREFIX.mi.txt (src/giremi/script/giremi.py:396-401:
 * chromosome, strand, site1_pos, site1_type, site2_pos, site2_type, mi [, p_perm]).  A float64 is written as numpy's
 * astype(str) writes it (what to_csv does): the shortest digits that read back as the same double, positional for
 * 1e-4 <= |x| < 1e16 with at least one digit behind the point, d[.ddd]e+XX otherwise, NaN as an empty field.
 * lgio_format_doubles writes the same text into out + k * stride (NUL-terminated, stride >= 33): the tests compare it with
 * numpy on millions of values.  Names (columns and dictionary entries) that pandas would quote are refused (LGIO_E_ARG). */
#define LGIO_COL_I64  0u       /* data: const int64_t[n_rows] */
#define LGIO_COL_F64  1u       /* data: const double[n_rows] */
#define LGIO_COL_DICT 2u       /* data: const int32_t[n_rows], codes into names[n_names] */
typedef struct lgio_table_col {
    const char* name;
    uint32_t kind, n_names;
    const void* data;
    const char* const* names;
} lgio_table_col;
int  lgio_write_table(const char* path, int append, int header, uint64_t n_rows, uint32_t n_cols, const lgio_table_col* cols,
                      int threads);
int  lgio_format_doubles(uint64_t n, const double* x, char* out, uint32_t stride);

/* bytes of compressed file read so far through this handle (tests use it to show that a region query does not
 * read the whole file) */
uint64_t lgio_bam_bytes_read(const lgio_bam* bam);

#ifdef __cplusplus
}
#endif
#endif /* LGMI_IO_H */
