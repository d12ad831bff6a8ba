/*
** gpumap.h -- C-ABI of the MI355X-native query hot path for ezorita/mapper.
**
** The library (mapper_amd/lib/libgpumap.so) replaces, for whole batches of reads at once,
** the per-read composition the reference runs on the CPU:
**
**     sym_str_index                (src/index_sym.c:293-319)
**     blocksearch_trail_rec        (src/blocksearch.c:106-177)  + seqsearch_fw/bw (:184-311)
**                                                               + seqdash_fw/bw   (:381-496)
**     bwt_query_all / get_occ_all  (src/index_bwt.c:198-304, :840-905)
**     sar_get / sar_get_range      (src/index_sar.c:92-161)
**     strand + coordinate rule     (src/index_txt.c:492-499)
**
** Everything here is plain C: pointers and sizes, no C++/torch types.  Conventions follow the
** reference's errhandler (src/errhandler.h:11-70, src/errhandler.c:3-16): functions return
** 0 / -1 (pointers: NULL), print `func\t(file:line) error: msg` on stderr and set errno = 1.
** There is no CPU fallback: every compute entry point needs a HIP device and fails without one.
**
** INTEGRATION.md shows the few lines a maintainer adds to src/mapper.c to call this library.
*/
#ifndef GPUMAP_H_
#define GPUMAP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GPUMAP_QUERY_PREFIX 0      /* BWT_QUERY_PREFIX, src/index_bwt.h:22 */
#define GPUMAP_QUERY_SUFFIX 1      /* BWT_QUERY_SUFFIX, src/index_bwt.h:23 */
#define GPUMAP_ALIGN_WORDS  4      /* ALIGN_WORDS,      src/blocksearch.h:12 */
#define GPUMAP_MAX_READ_LEN 256    /* ALIGN_WORDS * ALIGN_WORD_SIZE, src/blocksearch.h:12-13 */
#define GPUMAP_MAX_TAU      7      /* mismatch positions are kept in 7 bytes per search node */

typedef struct gpumap_index gpumap_index;   /* host-side index handle (mmapped files)  */
typedef struct gpumap_ctx   gpumap_ctx;     /* one GPU: device-resident index + scratch */
typedef struct gpumap_batch gpumap_batch;   /* one batch of reads resident in HBM       */

/*
** Raw view of an index, i.e. the fields of the reference's private structs bwt_t
** (src/index_bwt.c:4-14), sar_t (src/index_sar.c:4-11), txt_t (src/index_txt.c:4-18) and sym_t
** (src/index_sym.c:4-9), as they lie in the mmapped files.  The struct
** bodies are private to the reference's .c files, so a caller fills this either from
** gpumap_index_read() or with pointers it already owns.  Host memory only has to stay valid
** until gpumap_create() returns.
*/
typedef struct {
   /* .bwt */
   const uint64_t *occ;            /* interleaved marks / bit planes, src/index_bwt.c:525-564 */
   int64_t         occ_length;     /* in 64-bit words */
   int64_t         occ_mark_intv;  /* word groups between marks (14) */
   int64_t         occ_word_size;  /* must be 64 */
   const uint64_t *c;              /* c[sym_count+1], src/index_bwt.c:575-582 */
   /* .sar */
   const int64_t  *sa;             /* bit-packed, LSB first, src/index_sar.c:283-328 */
   int64_t         sar_bits;
   int64_t         sar_len;        /* in 64-bit words */
   /* .txt */
   int64_t         txt_len;        /* n = 2 * (sum of sequence lengths incl. separators) */
   int64_t         seq_cnt;
   const int64_t  *seq_len;        /* [seq_cnt] */
   const int64_t  *seq_beg;        /* [seq_cnt] */
   const char *const *seq_name;    /* [seq_cnt] */
   int64_t         wil_cnt;
   int64_t         rc_flag;
   const uint8_t  *text;           /* [txt_len] codes 0..5; not needed on the GPU, may be NULL */
   /* .sym */
   int32_t         sym_count;      /* must be 5 (A C G T N) */
   const uint8_t  *sym_table;      /* [256] char -> code, src/index_sym.c:121-132 */
   const uint8_t  *com_table;      /* [sym_count+1] complement, src/index_sym.c:204-236 */
} gpumap_index_view;

/* ------------------------------------------------------------------------------------------
** Host-side index handle.  Mirrors index_t and its functions (src/index.h:19-33).
** ---------------------------------------------------------------------------------------- */

/* index_read (src/index.c:7-96): mmaps base.sym/.txt/.sar/.bwt (annotations are not read). */
gpumap_index *gpumap_index_read(const char *filename_base);
/* index_build (src/index.c:99-162): FASTA -> base.sym/.txt/.sar/.bwt, byte-identical to the files
** the reference writes (own suffix sorter instead of divsufsort; same order, src/index_sar.c:46-54). */
gpumap_index *gpumap_index_build(const char *ref_fasta, const char *filename_base);
/* index_free (src/index.c:216-238). */
void          gpumap_index_free(gpumap_index *index);
int           gpumap_index_view_of(const gpumap_index *index, gpumap_index_view *view);
/* txt_pos_to_str (src/index_txt.c:480-511): "name:pos:strand" into buf; returns its length. */
int           gpumap_pos_to_str(const gpumap_index *index, int64_t pos, char *buf, size_t bufsz);

/* ------------------------------------------------------------------------------------------
** Device context: uploads the index to the current process's GPU `device` and repacks the Occ
** table into cache-line blocks (one 64-byte block = 128 BWT positions + running counts).
** ---------------------------------------------------------------------------------------- */
gpumap_ctx *gpumap_create(const gpumap_index_view *view, int device);
/* Same, reading the view from base.* files (what src/mapper.c would pass: index->fname_base). */
gpumap_ctx *gpumap_create_from_files(const char *filename_base, int device);
void        gpumap_destroy(gpumap_ctx *ctx);

/* index_build on the device: the reference builds on the host (src/index.c:99-162: txt_commit_rc,
** sar_build = divsufsort + compact_array, bwt_build_opt), which takes hours at GRCh38 size; this
** produces the same arrays in HBM (text F$rc(F)$, packed suffix array, Occ array in the reference's
** layout; bit-identical to the host builder's) and the rank blocks from them.  `codes`: the
** sequences' symbol codes (0..4) one after another, in host memory or (codes_on_device != 0) in
** device memory; seq_lens[seq_cnt].  keep_reference != 0 keeps the text and the reference-layout Occ
** array in HBM so that gpumap_reference_fetch can hand them out (e.g. to write base.* files). */
gpumap_ctx *gpumap_build_device(const uint8_t *codes, int codes_on_device, const int64_t *seq_lens,
                                int64_t seq_cnt, int device, int keep_reference);
/* sizes[0] = txt_len, [1] = occ_length (64-bit words), [2] = sar_len (words), [3] = sar_bits,
** [4] = wil_cnt, [5] = bytes of text kept (0 if none). */
int         gpumap_reference_sizes(const gpumap_ctx *ctx, int64_t sizes[6]);
/* Copies to host memory (any pointer may be NULL): the .bwt Occ array and c[6], the .sar array,
** the .txt text. */
int         gpumap_reference_fetch(gpumap_ctx *ctx, uint64_t *occ, uint64_t c[6], int64_t *sa, uint8_t *text);

/* Device-resident index arrays, for replication across GPUs (rank 0 exports, RCCL broadcasts the
** three buffers, the other ranks import).  Pointers are device pointers; sizes in bytes.
** `header` is a small host blob (gpumap_devindex_header_size() bytes) describing the layout. */
typedef struct {
   void   *occ_blocks;  int64_t occ_bytes;   /* repacked Occ blocks               */
   void   *sa_words;    int64_t sa_bytes;    /* packed suffix array (verbatim)    */
   void   *tables;      int64_t tables_bytes;/* superblock counts, C, seq_beg ... */
   void   *text;        int64_t text_bytes;  /* text codes + 16 bytes of padding; NULL / 0: none (the
                                             ** search then stays on the Occ table, see k_verify)  */
} gpumap_devindex;
size_t      gpumap_devindex_header_size(void);
int         gpumap_devindex_export(const gpumap_ctx *ctx, gpumap_devindex *out, void *header);
/* Allocation-free import: the caller owns the three device buffers (e.g. torch tensors filled by
** a broadcast) and keeps them alive as long as the ctx. */
gpumap_ctx *gpumap_create_from_device(const gpumap_devindex *dev, const void *header, int device);

/* ------------------------------------------------------------------------------------------
** Batched operators (what the reference exposes one query at a time).  Host pointers.
** ---------------------------------------------------------------------------------------- */

/* get_occ_all (src/index_bwt.c:840-905) for n positions: occ[i][0..4] = symbol counts in
** BWT[0..ptr[i]]; ptr[i] == -1 gives zeros (:858-861). */
int gpumap_occ_all(gpumap_ctx *ctx, int64_t n, const int64_t *ptr, int64_t *occ);
/* bwt_query_all (src/index_bwt.c:198-304) for n intervals: out[i][j] = {fp, rp, sz} of the
** extension by symbol j at `end` (GPUMAP_QUERY_PREFIX / _SUFFIX); depth is the caller's dp + 1. */
int gpumap_query_all(gpumap_ctx *ctx, int end, int64_t n, const int64_t *fp, const int64_t *rp,
                     const int64_t *sz, int64_t *out);
/* gpumap_query_all without any allocation.  No counterpart in the reference: it is for callers that cannot
** call hipMalloc (code under HIP graph capture, tight serving loops, harnesses that forbid allocation).
** Same arguments and results as gpumap_query_all, at most GPUMAP_NOALLOC_MAX_QUERIES intervals per call
** (-1 with a message beyond that).  The staging buffers are a static device array of 2.25 MiB that every GPU the
** library runs on holds until the process ends; every call on a GPU shares it, so calls that overlap
** on one GPU (also through different contexts) race: serialise them. */
#define GPUMAP_NOALLOC_MAX_QUERIES 16384
int gpumap_query_all_noalloc(gpumap_ctx *ctx, int end, int64_t n, const int64_t *fp, const int64_t *rp,
                             const int64_t *sz, int64_t *out);
/* sar_get (src/index_sar.c:92-118) for n positions; out-of-range positions give -1. */
int gpumap_sar_get(gpumap_ctx *ctx, int64_t n, const int64_t *pos, int64_t *out);
/* sar_get_range (src/index_sar.c:121-161). */
int gpumap_sar_get_range(gpumap_ctx *ctx, int64_t beg, int64_t size, int64_t *vec);
/* sym_str_index (src/index_sym.c:293-319) for a batch: text = concatenated read letters,
** offsets[n_reads+1]; codes gets the same layout. */
int gpumap_encode(gpumap_ctx *ctx, const char *text, const int64_t *offsets, int64_t n_reads,
                  uint8_t *codes);

/* Page-locked host memory for read buffers handed to gpumap_batch_upload* (the copy to the GPU then
** runs at the full rate of the link; pageable memory works too, slower).  No reference counterpart:
** the reference never leaves the host. */
void *gpumap_host_alloc(size_t bytes);
void  gpumap_host_free(void *p);

/* ------------------------------------------------------------------------------------------
** Read mapping: blocksearch_trail_rec(q, 0, m-1, tau+1, 0, bwt, tree) for every read, then
** sar_get_range for every hit, then the strand rule.
** ---------------------------------------------------------------------------------------- */

typedef struct {
   int64_t   n_reads;
   /* status[r]: 0 = searched; -1 = the reference returns -1 for this read (empty read, fewer
   ** letters than tau+1 blocks can share: src/blocksearch.c:126; or longer than 256). */
   int64_t  *status;          /* [n_reads] */

   /* Located hits, reads in input order, within a read by ascending text position. */
   int64_t   n_locs;
   int64_t  *read_first_loc;  /* [n_reads+1] CSR offsets into the loc_* arrays */
   int64_t  *loc_read;        /* [n_locs] */
   int64_t  *loc_pos;         /* [n_locs] SA value: position in text = F$rc(F)$ (sar_get)       */
   int64_t  *loc_fwd;         /* [n_locs] pos < n/2 ? pos : n-2-pos   (src/index_txt.c:495)      */
   int64_t  *loc_seq;         /* [n_locs] sequence id (seq_bisect, src/index_txt.c:499,770-787)  */
   int64_t  *loc_seqpos;      /* [n_locs] 1-based position inside the sequence (:505)            */
   uint8_t  *loc_strand;      /* [n_locs] 0 '+', 1 '-'                (src/index_txt.c:494)      */
   uint8_t  *loc_score;       /* [n_locs] mismatches (query N not counted, as in the reference)  */

   /* Interval hits = content of tree->stack after blocksearch_trail_rec (spath_t,
   ** src/blocksearch.h:16-20), sorted by (read, fp).  Filled when want_intervals != 0. */
   int64_t   n_hits;
   int64_t  *hit_read, *hit_fp, *hit_rp, *hit_sz, *hit_dp, *hit_score;   /* [n_hits] */
   uint64_t *hit_align;       /* [n_hits][GPUMAP_ALIGN_WORDS] mismatch + query-N bitmap */

   /* Work counters (for the roofline): all-symbol extension steps executed (= bwt_query_all
   ** calls of the reference schedule) and device milliseconds of the stages. */
   int64_t   n_ext;
   float     ms_search, ms_locate, ms_sort, ms_total;
   /* Not NULL: the arrays lie in the context's pinned result arena (option "pinned_results": two
   ** arenas used in turn) and stay valid until the second gpumap_batch_fetch / gpumap_map_batch after
   ** this one on that context, so that one result can be consumed while the next is fetched;
   ** gpumap_hits_free then only clears the struct. */
   void     *arena;
} gpumap_hits;

/* Host buffers in, host arrays out (malloc'ed by the library, released by gpumap_hits_free).
** codes: concatenated symbol codes 0..4 (sym_str_index), offsets[n_reads+1]. */
int  gpumap_map_batch(gpumap_ctx *ctx, const uint8_t *codes, const int64_t *offsets,
                      int64_t n_reads, int tau, int want_intervals, gpumap_hits *out);
void gpumap_hits_free(gpumap_hits *hits);

/* The same in three steps, so that a caller can keep batches resident in HBM and time the
** device work alone (bench.py): upload -> run (search + locate + sort, results stay on the
** device) -> fetch. */
gpumap_batch *gpumap_batch_upload(gpumap_ctx *ctx, const uint8_t *codes, const int64_t *offsets,
                                  int64_t n_reads);
/* Letters instead of codes: encoded on the device with the index's sym_table. */
gpumap_batch *gpumap_batch_upload_text(gpumap_ctx *ctx, const char *text, const int64_t *offsets,
                                       int64_t n_reads);
/* The next chunk into the same batch object (the iteration of mapper()'s loop, src/mapper.c:23-40):
** the device buffers the batch holds are kept and grow only when the new chunk needs more, so a
** caller that maps chunk after chunk pays for the allocations once.  Results of the previous chunk
** that were not fetched are gone. */
int           gpumap_batch_refill_text(gpumap_ctx *ctx, gpumap_batch *batch, const char *text,
                                       const int64_t *offsets, int64_t n_reads);
int           gpumap_batch_run(gpumap_ctx *ctx, gpumap_batch *batch, int tau, int want_intervals);
int           gpumap_batch_fetch(gpumap_ctx *ctx, gpumap_batch *batch, gpumap_hits *out);
/* Counters of the last run without copying hits: n_hits, n_locs, n_ext, and stage times (ms). */
int           gpumap_batch_counters(const gpumap_batch *batch, int64_t *n_hits, int64_t *n_locs,
                                    int64_t *n_ext, float ms[4]);
/* Work figures of the last search (tuning aid; no counterpart in the reference).  n_ext, the
** bwt_query_all calls of the reference schedule, is made of three parts: steps k_search executed on
** the Occ table, steps the jump table stood in for (stats[2]) and steps k_verify replayed on the text
** (stats[4]).  stats[0] = passes of the search loop summed over all wavefronts (64 nodes each when
** full), stats[1] = nodes pushed back as continuations, stats[3] = entries k_search wrote to its
** output list (hits and candidates for k_verify), stats[5] = length of that list with the unused ends
** of the wavefronts' reservations, stats[6], stats[7] = microseconds of k_search and of k_verify
** (HIP events on the library's stream; their sum is ms_search). */
int           gpumap_batch_stats(const gpumap_batch *batch, int64_t stats[8]);
void          gpumap_batch_free(gpumap_batch *batch);

/* ------------------------------------------------------------------------------------------
** mapper() (src/mapper.c:3-47, src/mapper.h:10): reads `qfile` (FASTQ / FASTA / raw, detected
** as io_stream_set_parser does, src/io.c:232-266) in chunks, maps each chunk on the GPU and
** writes one line per located hit to `out_path` (NULL or "-" = stdout):
**     tag \t name:pos:strand \t score
** ---------------------------------------------------------------------------------------- */
int gpumap_mapper(const char *qfile, const gpumap_index *index, gpumap_ctx *ctx, int tau,
                  const char *out_path);

/* Options.  None changes a result, only where and how the work is done:
**   "waves_per_cu"   persistent waves per CU of the search kernel (0 = as many as are resident)
**   "spill_nodes"    per-wave spill capacity of the search deque, in nodes
**   "pinned_results" 1 = gpumap_batch_fetch hands out slices of page-locked arenas of the context
**   "jump_k"         letters of the jump table (0 = none, -1 = by the size of the text and the free memory)
**   "verify"         0 = every step on the Occ table; 1 = nodes of one suffix are finished on the text;
**                    2 = also nodes of two suffixes; -1 = by the size of the text (the default) */
int gpumap_set_option(gpumap_ctx *ctx, const char *name, int64_t value);
const char *gpumap_version(void);

#ifdef __cplusplus
}
#endif
#endif
