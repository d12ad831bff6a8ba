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
   int64_t         txt_len;        /* n = 2 * (sum of sequence lengths incl. separators); 2 <= n < 2^33 */
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
/* bwt_prefix_all (src/index_bwt.c:390-446) for n intervals: out[i][j] = {fp, sz} of the one-letter prefix
** extension by symbol j, fp = c[j] + occ(fp_i - 1)[j], sz clamped at 0 (:432); the reference leaves rp = -1, so no
** reverse pointer here either.  An interval whose ranks get_occ_all refuses (fp < 0, fp + sz outside 0..n) gives
** ten words of -1. */
int gpumap_prefix_all(gpumap_ctx *ctx, int64_t n, const int64_t *fp, const int64_t *sz, int64_t *out);
/* gpumap_query_all without any allocation.  No counterpart in the reference: it is for callers that cannot
** call hipMalloc (tight serving loops, harnesses that forbid allocation, a phase of a program in which the device
** heap must not change).  The call copies from and to the caller's host arrays on the library's stream and waits for
** it, so it cannot itself be recorded into a HIP graph.  Same arguments and results as gpumap_query_all, at most GPUMAP_NOALLOC_MAX_QUERIES intervals per call
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
/* The library keeps the page-locked blocks of finished pipelines and mapper() calls for the next ones (allocating
** them costs about as much as mapping the chunk they hold; at most GPUMAP_PINNED_POOL_BYTES, default 8 GiB): this call
** frees the ones not in use. */
void  gpumap_host_trim(void);

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
   ** src/blocksearch.h:16-20), sorted by (read, fp) (the reference's order is the order of its depth-first
   ** search; the sets are equal).  Filled when want_intervals != 0; fewer than 2^24 - 1 reads per batch then. */
   int64_t   n_hits;
   int64_t  *hit_read, *hit_fp, *hit_rp, *hit_sz, *hit_dp, *hit_score;   /* [n_hits] */
   uint64_t *hit_align;       /* [n_hits][GPUMAP_ALIGN_WORDS] mismatch + query-N bitmap */

   /* Work counters (for the roofline): all-symbol extension steps executed (= bwt_query_all
   ** calls of the reference schedule) and device milliseconds of the stages. */
   int64_t   n_ext;
   float     ms_search, ms_locate, ms_sort, ms_total;
   /* Not NULL: the arrays lie in a page-locked result arena of the batch (option "pinned_results": two
   ** arenas used in turn) and stay valid until the second gpumap_batch_fetch after this one on that batch
   ** (gpumap_map_batch: on that context), so that one result can be consumed while the next is fetched, and
   ** no longer than the batch (the context); gpumap_hits_free then only clears the struct. */
   void     *arena;
   /* Compact results (option "compact_results", or GPUMAP_PIPE_COMPACT): the located hits as their sorted 64-bit
   ** keys, key = read << (key_pos_bits + 4) | text position << 4 | score; loc_read .. loc_score are then NULL
   ** (8 bytes cross the link per hit instead of 42; everything else follows from the key: GPUMAP_KEY_*,
   ** gpumap_hits_expand).  NULL in a full result. */
   uint64_t *loc_key;         /* [n_locs] */
   int32_t   key_pos_bits;    /* = sar_bits of the index */
   int32_t   expanded;        /* bit 0: gpumap_hits_expand has filled loc_read .. loc_score (malloc'ed); bit 1: status,
                              ** read_first_loc and loc_key of a packed result (malloc'ed) */
   /* Hit lines (GPUMAP_PIPE_LINES): the located hits as the text mapper() prints, one line per hit in the order of
   ** the loc_* arrays, "tag \t name:pos:strand \t score \n" (txt_pos_to_str, src/index_txt.c:480-511); every other
   ** array is NULL then (n_reads, n_locs, n_ext are set).  NULL otherwise. */
   char     *line_text;       /* [line_bytes], not NUL-terminated */
   int64_t   line_bytes;
   /* Packed results (GPUMAP_PIPE_PACKED): the compact result with nothing in it that follows from something else.  Per
   ** located hit 5 bytes, little-endian: text position << 4 | score (the read is the one whose offsets hold the hit);
   ** per read a 4-byte offset (there are fewer than 2^31 hits per batch) and one status byte (0 = searched, 1 = the
   ** reference returns -1).  At 3.3 hits per read that is 21.7 bytes per read on the link instead of 42.7.  status,
   ** read_first_loc and loc_key are NULL until gpumap_hits_expand has made them (malloc'ed) from these.  NULL otherwise. */
   uint8_t  *loc_packed;      /* [5 * n_locs] */
   uint32_t *read_first32;    /* [n_reads + 1] */
   uint8_t  *status8;         /* [n_reads] */
} gpumap_hits;
/* hit i of a packed result: text position << 4 | score */
#define GPUMAP_PACKED_AT(p, i)    ((uint64_t)(p)[5 * (i)] | ((uint64_t)(p)[5 * (i) + 1] << 8) | ((uint64_t)(p)[5 * (i) + 2] << 16) | \
                                   ((uint64_t)(p)[5 * (i) + 3] << 24) | ((uint64_t)(p)[5 * (i) + 4] << 32))
#define GPUMAP_KEY_SCORE(k)       ((int)((k) & 15))
#define GPUMAP_KEY_POS(k, bits)   ((int64_t)(((k) >> 4) & ((1ull << (bits)) - 1ull)))
#define GPUMAP_KEY_READ(k, bits)  ((int64_t)((k) >> ((bits) + 4)))

/* Host buffers in, host arrays out (malloc'ed by the library, released by gpumap_hits_free).
** codes: concatenated symbol codes 0..4 (sym_str_index), offsets[n_reads+1]. */
int  gpumap_map_batch(gpumap_ctx *ctx, const uint8_t *codes, const int64_t *offsets,
                      int64_t n_reads, int tau, int want_intervals, gpumap_hits *out);
void gpumap_hits_free(gpumap_hits *hits);
/* blocksc_trail(query, qarray, slen, tau, 0, tree) (src/blocksearch.c:7-103, with scsearch_fw :314-378) for a batch
** of queries: the search ann_build_mt runs for every k-mer of the text (src/index_ann.c:556-567): the neighbours of the
** query that are lexicographically at or below it, every query N lowering tau by one.  The path array the reference's
** callers compute (one bwt_query SUFFIX per letter, src/index_ann.c:791) is computed on the device.  `trail`, the
** caller's reuse of the last query's left block, is not needed: every query is searched afresh, which gives the same
** hits (tests/test_oracle_blocksc.py).  out: status[q] (0 / -1 as blocksc_trail returns; -1 also for tau = 0 on the
** last strand class, where the reference's one-node tree fails), hit_* = tree->stack (score with the N count added,
** :94-97) sorted by (query, fp), loc_* = their suffix-array entries.  At most 256 / (2 (tau + 1)) distinct query
** lengths per batch. */
int  gpumap_blocksc_batch(gpumap_ctx *ctx, const uint8_t *codes, const int64_t *offsets, int64_t n_queries,
                          int tau, gpumap_hits *out);

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
/* An empty batch object (its own stream, scratch and result arenas: batches of one context can be filled, run
** and fetched side by side from different threads), to be filled with gpumap_batch_stage / _refill_text. */
gpumap_batch *gpumap_batch_new(gpumap_ctx *ctx);
/* Stages the next chunk without encoding it: gpumap_batch_run then starts with the encoding (letters -> codes ->
** bit planes on the device), so that the run covers encode + search + locate + order.  flags: GPUMAP_SRC_TEXT =
** `src` holds letters (else codes 0..4); GPUMAP_SRC_DEVICE = `src` and `offsets` are device pointers that the
** batch uses in place (no copy; they must stay valid until the last run of the staged chunk is over: a run with
** interval output after a run without encodes the chunk again), else host pointers that are copied to the device
** by the time the batch has been run. */
#define GPUMAP_SRC_TEXT   1
#define GPUMAP_SRC_DEVICE 2
/* GPUMAP_SRC_FILE (gpumap_pipe only) = `src` holds a block of a FASTQ / FASTA / raw read file (host memory) that starts
** at the first line of a record, and `offsets` points to five int64 that describe it: [0] its bytes (< 2 GiB), [1] lines
** per record (4 FASTQ, 2 FASTA, 1 raw: io_stream_set_parser, src/io.c:232-266), [2] the lines it holds (a last line
** without newline counts), [3] unused, [4] the line number of its first line in the file (raw files: a read's tag is
** its line number, src/io.c:171-174); n_reads = the records to map (whole records only: [2] / [1] at most).  The block
** is parsed on the device by the rules of seqread_parse_fastq / _fasta (src/seqread.c:119-164): sequence = line 1 of a
** record (line 0 of a raw file), tag = line 0 without its first character, a line ends at its newline or at a NUL, a
** FASTQ quality line of another length than the sequence fails the chunk. */
#define GPUMAP_SRC_FILE   8
int           gpumap_batch_stage(gpumap_ctx *ctx, gpumap_batch *batch, const void *src, const int64_t *offsets,
                                 int64_t n_reads, int flags);
/* The arrays loc_read .. loc_score of a compact result, derived on the host from loc_key by the rule of
** txt_pos_to_str / seq_bisect (src/index_txt.c:492-505, :770-787); malloc'ed, released by gpumap_hits_free. */
int           gpumap_hits_expand(const gpumap_ctx *ctx, gpumap_hits *hits);
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
/* More of the same: stats[0] = searches the run took (2: the first guess of the hit list's length was too short and
** the search ran again with the exact length), stats[1] = microseconds of the encoding stage of a staged batch,
** stats[2] = 1 when the run left compact results, stats[3] = capacity of the hit list in entries, stats[4] = chunks
** of 64 nodes that the wavefronts' deques spilled to global memory, stats[5] = suffixes (windows of the text)
** that k_verify and k_verify_group replayed, stats[6] = candidates of three and four suffixes (k_verify_group);
** stats[7] is 0. */
int           gpumap_batch_stats_ext(const gpumap_batch *batch, int64_t stats[8]);
void          gpumap_batch_free(gpumap_batch *batch);

/* ------------------------------------------------------------------------------------------
** The loop of mapper() (src/mapper.c:23-40: read a chunk, work on it, print it, next chunk) as a pipeline:
** `depth` batch objects, each with a worker thread and a stream of its own, so that the copy of chunk i+1 to
** the device, the search of chunk i and the copy of the results of chunk i-1 to the host overlap.  Chunks
** come back in the order they went in.  With several contexts (replicas of the index on several GPUs,
** gpumap_multi) the chunks go round the GPUs: reads stay in input order.
**   flags: GPUMAP_SRC_TEXT, GPUMAP_SRC_DEVICE (as gpumap_batch_stage), GPUMAP_SRC_FILE (blocks of the read file, parsed
**   on the device), GPUMAP_PIPE_COMPACT (compact results), GPUMAP_PIPE_PACKED (packed results), GPUMAP_PIPE_LINES (hit lines made on the
**   device).
** Results lie in page-locked arenas of the pipe and stay valid until gpumap_pipe_release.
** ---------------------------------------------------------------------------------------- */
typedef struct gpumap_pipe gpumap_pipe;
#define GPUMAP_PIPE_COMPACT 4
/* Results as hit lines (gpumap_hits.line_text), formatted on the device; needs GPUMAP_SRC_FILE (the tags) and
** gpumap_set_seq_names on every context of the pipe. */
#define GPUMAP_PIPE_LINES   16
/* Packed results (gpumap_hits.loc_packed, read_first32, status8): what GPUMAP_PIPE_COMPACT returns in half the bytes. */
#define GPUMAP_PIPE_PACKED  32
gpumap_pipe *gpumap_pipe_create(gpumap_ctx *const *ctxs, int n_ctx, int depth, int tau, int flags);
/* Hands a chunk to the pipe.  A slot is taken from submit to release; while all `depth` slots are taken the call
** waits for the caller's other thread (the one that calls gpumap_pipe_next / gpumap_pipe_release) to free one.  If no
** other thread has ever taken a result (mapper() is single-threaded, src/mapper.c:23-40) nobody can free a slot: the
** call then fails with -1 and a message (after 2 s of grace for a consumer thread that is only just starting) instead
** of waiting for ever.  src / offsets must stay valid until the chunk's result has been returned.  `user` comes
** back with the result.  Several threads may submit; the order of their chunks is the order the calls get the lock. */
int          gpumap_pipe_submit(gpumap_pipe *pipe, const void *src, const int64_t *offsets, int64_t n_reads, void *user);
/* The same without waiting: 1 = submitted, 0 = every slot is taken (take a result first), -1 = error. */
int          gpumap_pipe_try_submit(gpumap_pipe *pipe, const void *src, const int64_t *offsets, int64_t n_reads, void *user);
/* No more chunks will be submitted: gpumap_pipe_next returns 0 once everything has been returned. */
void         gpumap_pipe_close(gpumap_pipe *pipe);
/* Chunks submitted whose results have not been returned yet. */
int64_t      gpumap_pipe_in_flight(gpumap_pipe *pipe);
/* The next result in submit order: 1 = *out and *user filled, -1 = that chunk failed (its slot is released),
** 0 = the pipe is closed and everything has been returned.  With nothing in flight and the pipe still open the call
** waits for a submit from another thread: a single-threaded caller asks gpumap_pipe_in_flight first. */
int          gpumap_pipe_next(gpumap_pipe *pipe, gpumap_hits *out, void **user);
/* The caller is done with a result of gpumap_pipe_next: its slot takes the next chunk. */
void         gpumap_pipe_release(gpumap_pipe *pipe, gpumap_hits *hits);
void         gpumap_pipe_destroy(gpumap_pipe *pipe);

/* ------------------------------------------------------------------------------------------
** Several GPUs (no counterpart in the reference, which never leaves one CPU).  n_members replicas of
** the index, member i on GPU devices[i] (devices == NULL: GPUs 0 .. n_members-1).  The index is uploaded and
** repacked once, on the first member's GPU; the other GPUs receive the device arrays by ncclBroadcast (RCCL over
** xGMI; librccl.so is loaded only then); members that share a GPU with an earlier member adopt its arrays (several
** members on one GPU: tests).  gpumap_multi_map_batch is gpumap_map_batch with the reads split into contiguous ranges
** in input order, one host thread per member, the hit lists concatenated in member order: the output of one GPU.
** gpumap_multi_ctxs gives the members to gpumap_pipe_create / gpumap_mapper_multi (chunks go round the GPUs).
** ---------------------------------------------------------------------------------------- */
typedef struct gpumap_multi gpumap_multi;
gpumap_multi *gpumap_create_multi(const gpumap_index_view *view, int n_members, const int *devices);
/* The same around an existing context (e.g. of gpumap_build_device), which becomes member 0 and stays the caller's. */
gpumap_multi *gpumap_multi_from_ctx(gpumap_ctx *root, int root_device, int n_members, const int *devices);
int           gpumap_multi_size(const gpumap_multi *multi);
int           gpumap_multi_used_rccl(const gpumap_multi *multi);   /* 1: the arrays went through ncclBroadcast */
/* Milliseconds of the set-up: i = -1 the broadcast of the arrays; i >= 1 what member i took to adopt them and derive
** its own tables (jump table, bit planes), the members working side by side. */
double        gpumap_multi_setup_ms(const gpumap_multi *multi, int i);
gpumap_ctx   *gpumap_multi_ctx(const gpumap_multi *multi, int i);
gpumap_ctx *const *gpumap_multi_ctxs(const gpumap_multi *multi);
int           gpumap_multi_map_batch(gpumap_multi *multi, const uint8_t *codes, const int64_t *offsets, int64_t n_reads,
                                     int tau, int want_intervals, gpumap_hits *out);
void          gpumap_multi_destroy(gpumap_multi *multi);

/* ------------------------------------------------------------------------------------------
** ann_build(kmer, tau, bwt, sar, threads) (src/index_ann.c:56-257), the reference's consumer of blocksc_trail: for
** every distinct k-mer of the text (in suffix-array order, ann_build_mt :500-609) the neighbours within tau mismatches,
** folded into one byte per position of the forward strand (hits_push :920-1071, neigh_push :664-729): bits 0-3 the
** class of the number of closest neighbours, bits 4-5 their distance - 1, bit 6 "mismatch positions known", bit 7
** "a closest neighbour of the k-mer that starts at an earlier marked position differs here".  `table` (host,
** txt_len / 2 bytes) = ann->info of the reference, byte for byte, also next to k-mers with N, where the reference's
** reuse of the last k-mer's search (`trail`, :552-570, src/blocksearch.c:41-43, :63-66, :73-86) extends hits that were
** made for other letters or another budget.  Options of the context:
**   "ann_threads" (default 1)      the `threads` argument of the reference whose job list is reproduced (it decides
**                                  which suffixes are visited and where the reuse starts afresh, :116-132)
**   "ann_trail_reuse" (default 1)  0 = every k-mer is searched afresh (blocksc_trail with trail = 0): the same table on
**                                  texts without N, independent of the order of the k-mers everywhere
** Needs a context that holds the text; 17 bytes of device memory per symbol of the text beside the index.
** stats (may be NULL):
** [0] k-mers searched, [1] extension steps (those the reference executes: none for blocks it takes from its tree),
** [2] hits, [3] updates of records, [4] microseconds on the device,
** [5] letters of the second jump table the build put in place for the k-mers' left blocks (0: none; it stays with the
** context and also serves reads whose leaf blocks the first table is too long for), [6] k-mers searched as two queries
** (their first leaf block with the letters of an earlier k-mer).
** gpumap_ann_query = ann_query (:278-347) on such a table (align_pos: room for kmer entries, may be NULL);
** gpumap_ann_file_write = ann_file_write (:384-434): the file index_read picks up as <base>.ann.<kmer>.<tau>.
** ---------------------------------------------------------------------------------------- */
int gpumap_ann_build(gpumap_ctx *ctx, int kmer, int tau, uint8_t *table, int64_t stats[8]);
int gpumap_ann_query(const uint8_t *table, int64_t size, int kmer, int64_t pos, int32_t *dist, int32_t *neigh_cnt,
                     int32_t *align_cnt, int32_t *align_pos);
int gpumap_ann_file_write(const char *filename, const uint8_t *table, int64_t size, int kmer, int tau);

/* ------------------------------------------------------------------------------------------
** mapper() (src/mapper.c:3-47, src/mapper.h:10): reads `qfile` (FASTQ / FASTA / raw, detected
** as io_stream_set_parser does, src/io.c:232-266) in chunks, maps each chunk on the GPU and
** writes one line per located hit to `out_path` (NULL or "-" = stdout):
**     tag \t name:pos:strand \t score
** The host cuts the file into chunks (the cut of io_stream_read_seq, src/io.c:196-198) and moves bytes; the chunks
** are parsed and the lines formatted on the device (GPUMAP_SRC_FILE | GPUMAP_PIPE_LINES).  Environment:
** GPUMAP_MAPPER_HOST_PARSE=1 parses and formats on the host instead (same output); GPUMAP_CHUNK_BYTES = bytes of input
** per chunk (default 256 MB, IO_DEFAULT_BUFSIZE, src/io.h:13).
** ---------------------------------------------------------------------------------------- */
int gpumap_mapper(const char *qfile, const gpumap_index *index, gpumap_ctx *ctx, int tau,
                  const char *out_path);
/* The same over replicas of the index on several GPUs (gpumap_multi_ctx): the chunks of the read file go round
** the GPUs, the output is in input order, byte for byte what one GPU writes. */
int gpumap_mapper_multi(const char *qfile, const gpumap_index *index, gpumap_ctx *const *ctxs, int n_ctx, int tau,
                        const char *out_path);

/* Options.  None changes a result, only where and how the work is done:
**   "waves_per_cu"   persistent waves per CU of the search kernel (0 = as many as are resident)
**   "spill_nodes"    per-wave spill capacity of the search deque, in nodes
**   "pinned_results" 1 = gpumap_batch_fetch hands out slices of two page-locked arenas of the batch, used in turn
**   "compact_results" 1 = located hits come back as sorted keys + CSR offsets (gpumap_hits.loc_key)
**   "jump_bytes"     most bytes the jump table may take (0 = a third of the free device memory); rebuilds the table
**   "hit_list_guess" first guess of the hit list's length in entries (0 = from the batch; a short guess costs a second search)
**   "jump_k"         letters of the jump table (0 = none, -1 = by the size of the text and the free memory)
**   "verify"         0 = every step on the Occ table; 1 = nodes of one suffix are finished on the text;
**                    2 = also nodes of two suffixes; 4 = also nodes of three and four; 8 / 16 = also nodes of up to
**                    eight / sixteen (measured slower than 4 at GRCh38 size); -1 = by the size of the text (the
**                    default: 4 from 2^26 symbols on, else 1)
**   "sort_all_bits"  1 = the sort of the located hits runs over every bit of the keys (default 0: it leaves out the
**                    score and up to three low bits of the position where that saves a pass; short runs are ordered after)
**   "sort_by_read"   1 = the sort of the located hits runs over the read's bits only and the hits of each read are put in
**                    order where they lie; 0 = never; -1 = the default: where nearly every read has one hit (located hits
**                    <= 1.25 x reads; with reads of hundreds of hits each the sort over the key is faster);
**                    2 = the sort runs over the key's top 16 bits (segments of consecutive reads) and a workgroup orders each
**                    segment in LDS (also part of the default: where reads have up to six hits on average and there are
**                    4 096 segments or more: GRCh38 size with 12.5 M reads, not chr21 size)
**   "seed_candidates" 1 (default) = a seed whose entry of the jump table holds a single suffix leaves the search as a candidate
**                    for the replay on the text at once; 0 = it takes a step on the Occ table first (same hits and step counts)
**   "verify_tile"    entries of the search's output list a workgroup of k_verify takes (0 = the default, 512; at most 1024) */
int gpumap_set_option(gpumap_ctx *ctx, const char *name, int64_t value);
/* The names of the index's sequences (gpumap_index_view.seq_name; txt_pos_to_str prints them, src/index_txt.c:505), for
** hit lines formatted on the device (GPUMAP_PIPE_LINES).  n_names = seq_cnt of the index. */
int gpumap_set_seq_names(gpumap_ctx *ctx, const char *const *names, int64_t n_names);
const char *gpumap_version(void);

#ifdef __cplusplus
}
#endif
#endif
