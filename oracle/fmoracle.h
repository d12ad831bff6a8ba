/*
** fmoracle.h -- TEST INFRASTRUCTURE ONLY.  CPU restatement of the query hot path of
** ezorita/mapper (block search over a bidirectional FM-index + suffix-array locate).
**
** Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
** The product (libgpumap.so) never links, loads or calls it.
**
** Parity status: PINNED.  tests/test_oracle_*.py check this restatement against
**   - the reference's own known-answer vectors (tests/golden/kat_*.json, transcribed as data
**     from test/test_index_bwt.c, test/test_index_sar.c, test/test_blocksearch.c,
**     test/test_index_txt.c, test/test_index.c), and
**   - hit dumps produced by the unmodified reference compiled in place (oracle/_ref/ref_driver,
**     fixtures under tests/golden/ made by tools/make_golden.py).
*/
#ifndef FMORACLE_H_
#define FMORACLE_H_
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_QUERY_PREFIX 0   /* src/index_bwt.h:22 */
#define ORC_QUERY_SUFFIX 1   /* src/index_bwt.h:23 */
#define ORC_ALIGN_WORDS  4   /* src/blocksearch.h:12 */
#define ORC_UNKNOWN_BASE 4   /* src/blocksearch.h:14 */

typedef struct orc_index orc_index_t;

typedef struct {
   int64_t  fp, rp, sz, dp;          /* struct bwtquery_t, src/index_bwt.c:16-22 */
   int64_t  score;                   /* spath_t.score, src/blocksearch.h:16-20 */
   uint64_t align[ORC_ALIGN_WORDS];  /* spath_t.align */
} orc_hit_t;

/* Result of a batch: same content as the "hit dump" written by oracle/ref_driver.c. */
typedef struct {
   int64_t  n_reads, n_hits, n_locs;
   int64_t *status;     /* [n_reads] 0 / -1 */
   int64_t *hits;       /* [n_hits][11]: read, fp, rp, sz, dp, score, align0..3, first_loc */
   int64_t *locs;       /* [n_locs] text positions */
   int64_t  n_ext;      /* number of all-symbol extension steps (bwt_query_all calls) */
   double   seconds;
} orc_batch_t;

/* Index files base.{sym,txt,sar,bwt} in the reference's on-disk format. */
orc_index_t *orc_index_open(const char *base);
/* Same index over arrays in memory (same layouts as the files); the caller keeps them alive. */
orc_index_t *orc_index_from_arrays(const uint64_t *occ, int64_t occ_length, int64_t mark_intv,
                                   const uint64_t *c, const uint64_t *sa, int64_t sar_bits,
                                   int64_t sar_len, int64_t txt_len, int64_t wil_cnt,
                                   const uint8_t *text);
void         orc_index_close(orc_index_t *idx);
int64_t      orc_txt_length(const orc_index_t *idx);
int64_t      orc_seq_count(const orc_index_t *idx);
int64_t      orc_sar_bits(const orc_index_t *idx);
int64_t      orc_wildcard_count(const orc_index_t *idx);
const uint64_t *orc_c_array(const orc_index_t *idx);

/* get_occ_all, src/index_bwt.c:840-905 */
void    orc_occ_all(const orc_index_t *idx, int64_t ptr, int64_t occ[5]);
/* get_occ, src/index_bwt.c:909-958 */
int64_t orc_occ(const orc_index_t *idx, int64_t ptr, int sym);
/* bwt_query_all, src/index_bwt.c:198-304: out[j] = {fp, rp, sz} for j = 0..4 (dp = dp_in + 1) */
void    orc_extend_all(const orc_index_t *idx, int end, int64_t fp, int64_t rp, int64_t sz,
                       int64_t out[5][3]);
/* bwt_prefix_all, src/index_bwt.c:390-446: out[j] = {fp, sz} */
void    orc_prefix_all(const orc_index_t *idx, int64_t fp, int64_t sz, int64_t out[5][2]);
/* sar_get / sar_get_range, src/index_sar.c:92-161 */
int64_t orc_sar_get(const orc_index_t *idx, int64_t pos);
int     orc_sar_get_range(const orc_index_t *idx, int64_t beg, int64_t size, int64_t *vec);
/* sym_str_index, src/index_sym.c:293-319 */
void    orc_encode(const orc_index_t *idx, const char *str, int64_t len, uint8_t *codes);
/* txt_pos_to_str, src/index_txt.c:480-511; returns strlen or -1 */
int     orc_pos_to_str(const orc_index_t *idx, int64_t pos, char *buf, size_t bufsz);
/* strand / coordinate rule alone (src/index_txt.c:492-496) */
void    orc_pos_to_coord(const orc_index_t *idx, int64_t pos, int64_t *seq_id, int64_t *seq_pos1,
                         int *strand);

/* blocksearch_trail_rec(query, 0, m-1, tau+1, 0, bwt, tree), src/blocksearch.c:106-177, with
** seqsearch_fw/bw (:184-311) and seqdash_fw/bw (:381-496).  Hits come back in the reference's
** order.  Returns the number of hits, or -1 where the reference returns -1. *hits is malloc'ed. */
int64_t orc_blocksearch(const orc_index_t *idx, const uint8_t *query, int m, int tau,
                        orc_hit_t **hits, int64_t *n_ext);
/* seqsearch_fw / seqsearch_bw from an arbitrary start path (for the KATs of
** test/test_blocksearch.c:294-434).  dir: ORC_QUERY_SUFFIX = fw, ORC_QUERY_PREFIX = bw. */
int64_t orc_seqsearch(const orc_index_t *idx, int dir, const orc_hit_t *start, const uint8_t *query,
                      int pos, int end, int tau, int score_ref, int score_diff, orc_hit_t **hits);

/* blocksc_trail(query, qarray, slen, tau, 0, tree), src/blocksearch.c:7-103, with scsearch_fw (:314-378): the
** "seeq & construct" search of ann_build_mt (src/index_ann.c:556-567): neighbours of the query that are
** lexicographically at or below it, query N lowering tau.  qarray is computed inside as the callers do.
** Hits in the reference's order; -1 where the reference returns -1. */
int64_t orc_blocksc(const orc_index_t *idx, const uint8_t *query, int slen, int tau,
                    orc_hit_t **hits, int64_t *n_ext);
/* scsearch_fw from an arbitrary start path (the KATs of test/test_blocksearch.c:497-594). */
int64_t orc_scsearch(const orc_index_t *idx, const orc_hit_t *start, const uint8_t *query, int pos, int end, int tau,
                     int score_ref, int score_diff, int boundary, orc_hit_t **hits);
/* orc_map_batch with orc_blocksc in place of orc_blocksearch. */
orc_batch_t *orc_blocksc_batch(const orc_index_t *idx, const uint8_t *codes, const int64_t *offsets,
                               int64_t n_reads, int tau, int threads);

/* ann_build(kmer, tau, bwt, sar, threads) (src/index_ann.c:56-257, workers :500-609, hits_push :920-1071) with the jobs
** run one after the other; table = ann->info, [txt_len / 2] bytes.  trail_reuse = 1: the reference's reuse of the last
** k-mer's search as it is (its table byte for byte); 0: every k-mer searched afresh (trail = 0). */
int orc_ann_build(const orc_index_t *idx, int kmer, int tau, int threads, int trail_reuse, uint8_t *table,
                  int64_t *n_queries, int64_t *n_ext, int64_t *dropped);

/* Whole batch, read-parallel with `threads` pthreads (static partition, private scratch; the
** model is ann_build_mt, src/index_ann.c:500-609).  codes: concatenated symbol codes,
** offsets[n_reads+1]. */
orc_batch_t *orc_map_batch(const orc_index_t *idx, const uint8_t *codes, const int64_t *offsets,
                           int64_t n_reads, int tau, int threads);
void         orc_batch_free(orc_batch_t *b);

#ifdef __cplusplus
}
#endif
#endif
