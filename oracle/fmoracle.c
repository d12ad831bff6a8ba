/*
** fmoracle.c -- TEST INFRASTRUCTURE ONLY (see fmoracle.h).  CPU restatement, in plain C, of the
** algorithm of ezorita/mapper's query hot path.  Every function cites the reference lines it
** follows; nothing here is used by, or linked into, the product library.
*/
#define _POSIX_C_SOURCE 200809L
#include "fmoracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <fcntl.h>
#include <unistd.h>
#include <pthread.h>
#include <sys/mman.h>
#include <sys/stat.h>

#define MAGIC_SYM 0x0fcb0fcb00010001ULL   /* src/index_sym.h:18 */
#define MAGIC_TXT 0x0fcb0fcb00020001ULL   /* src/index_txt.h:22 */
#define MAGIC_BWT 0x0fcb0fcb00030001ULL   /* src/index_bwt.h:19 */
#define MAGIC_SAR 0x0fcb0fcb00040001ULL   /* src/index_sar.h:19 */

typedef struct { void *ptr; size_t len; } map_t;

struct orc_index {
   map_t     m_sym, m_txt, m_sar, m_bwt;
   /* .sym (src/index_sym.c:361-429) */
   int       sym_cnt;
   char      canon[8];
   uint8_t   sym_table[256];
   uint8_t   com_table[8];
   /* .txt (src/index_txt.c:577-661) */
   int64_t   txt_len, seq_cnt, wil_cnt, rc_flag;
   const int64_t *seq_len, *seq_beg;
   const char   **seq_name;
   const uint8_t *text;
   /* .sar (src/index_sar.c:165-219) */
   int64_t   sar_bits, sar_len, sar_txt_len;
   const uint64_t *sa;
   /* .bwt (src/index_bwt.c:702-766) */
   int64_t   occ_length, mark_intv, word_size, mark_bits;
   const uint64_t *c, *occ;
};

/* ---------------------------------------------------------------- index files */

static int map_file(const char *base, const char *ext, map_t *m)
{
   char path[4096];
   snprintf(path, sizeof path, "%s.%s", base, ext);
   int fd = open(path, O_RDONLY);
   if (fd < 0) { fprintf(stderr, "fmoracle: cannot open %s\n", path); return -1; }
   struct stat sb;
   if (fstat(fd, &sb) != 0) { close(fd); return -1; }
   m->len = (size_t)sb.st_size;
   m->ptr = mmap(NULL, m->len, PROT_READ, MAP_SHARED, fd, 0);
   close(fd);
   if (m->ptr == MAP_FAILED) { m->ptr = NULL; return -1; }
   return 0;
}

orc_index_t *orc_index_open(const char *base)
{
   orc_index_t *x = calloc(1, sizeof *x);
   if (!x) return NULL;
   if (map_file(base, "sym", &x->m_sym) || map_file(base, "txt", &x->m_txt) ||
       map_file(base, "sar", &x->m_sar) || map_file(base, "bwt", &x->m_bwt)) goto fail;

   /* .sym: u64 magic, u8 count, char canon[count+1], u8 table[256], u8 com[count+1] */
   const uint8_t *s = x->m_sym.ptr;
   uint64_t magic; memcpy(&magic, s, 8);
   if (magic != MAGIC_SYM || x->m_sym.len < 8 + 1) goto fail;
   x->sym_cnt = s[8];
   if (x->sym_cnt != 5 || x->m_sym.len < (size_t)(9 + 6 + 256 + 6)) goto fail;
   memcpy(x->canon, s + 9, x->sym_cnt + 1);
   memcpy(x->sym_table, s + 9 + x->sym_cnt + 1, 256);
   memcpy(x->com_table, s + 9 + x->sym_cnt + 1 + 256, x->sym_cnt + 1);

   /* .txt: u64 magic, i64 txt_len seq_cnt wil_cnt rc_flag, seq_len[], seq_beg[], names, text */
   const int64_t *t = x->m_txt.ptr;
   if ((uint64_t)t[0] != MAGIC_TXT) goto fail;
   x->txt_len = t[1]; x->seq_cnt = t[2]; x->wil_cnt = t[3]; x->rc_flag = t[4];
   x->seq_len = t + 5;
   x->seq_beg = x->seq_len + x->seq_cnt;
   x->seq_name = calloc(x->seq_cnt, sizeof(char *));
   const char *p = (const char *)(x->seq_beg + x->seq_cnt);
   for (int64_t i = 0; i < x->seq_cnt; i++) { x->seq_name[i] = p; p += strlen(p) + 1; }
   x->text = (const uint8_t *)p;

   /* .sar: u64 magic, i64 bits, len, txt_len, i64 sa[len] */
   const int64_t *a = x->m_sar.ptr;
   if ((uint64_t)a[0] != MAGIC_SAR) goto fail;
   x->sar_bits = a[1]; x->sar_len = a[2]; x->sar_txt_len = a[3];
   x->sa = (const uint64_t *)(a + 4);

   /* .bwt: u64 magic, i64 occ_length, mark_intv, word_size, mark_bits, u64 c[6], u64 occ[] */
   const uint64_t *b = x->m_bwt.ptr;
   if (b[0] != MAGIC_BWT) goto fail;
   x->occ_length = (int64_t)b[1]; x->mark_intv = (int64_t)b[2];
   x->word_size = (int64_t)b[3]; x->mark_bits = (int64_t)b[4];
   x->c = b + 5;
   x->occ = x->c + x->sym_cnt + 1;
   if (x->word_size != 64) goto fail;
   return x;
fail:
   fprintf(stderr, "fmoracle: bad or missing index '%s'\n", base);
   orc_index_close(x);
   return NULL;
}

/* Index over arrays the caller owns, laid out as in the .bwt / .sar files (for indexes that never
** existed as files: the GRCh38-size synthetic index of bench.py).  One unnamed sequence; the text
** is optional (the search does not read it). */
orc_index_t *orc_index_from_arrays(const uint64_t *occ, int64_t occ_length, int64_t mark_intv,
                                   const uint64_t *c, const uint64_t *sa, int64_t sar_bits,
                                   int64_t sar_len, int64_t txt_len, int64_t wil_cnt,
                                   const uint8_t *text)
{
   static const int64_t zero = 0;
   static const char *noname = "seq";
   orc_index_t *x = calloc(1, sizeof *x);
   if (!x) return NULL;
   x->sym_cnt = 5;
   memcpy(x->canon, "ACGTN", 6);
   memset(x->sym_table, 4, 256);
   {
      const char *al[5] = {"Aa", "Cc", "Gg", "Tt", "Nn"};   /* src/index_sym.c:11 */
      const uint8_t com[6] = {3, 2, 1, 0, 4, 5};             /* src/index_sym.c:12 */
      for (int i = 0; i < 5; i++) for (const char *q = al[i]; *q; q++) x->sym_table[(uint8_t)*q] = (uint8_t)i;
      memcpy(x->com_table, com, 6);
   }
   x->txt_len = txt_len; x->seq_cnt = 1; x->wil_cnt = wil_cnt; x->rc_flag = 1;
   x->seq_len = &x->txt_len; x->seq_beg = &zero;
   x->seq_name = calloc(1, sizeof(char *));
   x->seq_name[0] = noname;
   x->text = text;
   x->sar_bits = sar_bits; x->sar_len = sar_len; x->sar_txt_len = txt_len; x->sa = sa;
   x->occ_length = occ_length; x->mark_intv = mark_intv; x->word_size = 64; x->mark_bits = 64 * mark_intv;
   x->c = c; x->occ = occ;
   return x;
}

void orc_index_close(orc_index_t *x)
{
   if (!x) return;
   if (x->m_sym.ptr) munmap(x->m_sym.ptr, x->m_sym.len);
   if (x->m_txt.ptr) munmap(x->m_txt.ptr, x->m_txt.len);
   if (x->m_sar.ptr) munmap(x->m_sar.ptr, x->m_sar.len);
   if (x->m_bwt.ptr) munmap(x->m_bwt.ptr, x->m_bwt.len);
   free((void *)x->seq_name);
   free(x);
}

int64_t orc_txt_length(const orc_index_t *x) { return x->txt_len; }
int64_t orc_seq_count(const orc_index_t *x) { return x->seq_cnt; }
int64_t orc_sar_bits(const orc_index_t *x) { return x->sar_bits; }
int64_t orc_wildcard_count(const orc_index_t *x) { return x->wil_cnt; }
const uint64_t *orc_c_array(const orc_index_t *x) { return x->c; }

/* ---------------------------------------------------------------- rank */

/* Restates get_occ_all (src/index_bwt.c:840-905): counts of the five symbols in BWT[0..ptr].
** The array interleaves one "mark" group (5 absolute counts) per `mark_intv` word groups
** (5 bit-planes of 64 positions, MSB first).  The nearest mark is used: popcounts are added
** when the word lies after the mark and subtracted when it lies before it. */
void orc_occ_all(const orc_index_t *x, int64_t ptr, int64_t occ[5])
{
   const int S = 5;
   if (ptr == -1) {                       /* :858-861 */
      for (int j = 0; j < S; j++) occ[j] = (int64_t)x->occ[j];
      return;
   }
   int64_t wrdnum = ptr / 64;
   int64_t wrdptr = (wrdnum + wrdnum / x->mark_intv + 1) * S;                      /* :865 */
   int64_t mrkptr = ((wrdnum + x->mark_intv / 2) / x->mark_intv) * (x->mark_intv + 1) * S; /* :866 */
   int     bit    = (int)(ptr % 64);
   int64_t off[5] = {0, 0, 0, 0, 0};
   if (wrdptr > mrkptr) {                 /* :872-883 */
      for (int64_t i = mrkptr + S; i < wrdptr; i += S)
         for (int j = 0; j < S; j++) off[j] += __builtin_popcountll(x->occ[i + j]);
      for (int j = 0; j < S; j++) off[j] += __builtin_popcountll(x->occ[wrdptr + j] >> (63 - bit));
      for (int j = 0; j < S; j++) occ[j] = (int64_t)x->occ[mrkptr + j] + off[j];
   } else {                               /* :884-897 */
      if (bit < 63)
         for (int j = 0; j < S; j++) off[j] += __builtin_popcountll(x->occ[wrdptr + j] << (bit + 1));
      for (int64_t i = wrdptr + S; i < mrkptr; i += S)
         for (int j = 0; j < S; j++) off[j] += __builtin_popcountll(x->occ[i + j]);
      for (int j = 0; j < S; j++) occ[j] = (int64_t)x->occ[mrkptr + j] - off[j];
   }
}

/* get_occ (src/index_bwt.c:909-958): the single-symbol form, same addressing. */
int64_t orc_occ(const orc_index_t *x, int64_t ptr, int sym)
{
   int64_t all[5];
   orc_occ_all(x, ptr, all);
   return all[sym];
}

/* ---------------------------------------------------------------- bidirectional extension */

/* Restates bwt_query_all (src/index_bwt.c:198-304). */
void orc_extend_all(const orc_index_t *x, int end, int64_t q_fp, int64_t q_rp, int64_t q_sz,
                    int64_t out[5][3])
{
   const int S = 5;
   if (end == ORC_QUERY_SUFFIX) { int64_t t = q_fp; q_fp = q_rp; q_rp = t; }   /* :235-238 */
   int64_t sp[5], ep[5], fp[5], rp[5], sz[5], tot = 0;
   orc_occ_all(x, q_fp - 1, sp);                                              /* :257 */
   orc_occ_all(x, q_fp + q_sz - 1, ep);                                       /* :258 */
   for (int j = 0; j < S; j++) {                                              /* :262-267 */
      fp[j] = (int64_t)x->c[j] + sp[j];
      sz[j] = ep[j] - sp[j];
      tot  += sz[j];
   }
   /* :270-273  reverse pointers accumulate in the order comp(0), comp(1), ... */
   const uint8_t *cm = x->com_table;
   rp[cm[0]] = q_rp + (q_sz - tot);
   for (int j = 1; j < S; j++) rp[cm[j]] = rp[cm[j - 1]] + sz[cm[j - 1]];
   if (end == ORC_QUERY_PREFIX) {                                             /* :275-279 */
      for (int j = 0; j < S; j++) { out[j][0] = fp[j]; out[j][1] = rp[j]; out[j][2] = sz[j]; }
   } else {                                                                   /* :280-286 */
      for (int j = 0; j < S; j++) {
         int c = cm[j];
         out[j][0] = rp[c]; out[j][1] = fp[c]; out[j][2] = sz[c];
      }
   }
}

/* Restates bwt_prefix_all (src/index_bwt.c:390-446). */
void orc_prefix_all(const orc_index_t *x, int64_t fp, int64_t sz, int64_t out[5][2])
{
   int64_t sp[5], ep[5];
   orc_occ_all(x, fp - 1, sp);
   orc_occ_all(x, fp + sz - 1, ep);
   for (int j = 0; j < 5; j++) {
      out[j][0] = (int64_t)x->c[j] + sp[j];
      int64_t s = (int64_t)x->c[j] + ep[j] - out[j][0];
      out[j][1] = s < 0 ? 0 : s;
   }
}

/* ---------------------------------------------------------------- suffix array */

/* Restates sar_get (src/index_sar.c:92-118): entry i occupies bits [i*b, (i+1)*b) of the
** little-endian bit stream and may straddle two words. */
int64_t orc_sar_get(const orc_index_t *x, int64_t pos)
{
   if (pos < 0 || pos >= x->sar_txt_len) return -1;
   int b = (int)x->sar_bits;
   uint64_t mask = ~(uint64_t)0 >> (64 - b);
   uint64_t bit = (uint64_t)pos * (uint64_t)b, w = bit / 64;
   bit %= 64;
   uint64_t v = x->sa[w] >> bit;
   if (bit + b > 64) v |= x->sa[w + 1] << (64 - bit);
   return (int64_t)(v & mask);
}

/* sar_get_range (src/index_sar.c:121-161): same values as `size` successive sar_get calls. */
int orc_sar_get_range(const orc_index_t *x, int64_t beg, int64_t size, int64_t *vec)
{
   if (!vec || beg < 0 || size < 1 || beg + size > x->sar_txt_len) return -1;
   for (int64_t i = 0; i < size; i++) vec[i] = orc_sar_get(x, beg + i);
   return 0;
}

/* ---------------------------------------------------------------- alphabet and coordinates */

void orc_encode(const orc_index_t *x, const char *str, int64_t len, uint8_t *codes)
{
   for (int64_t i = 0; i < len; i++) codes[i] = x->sym_table[(uint8_t)str[i]];   /* :276-319 */
}

/* seq_bisect (src/index_txt.c:770-787), iteratively. */
static int64_t bisect(const int64_t *vals, int64_t beg, int64_t end, int64_t target)
{
   while (end - beg >= 2) {
      int64_t mid = (beg + end) / 2;
      if (target < vals[mid]) end = mid; else beg = mid;
   }
   return beg;
}

void orc_pos_to_coord(const orc_index_t *x, int64_t pos, int64_t *seq_id, int64_t *seq_pos1, int *strand)
{
   int st = 0;
   if (x->rc_flag) {                                     /* src/index_txt.c:492-496 */
      st = pos >= x->txt_len / 2;
      if (st) pos = x->txt_len - 2 - pos;
   }
   int64_t id = bisect(x->seq_beg, 0, x->seq_cnt, pos);  /* :499 */
   *seq_id = id; *seq_pos1 = pos - x->seq_beg[id] + 1; *strand = st;
}

int orc_pos_to_str(const orc_index_t *x, int64_t pos, char *buf, size_t bufsz)
{
   if (pos < 0 || pos >= x->txt_len) return -1;
   int64_t id, p1; int st;
   orc_pos_to_coord(x, pos, &id, &p1, &st);
   return snprintf(buf, bufsz, "%s:%ld:%c", x->seq_name[id], (long)p1, st ? '-' : '+');   /* :505 */
}

/* ---------------------------------------------------------------- block search */

typedef struct { orc_hit_t *v; int64_t n, cap; } hitvec_t;

typedef struct {
   const orc_index_t *x;
   const uint8_t     *q;
   int64_t            n_ext;
   int                err;
} sctx_t;

static void hv_push(sctx_t *c, hitvec_t *hv, const orc_hit_t *h)   /* path_push, :600-624 */
{
   if (hv->n >= hv->cap) {
      int64_t nc = hv->cap ? 2 * hv->cap : 64;    /* PATHSTACK_DEF_SIZE, doubling */
      orc_hit_t *nv = realloc(hv->v, nc * sizeof *nv);
      if (!nv) { c->err = 1; return; }
      hv->v = nv; hv->cap = nc;
   }
   hv->v[hv->n++] = *h;
}

static inline void bit_set(orc_hit_t *h, int pos) { h->align[pos / 64] |= (uint64_t)1 << (pos % 64); } /* :631-642 */

/* seqdash_fw / seqdash_bw (src/blocksearch.c:381-496): exact continuation; a query N follows
** every non-empty branch and flags its position. step = +1 (fw, SUFFIX) or -1 (bw, PREFIX). */
static void dash(sctx_t *c, orc_hit_t p, int pos, int end, int step, hitvec_t *hits)
{
   int dir = step > 0 ? ORC_QUERY_SUFFIX : ORC_QUERY_PREFIX;
   int64_t qv[5][3];
   for (int d = pos; step > 0 ? d <= end : d >= end; d += step) {
      orc_extend_all(c->x, dir, p.fp, p.rp, p.sz, qv);
      c->n_ext++;
      int s = c->q[d];
      if (s != ORC_UNKNOWN_BASE) {
         p.fp = qv[s][0]; p.rp = qv[s][1]; p.sz = qv[s][2]; p.dp++;
         if (p.sz < 1) return;
      } else {
         bit_set(&p, d);
         for (int i = 0; i < 5; i++) {
            if (qv[i][2] < 1) continue;
            orc_hit_t b = p;
            b.fp = qv[i][0]; b.rp = qv[i][1]; b.sz = qv[i][2]; b.dp = p.dp + 1;
            dash(c, b, d + step, end, step, hits);
         }
         return;
      }
   }
   hv_push(c, hits, &p);
}

/* seqsearch_fw / seqsearch_bw (src/blocksearch.c:184-311). */
static void search(sctx_t *c, const orc_hit_t *path, int pos, int end, int step, int tau,
                   int score_ref, int score_diff, hitvec_t *hits)
{
   int dir = step > 0 ? ORC_QUERY_SUFFIX : ORC_QUERY_PREFIX;
   int64_t qv[5][3];
   orc_extend_all(c->x, dir, path->fp, path->rp, path->sz, qv);
   c->n_ext++;
   int qs = c->q[pos];
   for (int nt = 0; nt < 5; nt++) {
      if (qv[nt][2] < 1) continue;
      int ds = (nt != qs && qs != ORC_UNKNOWN_BASE);
      int s  = (int)path->score + ds;
      if (s > tau) continue;
      orc_hit_t p = *path;
      p.fp = qv[nt][0]; p.rp = qv[nt][1]; p.sz = qv[nt][2]; p.dp = path->dp + 1; p.score = s;
      if (ds || qs == ORC_UNKNOWN_BASE) bit_set(&p, pos);
      if (s == tau) {
         if (s - score_ref >= score_diff) dash(c, p, pos + step, end, step, hits);
      } else if (pos == end) {
         if (s - score_ref >= score_diff) hv_push(c, hits, &p);
      } else {
         search(c, &p, pos + step, end, step, tau, score_ref, score_diff, hits);
      }
   }
}

/* blocksearch_trail_rec with trail = 0 (src/blocksearch.c:106-177). */
static int block_rec(sctx_t *c, int pos, int end, int blocks, hitvec_t *out)
{
   if (pos < 0 || pos > end || end < 0 || blocks < 0) return -1;     /* :125-128 */
   out->n = 0;
   if (blocks == 1) {                                                /* :138-144 */
      orc_hit_t root;
      memset(&root, 0, sizeof root);
      root.sz = c->x->txt_len;
      search(c, &root, end, pos, -1, 0, 0, 0, out);
      return 0;
   }
   int blk_l = (blocks >> 1) + (blocks & 1), blk_r = blocks >> 1;   /* :147-148 */
   int slen = end - pos + 1;
   int pos_r = pos + (slen >> 1) + (slen & 1), end_l = pos_r - 1;    /* :151-153 */
   hitvec_t sub = {0};
   int rc = block_rec(c, pos, end_l, blk_l, &sub);                   /* :157 */
   if (rc == 0) {
      for (int64_t i = 0; i < sub.n; i++)                            /* :159-162 */
         search(c, &sub.v[i], pos_r, end, +1, blocks - 1, (int)sub.v[i].score, 0, out);
      rc = block_rec(c, pos_r, end, blk_r, &sub);                    /* :165 */
   }
   if (rc == 0) {
      for (int64_t i = 0; i < sub.n; i++)                            /* :167-170 */
         search(c, &sub.v[i], end_l, pos, -1, blocks - 1, (int)sub.v[i].score, blk_l, out);
   }
   free(sub.v);
   return rc;
}

int64_t orc_blocksearch(const orc_index_t *x, const uint8_t *query, int m, int tau,
                        orc_hit_t **hits, int64_t *n_ext)
{
   sctx_t c = { x, query, 0, 0 };
   hitvec_t out = {0};
   *hits = NULL;
   if (!query || m < 1 || tau < 0) return -1;
   int rc = block_rec(&c, 0, m - 1, tau + 1, &out);
   if (n_ext) *n_ext = c.n_ext;
   if (rc != 0 || c.err) { free(out.v); return -1; }
   *hits = out.v;
   return out.n;
}

/* scsearch_fw (src/blocksearch.c:314-378): seqsearch_fw restricted to extensions that stay lexicographically
** at or below the query: while `boundary` holds (every letter so far equal to the query's), only the symbols
** nt <= query[pos] are followed (:338), and the flag survives only through nt == query[pos] (:366).  The dash
** at s == tau is the plain seqdash_fw (:355-358). */
static void scsearch(sctx_t *c, const orc_hit_t *path, int pos, int end, int tau,
                     int score_ref, int score_diff, int boundary, hitvec_t *hits)
{
   int64_t qv[5][3];
   orc_extend_all(c->x, ORC_QUERY_SUFFIX, path->fp, path->rp, path->sz, qv);
   c->n_ext++;
   int qs = c->q[pos];
   int lim = boundary ? qs + 1 : 5;                                  /* :338 */
   for (int nt = 0; nt < lim; nt++) {
      if (qv[nt][2] < 1) continue;
      int ds = (nt != qs && qs != ORC_UNKNOWN_BASE);                 /* :345 */
      int s  = (int)path->score + ds;
      if (s > tau) continue;
      orc_hit_t p = *path;
      p.fp = qv[nt][0]; p.rp = qv[nt][1]; p.sz = qv[nt][2]; p.dp = path->dp + 1; p.score = s;
      if (ds || qs == ORC_UNKNOWN_BASE) bit_set(&p, pos);
      if (s == tau) {
         if (s - score_ref >= score_diff) dash(c, p, pos + 1, end, +1, hits);
      } else if (pos == end) {
         if (s - score_ref >= score_diff) hv_push(c, hits, &p);
      } else {
         scsearch(c, &p, pos + 1, end, tau, score_ref, score_diff, boundary && nt == qs, hits);   /* :366-367 */
      }
   }
}

int64_t orc_scsearch(const orc_index_t *x, const orc_hit_t *start, const uint8_t *query, int pos, int end, int tau,
                     int score_ref, int score_diff, int boundary, orc_hit_t **hits)
{
   sctx_t c = { x, query, 0, 0 };
   hitvec_t out = {0};
   scsearch(&c, start, pos, end, tau, score_ref, score_diff, boundary, &out);
   *hits = out.v;
   return c.err ? -1 : out.n;
}

/* blocksc_trail (src/blocksearch.c:7-103) with trail = 0, the path array `qarray` computed as its callers do
** (one bwt_query SUFFIX per letter from the root interval: src/index_ann.c:791, test/test_blocksearch.c:49-53;
** those steps are the caller's and are not counted in n_ext). */
int64_t orc_blocksc(const orc_index_t *x, const uint8_t *query, int slen, int tau,
                    orc_hit_t **hits, int64_t *n_ext)
{
   sctx_t c = { x, query, 0, 0 };
   hitvec_t out = {0}, left = {0};
   *hits = NULL;
   if (n_ext) *n_ext = 0;
   if (!query || slen < 1 || tau < 0) return -1;                     /* :34-39 */
   int64_t (*qa)[3] = malloc((size_t)(slen + 1) * sizeof *qa);
   if (!qa) return -1;
   qa[0][0] = 0; qa[0][1] = 0; qa[0][2] = x->txt_len;               /* bwt_new_query, src/index_bwt.c:48-52 */
   for (int i = 0; i < slen; i++) {
      int64_t qv[5][3];
      orc_extend_all(x, ORC_QUERY_SUFFIX, qa[i][0], qa[i][1], qa[i][2], qv);
      memcpy(qa[i + 1], qv[query[i]], sizeof qv[0]);
   }
   int last_fragment = qa[slen][0] >= qa[slen][1];                   /* :55 */
   int tau_arg = tau;
   int n_cnt = 0;
   for (int i = 0; i < slen && n_cnt <= tau; i++) n_cnt += query[i] == ORC_UNKNOWN_BASE;   /* :59-60 */
   tau -= n_cnt;
   if (tau < 0 || (tau == 0 && !last_fragment)) { free(qa); return 0; }                     /* :63-66 */
   /* The tree comes from the caller's alloc_stack_tree(tau) (src/index_ann.c:541-543, test/test_blocksearch.c:38): for
   ** tau = 0 it is a single node without a left subtree, and the left search returns -1 (:74 -> :123). */
   if (tau_arg == 0) { free(qa); return -1; }
   int pos_r = slen / 2 + (last_fragment ? slen % 2 : 0);            /* :69 */
   /* an empty right part (one letter, last fragment): scsearch_fw (:91) would read query[slen], undefined in the
   ** reference; reported as an error here and by the HIP path */
   if (pos_r > slen - 1) { free(qa); return -1; }
   int tau_l = tau / 2 - (last_fragment ? 0 : (1 - tau % 2));        /* :70 */
   int rc = 0;
   if (0 < pos_r) rc = block_rec(&c, 0, pos_r - 1, tau_l + 1, &left);   /* :73-74 (trail = 0): an empty left block is not searched, a fresh tree holds no hits */
   if (rc == 0) {
      int64_t max_sa_pos = qa[pos_r][0] + qa[pos_r][2];              /* :76 */
      int64_t i = 0;
      while (i < left.n) {                                            /* :77-85: out-of-boundary hits go, the last one takes the slot */
         if (left.v[i].fp < max_sa_pos) i++;
         else left.v[i] = left.v[--left.n];
      }
      for (i = 0; i < left.n; i++)                                    /* :89-92 */
         scsearch(&c, &left.v[i], pos_r, slen - 1, tau, (int)left.v[i].score, 0, 1, &out);
      if (n_cnt)
         for (i = 0; i < out.n; i++) out.v[i].score += n_cnt;        /* :95-97 */
   }
   free(left.v);
   free(qa);
   if (n_ext) *n_ext = c.n_ext;
   if (rc != 0 || c.err) { free(out.v); return -1; }
   *hits = out.v;
   return out.n;
}

int64_t orc_seqsearch(const orc_index_t *x, int dir, const orc_hit_t *start, const uint8_t *query,
                      int pos, int end, int tau, int score_ref, int score_diff, orc_hit_t **hits)
{
   sctx_t c = { x, query, 0, 0 };
   hitvec_t out = {0};
   search(&c, start, pos, end, dir == ORC_QUERY_SUFFIX ? +1 : -1, tau, score_ref, score_diff, &out);
   *hits = out.v;
   return c.err ? -1 : out.n;
}

/* ---------------------------------------------------------------- batch driver (CPU baseline) */

typedef struct {
   const orc_index_t *x;
   const uint8_t *codes;
   const int64_t *offsets;
   int64_t beg, end;
   int tau;
   int sc;              /* 0: blocksearch_trail_rec per read, 1: blocksc_trail per read */
   int64_t *status;
   int64_t n_hits, n_locs, cap_hits, cap_locs, n_ext;
   int64_t *hits, *locs;
   int err;
} bjob_t;

static void *batch_worker(void *arg)
{
   bjob_t *j = arg;
   for (int64_t r = j->beg; r < j->end; r++) {
      int64_t m = j->offsets[r + 1] - j->offsets[r];
      orc_hit_t *h = NULL;
      int64_t ext = 0;
      int64_t nh = -1;
      if (m >= 1 && m <= 256)
         nh = j->sc ? orc_blocksc(j->x, j->codes + j->offsets[r], (int)m, j->tau, &h, &ext)
                    : orc_blocksearch(j->x, j->codes + j->offsets[r], (int)m, j->tau, &h, &ext);
      if (nh < 0) { j->status[r] = -1; continue; }   /* steps of a failed search are not counted */
      j->n_ext += ext;
      j->status[r] = 0;
      int64_t nl = 0;
      for (int64_t i = 0; i < nh; i++) nl += h[i].sz;
      if (j->n_hits + nh > j->cap_hits) {
         while (j->n_hits + nh > j->cap_hits) j->cap_hits = j->cap_hits ? 2 * j->cap_hits : 4096;
         j->hits = realloc(j->hits, j->cap_hits * 11 * sizeof(int64_t));
         if (!j->hits) { j->err = 1; free(h); return NULL; }
      }
      if (j->n_locs + nl > j->cap_locs) {
         while (j->n_locs + nl > j->cap_locs) j->cap_locs = j->cap_locs ? 2 * j->cap_locs : 4096;
         j->locs = realloc(j->locs, j->cap_locs * sizeof(int64_t));
         if (!j->locs) { j->err = 1; free(h); return NULL; }
      }
      for (int64_t i = 0; i < nh; i++) {
         int64_t *o = j->hits + 11 * j->n_hits;
         o[0] = r; o[1] = h[i].fp; o[2] = h[i].rp; o[3] = h[i].sz; o[4] = h[i].dp; o[5] = h[i].score;
         for (int w = 0; w < 4; w++) o[6 + w] = (int64_t)h[i].align[w];
         o[10] = j->n_locs;
         orc_sar_get_range(j->x, h[i].fp, h[i].sz, j->locs + j->n_locs);
         j->n_locs += h[i].sz;
         j->n_hits++;
      }
      free(h);
   }
   return NULL;
}

static orc_batch_t *batch_run(const orc_index_t *x, const uint8_t *codes, const int64_t *offsets,
                              int64_t n_reads, int tau, int threads, int sc);

orc_batch_t *orc_map_batch(const orc_index_t *x, const uint8_t *codes, const int64_t *offsets,
                           int64_t n_reads, int tau, int threads)
{
   return batch_run(x, codes, offsets, n_reads, tau, threads, 0);
}

orc_batch_t *orc_blocksc_batch(const orc_index_t *x, const uint8_t *codes, const int64_t *offsets,
                               int64_t n_reads, int tau, int threads)
{
   return batch_run(x, codes, offsets, n_reads, tau, threads, 1);
}

static orc_batch_t *batch_run(const orc_index_t *x, const uint8_t *codes, const int64_t *offsets,
                              int64_t n_reads, int tau, int threads, int sc)
{
   if (threads < 1) threads = 1;
   if (threads > n_reads) threads = n_reads > 0 ? (int)n_reads : 1;
   orc_batch_t *b = calloc(1, sizeof *b);
   bjob_t *jobs = calloc(threads, sizeof *jobs);
   pthread_t *tid = calloc(threads, sizeof *tid);
   b->n_reads = n_reads;
   b->status = calloc(n_reads > 0 ? n_reads : 1, sizeof(int64_t));
   struct timespec t0, t1;
   clock_gettime(CLOCK_MONOTONIC, &t0);
   for (int t = 0; t < threads; t++) {
      jobs[t].x = x; jobs[t].codes = codes; jobs[t].offsets = offsets; jobs[t].tau = tau; jobs[t].sc = sc;
      jobs[t].status = b->status;
      jobs[t].beg = n_reads * t / threads; jobs[t].end = n_reads * (t + 1) / threads;
      pthread_create(&tid[t], NULL, batch_worker, &jobs[t]);
   }
   for (int t = 0; t < threads; t++) pthread_join(tid[t], NULL);
   clock_gettime(CLOCK_MONOTONIC, &t1);
   b->seconds = (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
   for (int t = 0; t < threads; t++) { b->n_hits += jobs[t].n_hits; b->n_locs += jobs[t].n_locs; b->n_ext += jobs[t].n_ext; }
   b->hits = malloc((b->n_hits > 0 ? b->n_hits : 1) * 11 * sizeof(int64_t));
   b->locs = malloc((b->n_locs > 0 ? b->n_locs : 1) * sizeof(int64_t));
   int64_t hb = 0, lb = 0;
   for (int t = 0; t < threads; t++) {
      for (int64_t i = 0; i < jobs[t].n_hits; i++) jobs[t].hits[11 * i + 10] += lb;
      if (jobs[t].n_hits) memcpy(b->hits + 11 * hb, jobs[t].hits, jobs[t].n_hits * 11 * sizeof(int64_t));
      if (jobs[t].n_locs) memcpy(b->locs + lb, jobs[t].locs, jobs[t].n_locs * sizeof(int64_t));
      hb += jobs[t].n_hits; lb += jobs[t].n_locs;
      free(jobs[t].hits); free(jobs[t].locs);
   }
   free(jobs); free(tid);
   return b;
}

void orc_batch_free(orc_batch_t *b)
{
   if (!b) return;
   free(b->status); free(b->hits); free(b->locs); free(b);
}

/* ---------------------------------------------------------------- neighbour annotation (ann_build) */

#define ANN_NO_INFO 0xFFFF                       /* src/index_ann.h:25 */
#define ANN_WORD    6                            /* word_size = max(3, tau) + 3 with tau < 4, src/index_ann.c:78, :86 */
#define ANN_ALN     3                            /* word_size - 3 */

static uint16_t rd16(const uint8_t *p) { uint16_t v; memcpy(&v, p, 2); return v; }
static void     wr16(uint8_t *p, uint16_t v) { memcpy(p, &v, 2); }

/* aln_positions, src/index_ann.c:882-917 */
static void ann_aln_positions(const uint64_t *bits, uint8_t *pos, int nbits, int npos, int reverse)
{
   uint8_t tmp[ANN_ALN + 1] = {0};
   int a = 0;
   for (int i = 0; i < nbits && a <= npos; i++)
      if ((bits[i / 64] >> (i % 64)) & 1) tmp[a++] = (uint8_t)(reverse ? nbits - i : i + 1);
   if (a > npos) memset(pos, 0xFF, npos);
   else memcpy(pos, tmp, npos);
}

/* aln_merge, src/index_ann.c:833-879 */
static void ann_aln_merge(uint8_t *a, const uint8_t *b, int len)
{
   uint8_t tmp[2 * ANN_ALN + 2] = {0};
   int i = 0, j = 0, k = 0;
   while (i < len && j < len && k <= len) {
      if (a[i] == 0 || b[j] == 0) break;
      if (a[i] == b[j]) { tmp[k++] = a[i]; i++; j++; }
      else if (a[i] < b[j]) tmp[k++] = a[i++];
      else tmp[k++] = b[j++];
   }
   while (i < len && k <= len && a[i]) tmp[k++] = a[i++];
   while (j < len && k <= len && b[j]) tmp[k++] = b[j++];
   while (k < len) tmp[k++] = 0;
   if (k > len) memset(a, 0xFF, len);
   else memcpy(a, tmp, len);
}

/* hits_push, src/index_ann.c:920-1071: `hits` = tree->stack after blocksc_trail, (fp, rp) = path[kmer]. */
static void ann_hits_push(uint8_t *info, int kmer, int tau_arg, const orc_hit_t *hits, int64_t n_hits, int64_t fp, int64_t rp)
{
   int nh = 0, tau = tau_arg + 1;
   if (fp > rp) wr16(info + fp * ANN_WORD, ANN_NO_INFO);                          /* :943-947 */
   const int64_t lo = fp < rp ? fp : rp;
   if (n_hits < 2) {                                                             /* :950-957 */
      if (rd16(info + lo * ANN_WORD) == 0) wr16(info + lo * ANN_WORD, ANN_NO_INFO);
      return;
   }
   uint64_t qalign[ORC_ALIGN_WORDS] = {0};
   for (int64_t i = 0; i < n_hits; i++) {                                        /* :964-1027 */
      const orc_hit_t *p = &hits[i];
      if (p->fp == fp) continue;                                                 /* the self hit */
      const int64_t nptr = p->rp < p->fp ? p->rp : p->fp;
      const int nrev = p->rp < p->fp;
      uint8_t *ninfo = info + nptr * ANN_WORD, *ntau = ninfo + 2, *naln = ntau + 1;
      const uint16_t ncnt = rd16(ninfo);
      if (*ntau == p->score) {                                                   /* :987-998 */
         const uint32_t c = (uint32_t)ncnt + 1;
         wr16(ninfo, (uint16_t)(c < 0xFFFE ? c : 0xFFFE));
         if (*naln != 0xFF) {
            uint8_t hit_aln[ANN_ALN];
            ann_aln_positions(p->align, hit_aln, kmer, ANN_ALN, nrev);
            ann_aln_merge(naln, hit_aln, ANN_ALN);
         }
      } else if (ncnt == 0 || ncnt == ANN_NO_INFO || *ntau > p->score) {           /* :1000-1009 */
         wr16(ninfo, 1);
         *ntau = (uint8_t)p->score;
         uint8_t hit_aln[ANN_ALN];
         ann_aln_positions(p->align, hit_aln, kmer, ANN_ALN, nrev);
         memcpy(naln, hit_aln, ANN_ALN);
      }
      if (p->score == tau) {                                                     /* :1014-1026 */
         for (int w = 0; w < ORC_ALIGN_WORDS; w++) qalign[w] |= p->align[w];
         nh++;
      } else if (p->score < tau) {
         memcpy(qalign, p->align, sizeof qalign);
         tau = (int)p->score;
         nh = 1;
      }
   }
   const int qrev = rp < fp;                                                     /* :1030-1061 */
   uint8_t *qi = info + lo * ANN_WORD, *qtau = qi + 2, *qaln = qtau + 1;
   const uint16_t qcnt = rd16(qi);
   if (qcnt == 0 || qcnt == ANN_NO_INFO || *qtau > tau) {
      *qtau = (uint8_t)tau;
      wr16(qi, (uint16_t)(nh < ANN_NO_INFO - 1 ? nh : ANN_NO_INFO - 1));
      ann_aln_positions(qalign, qaln, kmer, ANN_ALN, qrev);
   } else if (*qtau == tau) {
      const uint32_t c = (uint32_t)nh + (uint32_t)qcnt;
      wr16(qi, (uint16_t)(c < ANN_NO_INFO - 1 ? c : ANN_NO_INFO - 1));
      if (*qaln != 0xFF) {
         uint8_t tmp_aln[ANN_ALN];
         ann_aln_positions(qalign, tmp_aln, kmer, ANN_ALN, qrev);
         ann_aln_merge(qaln, tmp_aln, ANN_ALN);
      }
   }
}

/* neigh_push, src/index_ann.c:664-729.  `ann` may point outside the table where the reference's compression attaches
** unvisited suffix-array entries to a record (see orc_ann_build): those writes are dropped and counted. */
static void ann_neigh_push(const uint8_t *info, uint8_t *table, int64_t size, int64_t at, int reverse_kmer, int64_t *dropped)
{
   const uint16_t cnt = rd16(info);
   const uint8_t tau = info[2];
   if (cnt == 0 || cnt == ANN_NO_INFO) return;
   uint8_t v = 0;
   if (cnt <= 10) v = (uint8_t)cnt;
   else if (cnt <= 20) v = 0x0B;
   else if (cnt <= 50) v = 0x0C;
   else if (cnt <= 100) v = 0x0D;
   else if (cnt <= 500) v = 0x0E;
   else v = 0x0F;
   v |= (uint8_t)(((tau - 1) & 0x03) << 4);
   uint8_t aln[ANN_ALN];
   memcpy(aln, info + 3, ANN_ALN);
   const int have = aln[0] != 255;
   if (have) {
      if (reverse_kmer)
         for (int k = 0; k < ANN_ALN && aln[k]; k++) aln[k] = (uint8_t)(reverse_kmer + 1 - aln[k]);
      v |= 1u << 6;
   }
   if (at >= 0 && at < size) table[at] |= v; else (*dropped)++;
   if (have)
      for (int i = 0; i < ANN_ALN && aln[i]; i++) {
         const int64_t p = at + aln[i] - 1;
         if (p >= 0 && p < size) table[p] |= 1u << 7; else (*dropped)++;
      }
}

/* The reference's reuse of the last k-mer's search (`trail`), as it is: a tree of hit stacks per strand class
** (alloc_stack_tree, src/blocksearch.c:525-578) that keeps what the last search of each block left there. */
typedef struct pst { hitvec_t stack; struct pst *l, *r; } pst_t;

static pst_t *pst_new(int blocks)
{
   pst_t *n = calloc(1, sizeof *n);
   if (blocks > 1) { n->l = pst_new((blocks >> 1) + (blocks & 1)); n->r = pst_new(blocks >> 1); }
   return n;
}

static void pst_free(pst_t *n)
{
   if (!n) return;
   pst_free(n->l); pst_free(n->r);
   free(n->stack.v); free(n);
}

/* blocksearch_trail_rec, src/blocksearch.c:106-177, with its trail argument. */
static int block_rec_trail(sctx_t *c, int pos, int end, int blocks, int trail, pst_t *node)
{
   if (pos < 0 || pos > end || end < 0 || blocks < 0) return -1;     /* :125-128 */
   node->stack.n = 0;                                                /* :132-135 */
   if (blocks == 1) {
      orc_hit_t root;
      memset(&root, 0, sizeof root);
      root.sz = c->x->txt_len;
      search(c, &root, end, pos, -1, 0, 0, 0, &node->stack);
      return 0;
   }
   int blk_l = (blocks >> 1) + (blocks & 1), blk_r = blocks >> 1;
   int slen = end - pos + 1;
   int pos_r = pos + (slen >> 1) + (slen & 1), end_l = pos_r - 1;
   if (trail < pos_r)                                                /* :156-157 */
      if (block_rec_trail(c, pos, end_l, blk_l, trail, node->l) != 0) return -1;
   for (int64_t i = 0; i < node->l->stack.n; i++)                    /* :159-162 */
      search(c, &node->l->stack.v[i], pos_r, end, +1, blocks - 1, (int)node->l->stack.v[i].score, 0, &node->stack);
   if (block_rec_trail(c, pos_r, end, blk_r, trail, node->r) != 0) return -1;   /* :165 */
   for (int64_t i = 0; i < node->r->stack.n; i++)                    /* :167-170 */
      search(c, &node->r->stack.v[i], end_l, pos, -1, blocks - 1, (int)node->r->stack.v[i].score, blk_l, &node->stack);
   return 0;
}

/* blocksc_trail, src/blocksearch.c:7-103, with its trail argument; bound[i] = bwt_start + bwt_size of qarray[i]. */
static void blocksc_trail_emul(sctx_t *c, int slen, int tau, int trail, int last_fragment, const int64_t *bound, pst_t *tree)
{
   if (trail >= slen) return;                                        /* :41-43 */
   tree->stack.n = 0;                                                /* :48-52 */
   int n_cnt = 0;
   for (int i = 0; i < slen && n_cnt <= tau; i++) n_cnt += c->q[i] == ORC_UNKNOWN_BASE;
   tau -= n_cnt;
   if (tau < 0 || (tau == 0 && !last_fragment)) return;              /* :63-66 */
   int pos_r = slen / 2 + (last_fragment ? slen % 2 : 0);
   int tau_l = tau / 2 - (last_fragment ? 0 : (1 - tau % 2));
   if (trail < pos_r) {                                              /* :73-86 */
      if (block_rec_trail(c, 0, pos_r - 1, tau_l + 1, trail, tree->l) != 0) return;
      hitvec_t *ls = &tree->l->stack;
      int64_t i = 0;
      while (i < ls->n) {
         if (ls->v[i].fp < bound[pos_r]) i++;
         else ls->v[i] = ls->v[--ls->n];
      }
   }
   for (int64_t i = 0; i < tree->l->stack.n; i++)                    /* :89-92 */
      scsearch(c, &tree->l->stack.v[i], pos_r, slen - 1, tau, (int)tree->l->stack.v[i].score, 0, 1, &tree->stack);
   if (n_cnt)
      for (int64_t i = 0; i < tree->stack.n; i++) tree->stack.v[i].score += n_cnt;   /* :95-97 */
}

typedef struct { int64_t beg, end; } annjob_t;

/* jobs_by_prefix, src/index_ann.c:612-661 */
static void ann_jobs(const orc_index_t *x, int64_t fp, int64_t rp, int64_t sz, int tau, int n_cnt, int depth, int max_depth,
                     annjob_t *jobs, int *n_jobs)
{
   if (n_cnt > tau) return;
   if (depth == max_depth) { jobs[*n_jobs].beg = fp; jobs[*n_jobs].end = fp + sz; (*n_jobs)++; return; }
   int64_t qv[5][3];
   orc_extend_all(x, ORC_QUERY_SUFFIX, fp, rp, sz, qv);
   for (int i = 0; i < 5; i++)
      ann_jobs(x, qv[i][0], qv[i][1], qv[i][2], tau, n_cnt + (i == ORC_UNKNOWN_BASE), depth + 1, max_depth, jobs, n_jobs);
}

/* ann_build (src/index_ann.c:56-257) + ann_build_mt (:500-609) + neigh_next (:732-830), the jobs run one after the
** other in the order of the job list (what the reference does with threads = 1; `threads` only sets the depth of the
** prefixes the jobs are cut at, :116-121).  trail_reuse = 1: with the reference's reuse of the last k-mer's search, as
** it is (a tree of stacks per strand class and job; what a search that returned early or met an N left there is used
** again): the reference's table, byte for byte.  trail_reuse = 0: every k-mer is searched afresh (orc_blocksc), what
** blocksc_trail computes for trail = 0; the two differ only next to k-mers with N (tests/test_oracle_ann.py).
** table: [txt_len / 2] bytes, the content of ann->info (what ann_file_write stores behind its 32-byte header).
** Returns 0, -1 on bad arguments (:76-82); *dropped = writes of the compression that fell outside the table. */
int orc_ann_build(const orc_index_t *x, int kmer, int tau, int threads, int trail_reuse, uint8_t *table, int64_t *n_queries,
                  int64_t *n_ext_out, int64_t *dropped_out)
{
   if (kmer < 2 || tau < 1 || tau >= 4 || tau >= kmer || threads < 1 || !x || !table) return -1;
   const int64_t tlen = x->txt_len;
   uint8_t *info = calloc((size_t)tlen, ANN_WORD);
   if (!info) return -1;
   int depth = 1, num_jobs = 5;
   while (num_jobs < 5 * threads) { depth++; num_jobs *= 5; }
   annjob_t *jobs = malloc(sizeof *jobs * (size_t)num_jobs);
   int nj = 0;
   ann_jobs(x, 0, 0, tlen, tau, 0, 0, depth, jobs, &nj);
   uint8_t *seq = malloc((size_t)kmer), *query[2] = { malloc((size_t)kmer), malloc((size_t)kmer) };
   int64_t *bound = malloc(sizeof *bound * (size_t)(kmer + 1));
   int64_t n_q = 0, n_ext = 0, dropped = 0;
   for (int j = 0; j < nj; j++) {
      int64_t sa_ptr = jobs[j].beg;
      pst_t *tree[2] = { NULL, NULL };
      if (trail_reuse) {                                                         /* :523-544: per job */
         tree[0] = pst_new(tau + 1); tree[1] = pst_new(tau + 1);
         memset(query[0], 5, (size_t)kmer); memset(query[1], 5, (size_t)kmer);
      }
      while (sa_ptr < jobs[j].end) {
         /* neigh_next */
         const int64_t txt_pos = orc_sar_get(x, sa_ptr);
         int64_t next_sa = sa_ptr + 1;
         int valid = txt_pos + kmer <= tlen;                                     /* :765 */
         for (int i = 0; valid && i < kmer; i++) valid = x->text[txt_pos + i] < 5;   /* :773-777 */
         int64_t fp = 0, rp = 0, sz = tlen;
         int trail[2] = { 0, 0 };
         if (valid) {
            int n_cnt = 0;
            bound[0] = tlen;
            for (int i = 0; i < kmer; i++) {
               seq[i] = x->text[txt_pos + i];
               n_cnt += seq[i] == ORC_UNKNOWN_BASE;
               if (trail_reuse)                                                   /* :786-789 */
                  for (int t = 0; t < 2; t++) if (trail[t] == i && seq[i] == query[t][i]) trail[t]++;
               int64_t qv[5][3];
               orc_extend_all(x, ORC_QUERY_SUFFIX, fp, rp, sz, qv);               /* :791 */
               fp = qv[seq[i]][0]; rp = qv[seq[i]][1]; sz = qv[seq[i]][2];
               bound[i + 1] = fp + sz;
            }
            if (sz == 0) valid = 0;                                               /* :798 */
            else {
               next_sa = sa_ptr + sz;                                             /* :804 */
               if (n_cnt > tau) valid = 0;                                        /* :807 */
            }
         }
         if (!valid) {
            wr16(info + sa_ptr * ANN_WORD, ANN_NO_INFO);                          /* :559-561 */
         } else {
            n_q++;
            if (trail_reuse) {
               const int lf = fp >= rp;                                           /* :813-819 */
               memcpy(query[lf], seq, (size_t)kmer);
               sctx_t c = { x, query[lf], 0, 0 };
               blocksc_trail_emul(&c, kmer, tau, trail[lf], lf, bound, tree[lf]);
               n_ext += c.n_ext;
               ann_hits_push(info, kmer, tau, tree[lf]->stack.v, tree[lf]->stack.n, fp, rp);
            } else {
               orc_hit_t *hits = NULL;
               int64_t ne = 0;
               int64_t nh = orc_blocksc(x, seq, kmer, tau, &hits, &ne);           /* :564-567; its return value is not looked at */
               if (nh < 0) nh = 0;                                                /* (the stack was emptied before the error) */
               n_ext += ne;
               ann_hits_push(info, kmer, tau, hits, nh, fp, rp);                  /* :569 */
               free(hits);
            }
         }
         sa_ptr = next_sa;
      }
      pst_free(tree[0]); pst_free(tree[1]);
   }
   /* compression, :200-241 */
   memset(table, 0, (size_t)(tlen / 2));
   int64_t i = 0;
   while (1) {
      while (i < tlen && (rd16(info + i * ANN_WORD) == ANN_NO_INFO || rd16(info + i * ANN_WORD) == 0)) i++;
      if (i >= tlen) break;
      int64_t size = 1;
      while (i + size < tlen && rd16(info + (i + size) * ANN_WORD) == 0) size++;
      for (int64_t k = 0; k < size; k++) {
         const int64_t p = orc_sar_get(x, i + k);
         if (p >= tlen / 2) ann_neigh_push(info + i * ANN_WORD, table, tlen / 2, tlen - p - kmer, kmer, &dropped);
         else ann_neigh_push(info + i * ANN_WORD, table, tlen / 2, p, 0, &dropped);
      }
      i += size;
   }
   free(seq); free(query[0]); free(query[1]); free(bound); free(jobs); free(info);
   if (n_queries) *n_queries = n_q;
   if (n_ext_out) *n_ext_out = n_ext;
   if (dropped_out) *dropped_out = dropped;
   return 0;
}
