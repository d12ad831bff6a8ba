/*
** ref_driver.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
**
** A driver, written for this repository, around the *unmodified* reference sources of
** ezorita/mapper, which are compiled where they lie under /root/reference/src (see
** oracle/Makefile, target _ref/ref_driver).  No reference source is copied here: this
** file only calls the reference's public functions through its own headers.
**
** It performs the composition that test/test_blocksearch.c:179-188 of the reference
** performs (and that SURVEY.md section 3.2 names as the oracle for "read -> hits"):
**
**   sym_str_index(read)                                   src/index_sym.c:293
**   blocksearch_trail_rec(q, 0, m-1, tau+1, 0, bwt, tree)  src/blocksearch.c:106
**   for every hit: sar_get_range(fp, sz, buf, sar)         src/index_sar.c:121
**
** Commands
**   ref_driver build <ref.fa> <base>                 index_build() -> base.{sym,txt,sar,bwt}
**   ref_driver map   <base> <reads.txt> <tau> <out.bin> [threads]
**   ref_driver sc    <base> <kmers.txt> <tau> <out.bin>   blocksc_trail(q, qarray, slen, tau, 0, tree) per line
**                                                    (src/blocksearch.c:7-103; qarray as test/test_blocksearch.c:49-53)
**   ref_driver sctrail <base> <kmers.txt> <tau> <out.bin> the same, driven as ann_build_mt drives it
**                                                    (src/index_ann.c:556-567, :786-825): a tree and a previous query
**                                                    per strand class, trail = letters shared with that query
**   ref_driver kat   <out.txt>                       dumps answers of the reference's own
**                                                    primitives on literal texts (used to
**                                                    cross-check tests/golden/kat_*.json)
**
** reads.txt holds one read per line (raw letters).  out.bin is the "hit dump" format shared
** with oracle/fmoracle.c and the HIP path (tests/dumpfmt.py documents it):
**   u64 magic 0x48495444554d5031 ("1PMUDTIH"), i64 n_reads, i64 n_hits, i64 n_locs,
**   i64 status[n_reads]            (0 = searched, -1 = the reference returned an error)
**   i64 hit[n_hits][11]            read, fp, rp, sz, dp, score, align0..3, first_loc
**   i64 loc[n_locs]                text positions, hit after hit, in SA order
** Hits are in the reference's own order (DFS order, left extensions before right ones).
*/
#define _POSIX_C_SOURCE 200809L
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <time.h>
#include <pthread.h>
#include "index.h"
#include "blocksearch.h"

#define DUMP_MAGIC 0x48495444554d5031ULL

typedef struct {
   int64_t  n_hits, n_locs, cap_hits, cap_locs;
   int64_t *hits;   /* 11 per hit */
   int64_t *locs;
} out_t;

typedef struct {
   index_t  *index;
   char    **reads;
   int64_t   beg, end;
   int       tau;
   int       mode;     /* 0 map (blocksearch_trail_rec), 1 sc (blocksc_trail, trail 0), 2 sctrail (as ann_build_mt) */
   int64_t  *status;
   out_t     out;
   int       rc;
} job_t;

static int out_reserve(out_t *o, int64_t nh, int64_t nl)
{
   if (o->n_hits + nh > o->cap_hits) {
      int64_t c = o->cap_hits ? o->cap_hits : 1024;
      while (c < o->n_hits + nh) c *= 2;
      int64_t *p = realloc(o->hits, c * 11 * sizeof(int64_t));
      if (!p) return -1;
      o->hits = p; o->cap_hits = c;
   }
   if (o->n_locs + nl > o->cap_locs) {
      int64_t c = o->cap_locs ? o->cap_locs : 1024;
      while (c < o->n_locs + nl) c *= 2;
      int64_t *p = realloc(o->locs, c * sizeof(int64_t));
      if (!p) return -1;
      o->locs = p; o->cap_locs = c;
   }
   return 0;
}

/* The search of one line: 0 / -1 as the reference returns; *used = the tree that holds the hits. */
static int search_one(job_t *j, uint8_t *q, int m, pstree_t **trees, uint8_t **prev, int *prev_len, pstree_t **used)
{
   *used = trees[0];
   if (j->mode == 0)
      return blocksearch_trail_rec(q, 0, m - 1, j->tau + 1, 0, j->index->bwt, trees[0]);
   /* the path array of the callers of blocksc_trail: one bwt_query SUFFIX per letter from the root interval */
   bwtquery_t **qa = calloc((size_t)m + 1, sizeof *qa);
   int rc = 0;
   qa[0] = bwt_new_query(j->index->bwt);
   for (int i = 0; i < m && rc == 0; i++) {
      qa[i + 1] = bwt_new_query(j->index->bwt);
      if (bwt_query(q[i], BWT_QUERY_SUFFIX, qa[i], qa[i + 1]) == -1) rc = -1;
   }
   if (rc == 0) {
      int cls = 0, trail = 0;
      if (j->mode == 2) {
         /* as neigh_next (src/index_ann.c:786-825): the class is the strand test, the trail the letters shared with
         ** the last query of that class */
         cls = bwt_start(qa[m]) >= bwt_rcstart(qa[m]);
         while (trail < m && trail < prev_len[cls] && prev[cls][trail] == q[trail]) trail++;
         if (trail >= m) trail = 0;         /* the same k-mer twice: ann_build_mt never asks again, search afresh */
         memcpy(prev[cls], q, (size_t)m);
         prev_len[cls] = m;
      }
      if (j->mode == 1) {
         /* every line on a fresh tree: with trail = 0 blocksc_trail still skips the left search when the left block
         ** is empty (pos_r = 0, src/blocksearch.c:73) and would extend what the line before left in the tree */
         free_stack_tree(trees[0]);
         trees[0] = alloc_stack_tree(j->tau);
         if (trees[0] == NULL) rc = -1;
      }
      *used = trees[cls];
      if (rc == 0) rc = blocksc_trail(q, qa, m, j->tau, trail, trees[cls]);
      if (rc == -1 && j->mode == 2) prev_len[cls] = 0;
   }
   for (int i = 0; i <= m; i++) free(qa[i]);
   free(qa);
   return rc;
}

static void *map_worker(void *arg)
{
   job_t *j = (job_t *)arg;
   pstree_t *trees[2] = { alloc_stack_tree(j->tau), alloc_stack_tree(j->tau) };
   uint8_t *prev[2] = { calloc(257, 1), calloc(257, 1) };
   int prev_len[2] = { 0, 0 };
   pstree_t *tree = trees[0];
   if (trees[0] == NULL || trees[1] == NULL) { j->rc = -1; return NULL; }
   for (int64_t r = j->beg; r < j->end; r++) {
      char *read = j->reads[r];
      int m = (int)strlen(read);
      j->status[r] = 0;
      /* Longer than ALIGN_WORDS*ALIGN_WORD_SIZE (src/blocksearch.h:12-13): aln_bit_set would write
      ** past spath_t.align (undefined behaviour in the reference), so such reads are not searched. */
      if (m < 1 || m > 256) { j->status[r] = -1; continue; }
      uint8_t *q = sym_str_index(read, j->index->sym);
      if (q == NULL) { j->rc = -1; break; }
      if (search_one(j, q, m, trees, prev, prev_len, &tree) == -1) {
         /* e.g. more blocks than letters: src/blocksearch.c:126 */
         j->status[r] = -1;
         free(q);
         /* The tree may hold partial results of inner calls: drop and rebuild it. */
         int which = tree == trees[1];
         free_stack_tree(tree);
         trees[which] = tree = alloc_stack_tree(j->tau);
         if (tree == NULL) { j->rc = -1; break; }
         continue;
      }
      pathstack_t *st = tree->stack;
      int64_t nl = 0;
      for (size_t i = 0; i < st->pos; i++) nl += bwt_size(st->path[i].bwtq);
      if (out_reserve(&j->out, (int64_t)st->pos, nl) == -1) { j->rc = -1; free(q); break; }
      for (size_t i = 0; i < st->pos; i++) {
         spath_t *p = &st->path[i];
         int64_t *h = j->out.hits + 11 * j->out.n_hits;
         h[0] = r;
         h[1] = bwt_start(p->bwtq);
         h[2] = bwt_rcstart(p->bwtq);
         h[3] = bwt_size(p->bwtq);
         h[4] = bwt_depth(p->bwtq);
         h[5] = p->score;
         for (int w = 0; w < 4; w++) h[6 + w] = (int64_t)p->align[w];
         h[10] = j->out.n_locs;   /* thread-local; rebased when merged */
         if (sar_get_range(h[1], h[3], j->out.locs + j->out.n_locs, j->index->sar) == -1) {
            j->rc = -1;
         }
         j->out.n_locs += h[3];
         j->out.n_hits++;
      }
      free(q);
   }
   free_stack_tree(trees[0]); free_stack_tree(trees[1]);
   free(prev[0]); free(prev[1]);
   return NULL;
}

static char **read_lines(const char *fn, int64_t *n_out)
{
   FILE *f = fopen(fn, "r");
   if (!f) { perror(fn); return NULL; }
   int64_t cap = 1024, n = 0;
   char **v = malloc(cap * sizeof(char *));
   char *line = NULL; size_t sz = 0; ssize_t len;
   while ((len = getline(&line, &sz, f)) != -1) {
      while (len > 0 && (line[len - 1] == '\n' || line[len - 1] == '\r')) line[--len] = 0;
      if (n == cap) { cap *= 2; v = realloc(v, cap * sizeof(char *)); }
      v[n++] = strdup(line);
   }
   free(line);
   fclose(f);
   *n_out = n;
   return v;
}

static double now_s(void)
{
   struct timespec ts;
   clock_gettime(CLOCK_MONOTONIC, &ts);
   return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

static int cmd_map(int argc, char **argv, int mode)
{
   if (argc < 6) return 2;
   const char *base = argv[2], *rfile = argv[3], *ofile = argv[5];
   int tau = atoi(argv[4]);
   int threads = argc > 6 ? atoi(argv[6]) : 1;
   if (threads < 1 || mode == 2) threads = 1;          /* the trail depends on the order of the lines */

   index_t *index = index_read((char *)base);
   if (index == NULL) { fprintf(stderr, "ref_driver: cannot read index %s\n", base); return 1; }
   int64_t n_reads = 0;
   char **reads = read_lines(rfile, &n_reads);
   if (reads == NULL) return 1;
   int64_t *status = calloc(n_reads > 0 ? n_reads : 1, sizeof(int64_t));

   if (threads > n_reads) threads = n_reads > 0 ? (int)n_reads : 1;
   job_t *jobs = calloc(threads, sizeof(job_t));
   pthread_t *tid = calloc(threads, sizeof(pthread_t));
   double t0 = now_s();
   for (int t = 0; t < threads; t++) {
      jobs[t].index = index; jobs[t].reads = reads; jobs[t].tau = tau; jobs[t].status = status; jobs[t].mode = mode;
      jobs[t].beg = n_reads * t / threads;
      jobs[t].end = n_reads * (t + 1) / threads;
      pthread_create(&tid[t], NULL, map_worker, &jobs[t]);
   }
   int rc = 0;
   for (int t = 0; t < threads; t++) { pthread_join(tid[t], NULL); if (jobs[t].rc) rc = 1; }
   double t1 = now_s();

   int64_t n_hits = 0, n_locs = 0;
   for (int t = 0; t < threads; t++) { n_hits += jobs[t].out.n_hits; n_locs += jobs[t].out.n_locs; }
   FILE *fo = fopen(ofile, "wb");
   if (!fo) { perror(ofile); return 1; }
   uint64_t magic = DUMP_MAGIC;
   fwrite(&magic, 8, 1, fo);
   fwrite(&n_reads, 8, 1, fo); fwrite(&n_hits, 8, 1, fo); fwrite(&n_locs, 8, 1, fo);
   fwrite(status, 8, n_reads, fo);
   int64_t loc_base = 0;
   for (int t = 0; t < threads; t++) {
      for (int64_t i = 0; i < jobs[t].out.n_hits; i++) jobs[t].out.hits[11 * i + 10] += loc_base;
      fwrite(jobs[t].out.hits, 8, 11 * jobs[t].out.n_hits, fo);
      loc_base += jobs[t].out.n_locs;
   }
   for (int t = 0; t < threads; t++) fwrite(jobs[t].out.locs, 8, jobs[t].out.n_locs, fo);
   fclose(fo);
   /* One machine-readable line for the bench harness. */
   printf("{\"kind\": \"reference\", \"reads\": %ld, \"hits\": %ld, \"locs\": %ld, \"threads\": %d, "
          "\"seconds\": %.6f, \"reads_per_s\": %.1f}\n",
          (long)n_reads, (long)n_hits, (long)n_locs, threads, t1 - t0,
          (t1 - t0) > 0 ? n_reads / (t1 - t0) : 0.0);
   return rc;
}

static int cmd_build(int argc, char **argv)
{
   if (argc < 4) return 2;
   double t0 = now_s();
   index_t *index = index_build(argv[2], argv[3]);
   if (index == NULL) { fprintf(stderr, "ref_driver: index_build failed\n"); return 1; }
   printf("{\"kind\": \"reference\", \"txt_len\": %ld, \"seconds\": %.3f}\n",
          (long)txt_length(index->txt), now_s() - t0);
   index_free(index);
   return 0;
}

/* Answers of the reference primitives on literal texts: every "A<TAB>B" token is a text made of
** letters, with '$' for txt_append_wildcard().  Used to double-check the KAT fixtures. */
static int cmd_kat(int argc, char **argv)
{
   if (argc < 4) return 2;
   const char *spec = argv[2];
   FILE *fo = fopen(argv[3], "w");
   if (!fo) return 1;
   sym_t *sym = sym_new_dna();
   txt_t *txt = txt_new(sym);
   /* spec: letters and '$' */
   char buf[2] = {0, 0};
   for (const char *c = spec; *c; c++) {
      if (*c == '$') txt_append_wildcard(txt);
      else { buf[0] = *c; txt_append(buf, txt); }
   }
   sar_t *sar = sar_build(txt);
   bwt_t *bwt = bwt_build(txt, sar);
   int64_t n = txt_length(txt);
   fprintf(fo, "n %ld\nsa", (long)n);
   for (int64_t i = 0; i < n; i++) fprintf(fo, " %ld", (long)sar_get(i, sar));
   fprintf(fo, "\n");
   /* All one- and two-step bidirectional extensions from the root. */
   bwtquery_t *root = bwt_new_query(bwt);
   bwtquery_t **qv = bwt_new_vec(bwt), **qw = bwt_new_vec(bwt);
   for (int end = 0; end < 2; end++) {
      bwt_query_all(end, root, qv);
      for (int a = 0; a < 5; a++) {
         fprintf(fo, "ext %d %d : %ld %ld %ld %ld\n", end, a, (long)bwt_start(qv[a]),
                 (long)bwt_rcstart(qv[a]), (long)bwt_size(qv[a]), (long)bwt_depth(qv[a]));
         if (bwt_size(qv[a]) < 1) continue;
         for (int end2 = 0; end2 < 2; end2++) {
            bwt_query_all(end2, qv[a], qw);
            for (int b = 0; b < 5; b++)
               fprintf(fo, "ext %d %d %d %d : %ld %ld %ld %ld\n", end, a, end2, b,
                       (long)bwt_start(qw[b]), (long)bwt_rcstart(qw[b]), (long)bwt_size(qw[b]),
                       (long)bwt_depth(qw[b]));
         }
      }
   }
   fclose(fo);
   return 0;
}

/* ann <base> <kmer> <tau> <threads> <out.ann>: the reference's own ann_build (src/index_ann.c:56-257) on the index
** files of <base>, written with its own ann_file_write (:384-434).  The progress lines go to stderr. */
static int cmd_ann(int argc, char **argv)
{
   if (argc < 7) return 2;
   index_t *index = index_read(argv[2]);
   if (index == NULL) { fprintf(stderr, "ref_driver: index_read failed\n"); return 1; }
   double t0 = now_s();
   ann_t *ann = ann_build(atoi(argv[3]), atoi(argv[4]), index->bwt, index->sar, atoi(argv[5]));
   if (ann == NULL) { fprintf(stderr, "ref_driver: ann_build failed\n"); return 1; }
   double t1 = now_s();
   if (ann_file_write(argv[6], ann) == -1) { fprintf(stderr, "ref_driver: ann_file_write failed\n"); return 1; }
   printf("{\"kind\": \"reference\", \"kmer\": %d, \"tau\": %d, \"threads\": %d, \"txt_len\": %ld, \"seconds\": %.3f}\n",
          atoi(argv[3]), atoi(argv[4]), atoi(argv[5]), (long)txt_length(index->txt), t1 - t0);
   ann_free(ann);
   return 0;
}

int main(int argc, char **argv)
{
   if (argc >= 2 && strcmp(argv[1], "ann") == 0) return cmd_ann(argc, argv);
   if (argc >= 2 && strcmp(argv[1], "build") == 0) return cmd_build(argc, argv);
   if (argc >= 2 && strcmp(argv[1], "map") == 0) return cmd_map(argc, argv, 0);
   if (argc >= 2 && strcmp(argv[1], "sc") == 0) return cmd_map(argc, argv, 1);
   if (argc >= 2 && strcmp(argv[1], "sctrail") == 0) return cmd_map(argc, argv, 2);
   if (argc >= 2 && strcmp(argv[1], "kat") == 0) return cmd_kat(argc, argv);
   fprintf(stderr, "usage: ref_driver build <ref.fa> <base> | map <base> <reads.txt> <tau> <out.bin> [threads]"
                   " | kat <spec> <out.txt> | ann <base> <kmer> <tau> <threads> <out.ann>\n");
   return 2;
}
