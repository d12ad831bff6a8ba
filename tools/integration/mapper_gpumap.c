/* mapper_gpumap.c -- the binding INTEGRATION.md section 1 describes, as a complete C99 translation unit: what a
** maintainer of ezorita/mapper puts in place of the loop of src/mapper.c:23-40 to map every chunk of reads on the GPU
** through libgpumap.so.  It includes only the reference's own headers and include/gpumap.h and calls only functions
** they declare.  Not part of the library: tests/test_host.py compiles it (gcc -std=c99 -fsyntax-only) against the
** reference's headers where they are present, so that the header and the sketch cannot drift from the interface
** they bind.
**
** Two forms: mapper_gpu() maps one chunk at a time (upload -> run -> fetch); mapper_gpu_pipelined() keeps `DEPTH`
** chunks in flight on a gpumap_pipe, so that the copy of one chunk, the search of the next and the return of the
** results of the one before overlap, and prints from the compact result (sorted 64-bit keys). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "mapper.h"          /* errhandler.h, index.h, io.h: the reference's own includes for mapper.c */
#include "gpumap.h"

/* The letters of a chunk as one buffer + offsets (seqread_seq / seqread_len, src/seqread.h:27-28). */
static int
flatten_chunk
(
 gstack_t  * rstack,
 char     ** text,
 int64_t  ** offsets
)
{
   int64_t n = gstack_num_elm(rstack);
   int64_t * off = malloc((size_t)(n + 1) * sizeof *off);
   error_test_mem(off);
   off[0] = 0;
   for (int64_t i = 0; i < n; i++)
      off[i + 1] = off[i] + seqread_len(seqread_get((size_t)i, rstack));
   /* page-locked, so that the copy to the GPU runs at the rate of the link */
   char * txt = gpumap_host_alloc((size_t)off[n] + 1);
   if (txt == NULL) { free(off); goto failure_return; }
   for (int64_t i = 0; i < n; i++)
      memcpy(txt + off[i], seqread_seq(seqread_get((size_t)i, rstack)), (size_t)(off[i + 1] - off[i]));
   *text = txt;
   *offsets = off;
   return 0;

 failure_return:
   return -1;
}

/* One line per located hit: tag, locus as the reference prints it (txt_pos_to_str, src/index_txt.c:480-511), score. */
static int
print_hit
(
 gstack_t * rstack,
 index_t  * index,
 int64_t    read,
 int64_t    pos,
 int        score
)
{
   char * locus = txt_pos_to_str(pos, index->txt);
   error_test(locus == NULL);
   fprintf(stdout, "%s\t%s\t%d\n", seqread_tag(seqread_get((size_t)read, rstack)), locus, score);
   free(locus);
   return 0;

 failure_return:
   return -1;
}

int
mapper_gpu
(
 char    * qfile,
 index_t * index,
 int       tau
)
{
   iostream_t   * qinput = NULL;
   gstack_t     * rstack = NULL;
   gpumap_ctx   * ctx    = NULL;
   gpumap_batch * batch  = NULL;

   error_test_msg(qfile == NULL, "argument 'qfile' is NULL.");
   error_test_msg(index == NULL, "argument 'index' is NULL.");

   /* bwt_t / sar_t / txt_t are opaque outside their .c files: the context maps the index files itself. */
   ctx = gpumap_create_from_files(index->fname_base, 0);
   error_test(ctx == NULL);
   batch = gpumap_batch_new(ctx);
   error_test(batch == NULL);

   qinput = io_stream_open(qfile);
   error_test(qinput == NULL);

   while ((rstack = io_stream_read_seq(qinput)) != NULL) {
      if (gstack_num_elm(rstack) == 0) break;
      char * text = NULL;
      int64_t * off = NULL;
      gpumap_hits h;
      error_test(flatten_chunk(rstack, &text, &off) == -1);
      error_test(gpumap_batch_refill_text(ctx, batch, text, off, gstack_num_elm(rstack)) == -1);
      error_test(gpumap_batch_run(ctx, batch, tau, 0) == -1);
      error_test(gpumap_batch_fetch(ctx, batch, &h) == -1);
      /* reads in input order, positions ascending within a read */
      for (int64_t t = 0; t < h.n_locs; t++)
         error_test(print_hit(rstack, index, h.loc_read[t], h.loc_pos[t], h.loc_score[t]) == -1);
      gpumap_hits_free(&h);
      gpumap_host_free(text);
      free(off);
      gstack_free(rstack);
   }

   io_stream_free(qinput);
   gpumap_batch_free(batch);
   gpumap_destroy(ctx);
   return 0;

 failure_return:
   if (batch != NULL) gpumap_batch_free(batch);
   if (ctx != NULL) gpumap_destroy(ctx);
   return -1;
}

/* What travels with a chunk through the pipe. */
typedef struct {
   gstack_t * rstack;
   char     * text;
   int64_t  * off;
} chunk_t;

static int
print_chunk
(
 gpumap_pipe * pipe,
 index_t     * index
)
{
   gpumap_hits h;
   void * user = NULL;
   int rc = gpumap_pipe_next(pipe, &h, &user);
   error_test(rc == -1);
   if (rc == 0) return 0;
   chunk_t * c = user;
   for (int64_t t = 0; t < h.n_locs; t++) {
      uint64_t k = h.loc_key[t];
      error_test(print_hit(c->rstack, index, GPUMAP_KEY_READ(k, h.key_pos_bits), GPUMAP_KEY_POS(k, h.key_pos_bits),
                           GPUMAP_KEY_SCORE(k)) == -1);
   }
   gpumap_pipe_release(pipe, &h);
   gpumap_host_free(c->text);
   free(c->off);
   gstack_free(c->rstack);
   free(c);
   return 1;

 failure_return:
   return -1;
}

#define DEPTH 3

int
mapper_gpu_pipelined
(
 char    * qfile,
 index_t * index,
 int       tau
)
{
   iostream_t  * qinput = NULL;
   gstack_t    * rstack = NULL;
   gpumap_ctx  * ctx    = NULL;
   gpumap_pipe * pipe   = NULL;

   error_test_msg(qfile == NULL, "argument 'qfile' is NULL.");
   error_test_msg(index == NULL, "argument 'index' is NULL.");

   ctx = gpumap_create_from_files(index->fname_base, 0);
   error_test(ctx == NULL);
   pipe = gpumap_pipe_create(&ctx, 1, DEPTH, tau, GPUMAP_SRC_TEXT | GPUMAP_PIPE_COMPACT);
   error_test(pipe == NULL);

   qinput = io_stream_open(qfile);
   error_test(qinput == NULL);

   while ((rstack = io_stream_read_seq(qinput)) != NULL) {
      if (gstack_num_elm(rstack) == 0) break;
      chunk_t * c = malloc(sizeof *c);
      error_test_mem(c);
      c->rstack = rstack;
      error_test(flatten_chunk(rstack, &c->text, &c->off) == -1);
      /* single-threaded caller: never waits inside the pipe; takes a result whenever every slot is taken */
      for (;;) {
         int rc = gpumap_pipe_try_submit(pipe, c->text, c->off, gstack_num_elm(rstack), c);
         error_test(rc == -1);
         if (rc == 1) break;
         error_test(print_chunk(pipe, index) == -1);
      }
   }
   gpumap_pipe_close(pipe);
   for (;;) {
      int rc = print_chunk(pipe, index);
      error_test(rc == -1);
      if (rc == 0) break;
   }

   io_stream_free(qinput);
   gpumap_pipe_destroy(pipe);
   gpumap_destroy(ctx);
   return 0;

 failure_return:
   if (pipe != NULL) gpumap_pipe_destroy(pipe);
   if (ctx != NULL) gpumap_destroy(ctx);
   return -1;
}
