// gm_host.hpp -- host-side declarations shared by the translation units of libgpumap.so.
#pragma once
#include <cerrno>
#include <cstdint>
#include <cstdio>
#include <functional>
#include <string>
#include <vector>
#include "gpumap.h"

// Error convention of the reference (src/errhandler.h:11-70, src/errhandler.c:3-16).
#define GM_WARN(msg) std::fprintf(stderr, "%s\t(%s:%d) error: %s\n", __func__, __FILE__, __LINE__, (msg))
#define GM_FAIL(msg) do { GM_WARN(msg); errno = 1; goto failure_return; } while (0)
#define GM_FAIL_IF(cond, msg) do { if (cond) GM_FAIL(msg); } while (0)

constexpr uint64_t GM_MAGIC_SYM = 0x0fcb0fcb00010001ULL;  // src/index_sym.h:18
constexpr uint64_t GM_MAGIC_TXT = 0x0fcb0fcb00020001ULL;  // src/index_txt.h:22
constexpr uint64_t GM_MAGIC_BWT = 0x0fcb0fcb00030001ULL;  // src/index_bwt.h:19
constexpr uint64_t GM_MAGIC_SAR = 0x0fcb0fcb00040001ULL;  // src/index_sar.h:19

struct gm_mapping { void *ptr = nullptr; size_t len = 0; };

// Host index handle: either views into four mmapped files or vectors owned by a fresh build.
struct gpumap_index {
   std::string base;
   gm_mapping  m_sym, m_txt, m_sar, m_bwt;
   // owned storage (index_build)
   std::vector<uint8_t>  own_text;
   std::vector<int64_t>  own_sa, own_seq_len, own_seq_beg;
   std::vector<uint64_t> own_occ, own_c;
   std::vector<std::string> own_names;
   std::vector<const char *> name_ptrs;
   uint8_t sym_table[256];
   uint8_t com_table[8];
   char    canon[8];
   gpumap_index_view view;
};

// Suffix array of `data` (symbols already shifted so that the separator is the smallest,
// src/index_sar.c:46-51) in the order divsufsort gives (plain suffix order; a suffix that is a
// prefix of another sorts first).  gm_sufsort.cpp.
int gm_suffix_sort(const uint8_t *data, int64_t n, int64_t *sa, int threads);

// Growable byte buffer that does not value-initialise what it grows by (a 256-MB chunk is written by
// read() right away) and keeps its memory when cleared; optionally page-locked (gpumap_host_alloc).
struct gm_bytes {
   char  *p = nullptr;
   size_t n = 0, cap = 0;
   bool   pinned = false;
   gm_bytes() = default;
   gm_bytes(const gm_bytes &) = delete;
   gm_bytes &operator=(const gm_bytes &) = delete;
   ~gm_bytes() { release(); }
   void release();
   bool reserve(size_t c);                 // keeps the contents; false when out of memory
   bool resize(size_t k) { if (!reserve(k)) return false; n = k; return true; }
   void clear() { n = 0; }
   char *data() { return p; }
   const char *data() const { return p; }
   size_t size() const { return n; }
   bool empty() const { return n == 0; }
};

// FASTQ / FASTA / raw reader (gm_reads.cpp).
struct gm_read_chunk {
   gm_bytes              raw;      // the block of the file the chunk was cut from (tags point into it)
   gm_bytes              text;     // concatenated read letters
   std::vector<int64_t>  offsets;  // n+1
   std::vector<uint32_t> tag_off, tag_len;   // FASTQ / FASTA: the tag of read i is raw[tag_off[i], +tag_len[i])
   std::vector<uint64_t> eol;      // scratch: ends of the lines of the block
   int64_t desc[5] = {0, 0, 0, 0, 0};        // gm_reads_next_block: what GPUMAP_SRC_FILE asks for beside the block (raw)
   int64_t first_line = 1;         // raw format: the tag of read i is its line number, first_line + i
   bool    numbered = false;
   double  t_read = 0, t_parse = 0;   // seconds spent on read() and on parsing for this chunk
   // The tag of read i (not NUL-terminated); `num` is scratch for numbered tags.
   const char *tag(size_t i, char num[32], size_t *len) const
   {
      if (!numbered) { *len = tag_len[i]; return raw.data() + tag_off[i]; }
      *len = (size_t)snprintf(num, 32, "%ld", (long)(first_line + (int64_t)i));
      return num;
   }
};
struct gm_read_stream;
gm_read_stream *gm_reads_open(const char *path);
// Returns number of reads (0 at EOF, -1 on error); at most `max_bytes` of input per chunk.
int64_t gm_reads_next(gm_read_stream *s, size_t max_bytes, gm_read_chunk *chunk);
// The same cut, the block left unparsed in chunk->raw (for the device: GPUMAP_SRC_FILE), chunk->desc filled.
int64_t gm_reads_next_block(gm_read_stream *s, size_t max_bytes, gm_read_chunk *chunk);
void gm_reads_close(gm_read_stream *s);
// Page-locked host blocks that go round (gm_reads.cpp): take one of at least `want` bytes (*cap = its size), give it back.
void *gm_pinned_take(size_t want, size_t *cap);
void gm_pinned_give(void *p, size_t cap);
// fn(t) for t = 0..threads-1 on as many threads (the caller's among them); returns when all are done.
void gm_parallel_for(int threads, const std::function<void(int)> &fn);
int gm_host_threads();     // worker threads of the host stages: GPUMAP_HOST_THREADS, else min(16, cores)

// Internal forms of the batch calls (gm_device.hip), used by the pipeline (gm_pipe.cpp): the public entry points
// derive the flags from the context's options, the pipeline brings its own.
constexpr int GM_RESULT_PINNED = 1, GM_RESULT_COMPACT = 2, GM_RESULT_ONE_ARENA = 4, GM_RUN_SC = 8,
              GM_RUN_NO_LOCATE = 16;      // gm_batch_run: interval hits only (gpumap_ann_build), no located hits
constexpr int GM_RESULT_PACKED = 64;      // with GM_RESULT_COMPACT: 5 bytes per hit, 4-byte offsets, a status byte per read (k_pack_results)
constexpr int GM_RESULT_LINES = 32;       // gm_batch_run / gm_batch_fetch: the located hits as the lines mapper() prints (a batch filled with GM_SRC_FILE)
constexpr int GM_SRC_TEXT = GPUMAP_SRC_TEXT, GM_SRC_DEVICE = GPUMAP_SRC_DEVICE, GM_SRC_FILE = GPUMAP_SRC_FILE;
int gm_batch_fill(gpumap_ctx *ctx, gpumap_batch *b, const void *src, const int64_t *offsets, int64_t n_reads, int src_flags, bool encode_now);
int gm_batch_run(gpumap_ctx *ctx, gpumap_batch *b, int tau, int want_intervals, int result_flags);
int gm_batch_fetch(gpumap_ctx *ctx, gpumap_batch *b, gpumap_hits *out, int result_flags);
// The streams a batch puts its work on (HIP streams as void *: this header is host-only): the run's kernels,
// the copy of the results to the host, the uploads.  A pipeline shares three streams per context among its batches.
// Batch objects that go round between the pipelines of one context (gm_device.hip).
gpumap_batch *gm_ctx_take_batch(gpumap_ctx *ctx);
void gm_ctx_give_batch(gpumap_ctx *ctx, gpumap_batch *b);
void gm_batch_set_streams(gpumap_batch *b, void *run, void *copy, void *up, void *post);
int  gm_ctx_pipe_streams(gpumap_ctx *ctx, void **run, void **copy, void **up, void **post);
// Called by gm_batch_run when the search of the run is over and only locate / order / formatting remain (on the post stream).
void gm_batch_set_hook(gpumap_batch *b, std::function<void()> on_searched);
void gm_batch_set_gate(gpumap_batch *b, std::function<void()> before_search, std::function<void()> on_ordering);
