// gm_mapper.cpp -- mapper() (src/mapper.c:3-47): chunks of reads -> GPU batches -> one line per located
// hit.  The reference's loop (read a chunk, work on it, print, next chunk; src/mapper.c:23-40) as stages
// that overlap: a reader thread cuts and parses the chunks (gm_reads.cpp) and hands them to a gpumap_pipe
// (gm_pipe.cpp: copy to the device, search, copy of the results back, each chunk on a slot with a stream of
// its own, over one or several GPUs), the calling thread takes the results in input order, formats and writes
// them (txt_pos_to_str's "name:pos:strand", src/index_txt.c:480-511, from the compact keys: strand and
// coordinates by the rule of :492-505 with seq_bisect :770-787).  Host code only: no HIP in this file.
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <chrono>
#include <climits>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>
#include "gm_host.hpp"

namespace {

struct Job {
   gm_read_chunk chunk;
   int64_t n = 0;
   gpumap_hits h;
};

// Bounded hand-over between two stages.  close() wakes everybody; pop() returns nullptr when closed and empty.
struct Chan {
   std::mutex m;
   std::condition_variable cv;
   std::deque<Job *> q;
   size_t cap;
   bool closed = false;
   explicit Chan(size_t c) : cap(c) {}
   bool push(Job *j)
   {
      std::unique_lock<std::mutex> l(m);
      cv.wait(l, [&] { return q.size() < cap || closed; });
      if (closed) return false;
      q.push_back(j);
      cv.notify_all();
      return true;
   }
   Job *pop()
   {
      std::unique_lock<std::mutex> l(m);
      cv.wait(l, [&] { return !q.empty() || closed; });
      if (q.empty()) return nullptr;
      Job *j = q.front();
      q.pop_front();
      cv.notify_all();
      return j;
   }
   void wait_empty()
   {
      std::unique_lock<std::mutex> l(m);
      cv.wait(l, [&] { return q.empty() || closed; });
   }
   void close()
   {
      std::lock_guard<std::mutex> l(m);
      closed = true;
      cv.notify_all();
   }
   void drain()
   {
      std::lock_guard<std::mutex> l(m);
      for (Job *j : q) delete j;
      q.clear();
   }
};

inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

inline char *put_num(char *p, int64_t v)
{
   char tmp[24];
   int k = 0;
   uint64_t u = v < 0 ? (uint64_t)(-v) : (uint64_t)v;
   if (v < 0) *p++ = '-';
   do { tmp[k++] = (char)('0' + u % 10); u /= 10; } while (u);
   while (k) *p++ = tmp[--k];
   return p;
}

// "tag \t name:pos:strand \t score \n" for the located hits [lo, hi) of a chunk (compact results: sorted keys).
void format_range(const Job &j, const gpumap_index_view &v, const std::vector<size_t> &name_len, int64_t lo, int64_t hi,
                  std::vector<char> &out)
{
   const gpumap_hits &h = j.h;
   const int bits = h.key_pos_bits;
   const int64_t n = v.txt_len;
   size_t max_name = 0;
   for (size_t l : name_len) max_name = std::max(max_name, l);
   size_t need = 0;
   char num[32];
   for (int64_t t = lo; t < hi; t++) {
      size_t tl;
      (void)j.chunk.tag((size_t)GPUMAP_KEY_READ(h.loc_key[t], bits), num, &tl);
      need += tl + max_name + 32;
   }
   out.resize(need);
   char *p = out.data();
   for (int64_t t = lo; t < hi; t++) {
      const uint64_t k = h.loc_key[t];
      const int64_t pos = GPUMAP_KEY_POS(k, bits);
      const bool minus = pos >= n / 2;                                  // src/index_txt.c:492-496
      const int64_t f = minus ? n - 2 - pos : pos;
      int64_t a = 0, e = v.seq_cnt;                                     // seq_bisect, src/index_txt.c:770-787
      while (e - a >= 2) { const int64_t mid = (a + e) / 2; if (f < v.seq_beg[mid]) e = mid; else a = mid; }
      size_t tl;
      const char *tag = j.chunk.tag((size_t)GPUMAP_KEY_READ(k, bits), num, &tl);
      memcpy(p, tag, tl); p += tl;
      *p++ = '\t';
      const size_t nl = name_len[(size_t)a];
      memcpy(p, v.seq_name[a], nl); p += nl;
      *p++ = ':';
      p = put_num(p, f - v.seq_beg[a] + 1);
      *p++ = ':';
      *p++ = minus ? '-' : '+';
      *p++ = '\t';
      p = put_num(p, (int64_t)GPUMAP_KEY_SCORE(k));
      *p++ = '\n';
   }
   out.resize((size_t)(p - out.data()));
}

int write_all(int fd, const char *p, size_t n, int64_t at /* -1: at the file position */)
{
   while (n > 0) {
      ssize_t got = at >= 0 ? pwrite(fd, p, n, (off_t)at) : write(fd, p, n);
      if (got <= 0) return -1;
      p += got; n -= (size_t)got;
      if (at >= 0) at += got;
   }
   return 0;
}

}  // namespace

static int mapper_run(const char *qfile, const gpumap_index *index, gpumap_ctx *const *ctxs, int n_ctx, int tau, const char *out_path)
{
   gm_read_stream *in = nullptr;
   int fo = -1;
   bool fo_regular = false, fo_own = false;
   int64_t fo_off = 0, fo_alloc = 0;      // bytes written; bytes whose blocks are allocated
   gpumap_pipe *pipe = nullptr;
   Chan spare(256);              // chunk objects go round, their buffers are allocated once
   std::thread reader;
   int reader_rc = 0, writer_rc = 0;
   bool failed = false;
   const double t_enter = now_s();
   GM_FAIL_IF(qfile == nullptr, "argument 'qfile' is NULL.");
   GM_FAIL_IF(index == nullptr, "argument 'index' is NULL.");
   GM_FAIL_IF(ctxs == nullptr || n_ctx < 1 || ctxs[0] == nullptr, "argument 'ctx' is NULL.");
   in = gm_reads_open(qfile);
   GM_FAIL_IF(in == nullptr, "cannot open the read file.");
   if (out_path == nullptr || !strcmp(out_path, "-")) {
      fflush(stdout);
      fo = STDOUT_FILENO;
   } else {
      fo = open(out_path, O_WRONLY | O_CREAT | O_TRUNC, 0666);
      GM_FAIL_IF(fo < 0, "cannot open the output file.");
      fo_own = true;
      struct stat sb;
      fo_regular = fstat(fo, &sb) == 0 && S_ISREG(sb.st_mode);       // its parts can be written side by side
   }
   {
      // per GPU: one chunk on its way in, one being searched, one on its way out or being written
      const int depth = 3;
      // The chunks go to the device as the blocks of the file they are (parsed there, gm_parse.hpp) and come back as the
      // lines to write: the host only moves bytes.  GPUMAP_MAPPER_HOST_PARSE=1: the host parses the chunks and formats
      // the lines from compact keys (the earlier form; same output, kept for comparison).
      const bool on_device = !(getenv("GPUMAP_MAPPER_HOST_PARSE") && atoi(getenv("GPUMAP_MAPPER_HOST_PARSE")) != 0);
      const gpumap_index_view &v = index->view;
      if (on_device)
         for (int i = 0; i < n_ctx; i++)
            GM_FAIL_IF(gpumap_set_seq_names(ctxs[i], v.seq_name, v.seq_cnt) != 0, "cannot hand the sequences' names to the device.");
      pipe = gpumap_pipe_create(ctxs, n_ctx, depth, tau, on_device ? (GPUMAP_SRC_FILE | GPUMAP_PIPE_LINES) : (GPUMAP_SRC_TEXT | GPUMAP_PIPE_COMPACT));
      GM_FAIL_IF(pipe == nullptr, "cannot set up the pipeline.");
      std::vector<size_t> name_len((size_t)v.seq_cnt);
      for (int64_t i = 0; i < v.seq_cnt; i++) name_len[(size_t)i] = strlen(v.seq_name[i]);
      size_t chunk_bytes = (size_t)256 << 20;                     // IO_DEFAULT_BUFSIZE, src/io.h:13
      if (const char *e = getenv("GPUMAP_CHUNK_BYTES")) chunk_bytes = (size_t)std::max<long long>(1, atoll(e));
      const int T = gm_host_threads();
      double t_read = 0, t_parse = 0, t_format = 0, t_write = 0, t_wait = 0;
      const double t_begin = now_s();
      const int n_jobs = depth * n_ctx + 2;

      for (int i = 0; i < n_jobs; i++) {
         Job *j = new Job();
         j->chunk.text.pinned = true;                             // uploaded from where the parser writes it
         j->chunk.raw.pinned = on_device;                         // ... from where the file is read to
         spare.push(j);
      }
      reader = std::thread([&] {
         for (;;) {
            Job *j = spare.pop();
            if (j == nullptr) break;
            memset(&j->h, 0, sizeof j->h);
            j->n = on_device ? gm_reads_next_block(in, chunk_bytes, &j->chunk) : gm_reads_next(in, chunk_bytes, &j->chunk);
            t_read += j->chunk.t_read; t_parse += j->chunk.t_parse;
            if (j->n <= 0) { reader_rc = j->n < 0 ? -1 : 0; delete j; break; }
            fprintf(stderr, "read: %ld sequences.\n", (long)j->n);                 // src/mapper.c:30
            const int rc = on_device ? gpumap_pipe_submit(pipe, j->chunk.raw.data(), j->chunk.desc, j->n, j)
                                     : gpumap_pipe_submit(pipe, j->chunk.text.data(), j->chunk.offsets.data(), j->n, j);
            if (rc != 0) { delete j; reader_rc = -1; break; }
         }
         gpumap_pipe_close(pipe);
      });

      std::vector<std::vector<char>> parts((size_t)T);
      for (;;) {
         gpumap_hits h;
         void *user = nullptr;
         const double w_1 = now_s();
         const int got = gpumap_pipe_next(pipe, &h, &user);
         Job *j = (Job *)user;
         if (got == 0) break;
         if (got < 0) { failed = true; if (j && !spare.push(j)) delete j; break; }
         j->h = h;
         const int64_t L = h.n_locs;
         const double w0 = now_s();
         if (!on_device) gm_parallel_for(T, [&](int t) { format_range(*j, v, name_len, L * t / T, L * (t + 1) / T, parts[(size_t)t]); });
         const double w1 = now_s();
         if (on_device) {
            // the lines as they came from the device, slices of them written side by side
            // (buffered writes to one file take the inode's lock one at a time on ext4: 7 GB/s with 16 threads on this
            // platform, 12 GB/s into blocks allocated beforehand, tools/write_bench.cpp; so a few threads, and the
            // blocks of the next GiB allocated ahead of the writes, keeping the file's size)
            const int64_t B = h.line_bytes;
            if (fo_regular && fo_off + B > fo_alloc) {
#ifdef FALLOC_FL_KEEP_SIZE
               const int64_t step = std::max<int64_t>((int64_t)1 << 30, 2 * B);
               if (fallocate(fo, FALLOC_FL_KEEP_SIZE, (off_t)fo_alloc, (off_t)(fo_off + step - fo_alloc)) == 0) fo_alloc = fo_off + step;
               else fo_alloc = INT64_MAX;                  // (not supported here: no further attempts)
#endif
            }
            const int TW = std::min(T, 4);
            if (fo_regular && B >= ((int64_t)4 << 20)) {
               std::vector<int> bad((size_t)TW, 0);
               gm_parallel_for(TW, [&](int t) {
                  const int64_t a = B * t / TW, e = B * (t + 1) / TW;
                  if (e > a && write_all(fo, h.line_text + a, (size_t)(e - a), fo_off + a) != 0) bad[(size_t)t] = 1;
               });
               for (int x : bad) if (x) writer_rc = -1;
            } else if (B > 0 && write_all(fo, h.line_text, (size_t)B, fo_regular ? fo_off : -1) != 0) writer_rc = -1;
            fo_off += B;
         } else if (fo_regular) {
            std::vector<int64_t> at((size_t)T + 1, fo_off);
            for (int t = 0; t < T; t++) at[(size_t)t + 1] = at[(size_t)t] + (int64_t)parts[(size_t)t].size();
            std::vector<int> bad((size_t)T, 0);
            gm_parallel_for(T, [&](int t) {
               const auto &part = parts[(size_t)t];
               if (!part.empty() && write_all(fo, part.data(), part.size(), at[(size_t)t]) != 0) bad[(size_t)t] = 1;
            });
            for (int x : bad) if (x) writer_rc = -1;
            fo_off = at[(size_t)T];
         } else {
            for (auto &part : parts)
               if (!part.empty() && write_all(fo, part.data(), part.size(), -1) != 0) writer_rc = -1;
         }
         t_wait += w0 - w_1; t_format += w1 - w0; t_write += now_s() - w1;
         gpumap_pipe_release(pipe, &j->h);
         if (!spare.push(j)) delete j;
         if (writer_rc != 0) break;
      }
      // on an error the reader may be waiting for a chunk object or for a slot: the results still in flight are
      // taken and dropped until the pipe has been closed and has run empty
      spare.close();
      if (failed || writer_rc != 0) {
         gpumap_hits h;
         void *user = nullptr;
         int got;
         while ((got = gpumap_pipe_next(pipe, &h, &user)) != 0) {
            if (got > 0) gpumap_pipe_release(pipe, &h);
            delete (Job *)user;
         }
      }
      reader.join();
      spare.drain();
      if (getenv("GPUMAP_MAPPER_VERBOSE"))
         fprintf(stderr, "mapper stages (s): set-up %.3f | reader read %.3f parse %.3f | writer wait-for-gpu %.3f format %.3f write %.3f | wall %.3f\n",
                 t_begin - t_enter, t_read, t_parse, t_wait, t_format, t_write, now_s() - t_enter);
   }
   GM_FAIL_IF(failed, "batch failed.");
   GM_FAIL_IF(reader_rc != 0, "error while reading the read file.");
   GM_FAIL_IF(writer_rc != 0, "error while writing the output.");
   {
      const double t0 = now_s();
      gpumap_pipe_destroy(pipe);
      if (fo_own) {
         if (fo_regular && fo_alloc > fo_off && fo_alloc != INT64_MAX) (void)ftruncate(fo, (off_t)fo_off);      // gives back the blocks allocated ahead
         close(fo);
      }
      gm_reads_close(in);
      if (getenv("GPUMAP_MAPPER_VERBOSE")) fprintf(stderr, "mapper tear-down %.3f s, whole call %.3f s\n", now_s() - t0, now_s() - t_enter);
   }
   return 0;
failure_return:
   if (reader.joinable()) { spare.close(); if (pipe) gpumap_pipe_close(pipe); reader.join(); }
   gpumap_pipe_destroy(pipe);
   if (fo_own && fo >= 0) close(fo);
   gm_reads_close(in);
   return -1;
}

extern "C" int gpumap_mapper(const char *qfile, const gpumap_index *index, gpumap_ctx *ctx, int tau, const char *out_path)
{
   return mapper_run(qfile, index, &ctx, 1, tau, out_path);
}

// mapper() over several GPUs: `ctxs` are replicas of one index (gpumap_multi_ctx); the chunks of the read file go
// round the GPUs, the output is in input order, byte for byte what one GPU writes.
extern "C" int gpumap_mapper_multi(const char *qfile, const gpumap_index *index, gpumap_ctx *const *ctxs, int n_ctx, int tau,
                                   const char *out_path)
{
   return mapper_run(qfile, index, ctxs, n_ctx, tau, out_path);
}
