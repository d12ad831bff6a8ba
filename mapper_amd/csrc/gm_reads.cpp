// gm_reads.cpp -- chunked read input, behaviour of io_stream_read_seq (src/io.c:113-266) and
// seqread_parse_fastq / _fasta (src/seqread.c:119-164) without a getline, two strdup and a malloc
// per read: a chunk is one block of the file, its lines are found by several threads, letters go
// into one flat buffer with offsets (ready for a single upload) and tags stay where they are in the
// block.
//   first character '@' : FASTQ, 4 lines per read, tag = line 1 without '@', quality length checked
//                 '>' : FASTA, 2 lines per read (one sequence line, as in the reference)
//                 else: raw,   1 line per read, tag = its 1-based line number (src/io.c:171-174)
// A chunk ends at the first record boundary at which the lines read so far hold max_bytes or more
// (src/io.c:196-198); a line's content stops at a NUL (the reference handles lines as C strings).
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <mutex>
#include <thread>
#include "gm_host.hpp"

struct gm_read_stream {
   int     fd = -1;
   int64_t line = 1;          // src/io.c:61
   int     parser_lines = 0;  // 0 = not yet detected
   bool    eof = false;       // the reference's flag: set when a read attempt found the end of the file
   bool    file_end = false;  // no byte of the file is left unread
   bool    regular = false;   // a regular file: its size is known and slices can be read side by side
   int64_t file_size = 0, file_off = 0;
   gm_bytes carry;            // bytes read from the file that belong to the next chunk
};

// Page-locked blocks go round: allocating one costs about a millisecond per 10 MB, which a mapper() call over a few
// GB of reads would pay again for every chunk object and result arena.  Blocks given back are kept (up to
// GPUMAP_PINNED_POOL_BYTES, default 8 GiB) and handed out again to requests they fit; gpumap_host_trim frees them.
namespace {
struct PinnedPool {
   std::mutex m;
   std::vector<std::pair<void *, size_t>> blocks;
   size_t bytes = 0;
};
PinnedPool &pinned_pool() { static PinnedPool *p = new PinnedPool(); return *p; }      // (never destroyed: the HIP runtime may be gone by then)
}  // namespace

void *gm_pinned_take(size_t want, size_t *cap)
{
   {
      PinnedPool &pool = pinned_pool();
      std::lock_guard<std::mutex> l(pool.m);
      size_t best = pool.blocks.size();
      for (size_t i = 0; i < pool.blocks.size(); i++)
         if (pool.blocks[i].second >= want && (best == pool.blocks.size() || pool.blocks[i].second < pool.blocks[best].second)) best = i;
      if (best < pool.blocks.size() && pool.blocks[best].second <= 2 * want + ((size_t)16 << 20)) {
         void *p = pool.blocks[best].first;
         *cap = pool.blocks[best].second;
         pool.bytes -= *cap;
         pool.blocks.erase(pool.blocks.begin() + (long)best);
         return p;
      }
   }
   *cap = want;
   return gpumap_host_alloc(want);
}

void gm_pinned_give(void *p, size_t cap)
{
   if (!p) return;
   static const size_t limit = getenv("GPUMAP_PINNED_POOL_BYTES") ? (size_t)atoll(getenv("GPUMAP_PINNED_POOL_BYTES")) : ((size_t)8 << 30);
   {
      PinnedPool &pool = pinned_pool();
      std::lock_guard<std::mutex> l(pool.m);
      if (pool.bytes + cap <= limit && pool.blocks.size() < 64) { pool.blocks.emplace_back(p, cap); pool.bytes += cap; return; }
   }
   gpumap_host_free(p);
}

extern "C" void gpumap_host_trim(void)
{
   PinnedPool &pool = pinned_pool();
   std::lock_guard<std::mutex> l(pool.m);
   for (auto &b : pool.blocks) gpumap_host_free(b.first);
   pool.blocks.clear();
   pool.bytes = 0;
}

void gm_bytes::release()
{
   if (p) { if (pinned) gm_pinned_give(p, cap); else free(p); }
   p = nullptr; n = cap = 0;
}

bool gm_bytes::reserve(size_t c)
{
   if (c <= cap) return true;
   c = std::max(c, cap + cap / 2);
   if (!pinned) {
      char *q = (char *)realloc(p, c);
      if (!q) return false;
      p = q; cap = c;
      return true;
   }
   size_t got = 0;
   char *q = (char *)gm_pinned_take(c, &got);
   if (!q) return false;
   if (n) memcpy(q, p, n);
   if (p) gm_pinned_give(p, cap);
   p = q; cap = got;
   return true;
}

void gm_parallel_for(int threads, const std::function<void(int)> &fn)
{
   if (threads <= 1) { fn(0); return; }
   std::vector<std::thread> pool;
   pool.reserve((size_t)threads - 1);
   for (int t = 1; t < threads; t++) pool.emplace_back(fn, t);
   fn(0);
   for (auto &th : pool) th.join();
}

int gm_host_threads()
{
   const char *e = getenv("GPUMAP_HOST_THREADS");
   int t = e ? atoi(e) : (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
   return std::max(1, std::min(t, 64));
}

gm_read_stream *gm_reads_open(const char *path)
{
   gm_read_stream *s = nullptr;
   int fd = -1;
   struct stat sb;
   GM_FAIL_IF(path == nullptr, "argument 'path' is NULL.");
   fd = open(path, O_RDONLY);
   GM_FAIL_IF(fd < 0, "cannot open file.");
   s = new gm_read_stream();
   s->fd = fd;
   if (fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode)) { s->regular = true; s->file_size = (int64_t)sb.st_size; }
   return s;
failure_return:
   return nullptr;
}

void gm_reads_close(gm_read_stream *s)
{
   if (!s) return;
   if (s->fd >= 0) close(s->fd);
   delete s;
}

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// Appends file bytes to `raw` until it holds `want` bytes or the file ends.  A regular file is read in
// slices side by side (the copy out of the page cache is what takes the time).
// (the plain loop runs at 2 GB/s per thread as compiled here: a third of the time of reading the file)
static size_t count_newlines(const char *p, size_t n)
{
   size_t c = 0, i = 0;
#if defined(__SSE2__)
   const __m128i nl = _mm_set1_epi8('\n'), zero = _mm_setzero_si128();
   while (i + 16 <= n) {
      __m128i acc = zero;                                   // bytes count up to 255 matches each
      const size_t stop = std::min(n - 15, i + 16 * 255);
      for (; i < stop; i += 16) acc = _mm_sub_epi8(acc, _mm_cmpeq_epi8(_mm_loadu_si128((const __m128i *)(p + i)), nl));
      const __m128i sum = _mm_sad_epu8(acc, zero);
      c += (size_t)_mm_cvtsi128_si64(sum) + (size_t)_mm_cvtsi128_si64(_mm_unpackhi_epi64(sum, sum));
   }
#endif
   for (; i < n; i++) c += p[i] == '\n';
   return c;
}

// nl_below (or nullptr): += the newlines among the bytes appended that lie at positions below count_limit of `raw`
// (counted piece by piece right after the piece was read, while it is in the cache).
static int fill(gm_read_stream *s, gm_bytes &raw, size_t want, int threads, size_t count_limit = 0, size_t *nl_below = nullptr)
{
   if (raw.size() >= want || s->file_end) return 0;
   if (!raw.reserve(want)) return -1;
   if (s->regular) {
      const size_t have = raw.size();
      const size_t take = (size_t)std::min<int64_t>((int64_t)(want - have), std::max<int64_t>(0, s->file_size - s->file_off));
      const int T = take >= ((size_t)8 << 20) ? threads : 1;
      std::vector<int> bad((size_t)T, 0);
      std::vector<size_t> cnt((size_t)T, 0);
      gm_parallel_for(T, [&](int t) {
         size_t a = take * (size_t)t / (size_t)T;
         const size_t b = take * (size_t)(t + 1) / (size_t)T;
         while (a < b) {
            const size_t piece = nl_below ? std::min<size_t>(b - a, (size_t)128 << 10) : b - a;      // (counted from the core's own cache)
            ssize_t got = pread(s->fd, raw.data() + have + a, piece, (off_t)(s->file_off + (int64_t)a));
            if (got <= 0) { bad[(size_t)t] = 1; return; }         // the file shrank, or an I/O error
            if (nl_below && have + a < count_limit)
               cnt[(size_t)t] += count_newlines(raw.data() + have + a, std::min((size_t)got, count_limit - (have + a)));
            a += (size_t)got;
         }
      });
      for (int x : bad) if (x) return -1;
      if (nl_below) for (size_t c : cnt) *nl_below += c;
      raw.n = have + take;
      s->file_off += (int64_t)take;
      if (s->file_off >= s->file_size) s->file_end = true;
      return 0;
   }
   while (raw.size() < want && !s->file_end) {
      ssize_t got = read(s->fd, raw.data() + raw.size(), want - raw.size());
      if (got < 0) return -1;
      if (nl_below && raw.size() < count_limit)
         *nl_below += count_newlines(raw.data() + raw.size(), std::min((size_t)got, count_limit - raw.size()));
      raw.n += (size_t)got;
      if (got == 0) s->file_end = true;
   }
   return 0;
}

int64_t gm_reads_next(gm_read_stream *s, size_t max_bytes, gm_read_chunk *chunk)
{
   GM_FAIL_IF(s == nullptr || chunk == nullptr, "NULL argument.");
   GM_FAIL_IF(max_bytes > ((size_t)1 << 31), "chunks hold at most 2 GiB.");
   chunk->raw.clear(); chunk->text.clear(); chunk->offsets.assign(1, 0);
   chunk->tag_off.clear(); chunk->tag_len.clear(); chunk->first_line = s->line; chunk->numbered = false;
   chunk->t_read = chunk->t_parse = 0;
   if (s->eof) return 0;                                         // src/io.c:128-130
   {
      gm_bytes &raw = chunk->raw;
      const size_t slack = (size_t)1 << 20;
      const int T = gm_host_threads();
      const double t0 = now_s();
      GM_FAIL_IF(!raw.resize(s->carry.size()), "out of memory.");
      if (s->carry.size()) memcpy(raw.data(), s->carry.data(), s->carry.size());
      s->carry.clear();
      GM_FAIL_IF(fill(s, raw, std::max<size_t>(max_bytes, 1) + slack, T) != 0, "error while reading the file.");
      double t_read = now_s() - t0;
      if (s->parser_lines == 0)                                   // io_stream_set_parser, src/io.c:232-266
         s->parser_lines = raw.empty() ? 1 : (raw.data()[0] == '@' ? 4 : (raw.data()[0] == '>' ? 2 : 1));
      const size_t P = (size_t)s->parser_lines;
      std::vector<std::vector<uint64_t>> part((size_t)T);
      std::vector<uint64_t> &eol = chunk->eol;                    // end of every line (one past its '\n', or the end of the file)
      eol.clear();
      size_t scanned = 0, n_lines = 0, used = 0;
      bool at_end = false;                                        // the chunk takes every line the file still has
      for (;;) {
         // ends of the lines in raw[scanned, size): slices of the block, one per thread
         const size_t lo = scanned, hi = raw.size();
         const char *base = raw.data();
         gm_parallel_for(hi - lo >= ((size_t)1 << 20) ? T : 1, [&](int t) {
            const int TT = hi - lo >= ((size_t)1 << 20) ? T : 1;
            std::vector<uint64_t> &v = part[(size_t)t];
            v.clear();
            const size_t a = lo + (hi - lo) * (size_t)t / (size_t)TT, b = lo + (hi - lo) * (size_t)(t + 1) / (size_t)TT;
            v.reserve((b - a) / 48 + 16);
            const char *q = base + a, *e = base + b;
            while (q < e && (q = (const char *)memchr(q, '\n', (size_t)(e - q))) != nullptr) { v.push_back((uint64_t)(q - base) + 1); q++; }
         });
         {  // the slices' line ends one after another (copied side by side: 4 M entries per chunk)
            std::vector<size_t> at(part.size() + 1, eol.size());
            for (size_t t = 0; t < part.size(); t++) at[t + 1] = at[t] + part[t].size();
            eol.resize(at.back());
            auto copy = [&](size_t t) {
               if (!part[t].empty()) memcpy(eol.data() + at[t], part[t].data(), 8 * part[t].size());
               part[t].clear();
            };
            if (at.back() - at[0] >= ((size_t)1 << 16)) gm_parallel_for((int)part.size(), [&](int t) { copy((size_t)t); });
            else for (size_t t = 0; t < part.size(); t++) copy(t);
         }
         scanned = hi;
         // first record boundary with max_bytes or more behind it
         size_t k = (size_t)(std::lower_bound(eol.begin(), eol.end(), (uint64_t)std::max<size_t>(max_bytes, 1)) - eol.begin());   // lines 0..k hold >= max_bytes
         size_t need = ((k + 1 + P - 1) / P) * P;
         if (need <= eol.size()) { n_lines = need; used = (size_t)eol[need - 1]; break; }
         if (s->file_end) {
            if (eol.empty() ? !raw.empty() : eol.back() < raw.size()) eol.push_back(raw.size());   // last line without '\n'
            n_lines = eol.size(); used = raw.size(); at_end = true;
            break;
         }
         const double r0 = now_s();
         GM_FAIL_IF(fill(s, raw, raw.size() + std::max(slack, raw.size() / 2), T) != 0, "error while reading the file.");
         t_read += now_s() - r0;
      }
      const double t1 = now_s();
      if (at_end) s->eof = true;                                  // getline returned -1 inside this call
      GM_FAIL_IF(!s->carry.resize(raw.size() - used), "out of memory.");
      if (raw.size() > used) memcpy(s->carry.data(), raw.data() + used, raw.size() - used);
      const size_t n_rec = n_lines / P;                           // an incomplete trailing record is dropped,
      s->line += (int64_t)n_lines;                                // as the reference never hands it to the parser
      const char *base = raw.data();
      auto line_beg = [&](size_t i) { return i == 0 ? (size_t)0 : (size_t)eol[i - 1]; };
      auto line_len = [&](size_t i) {                             // content: without '\n', up to a NUL
         size_t b = line_beg(i), e = (size_t)eol[i];
         if (e > b && base[e - 1] == '\n') e--;
         const void *z = memchr(base + b, 0, e - b);
         return z ? (size_t)((const char *)z - (base + b)) : e - b;
      };
      const size_t seq_line = P == 1 ? 0 : 1;
      chunk->offsets.resize(n_rec + 1);
      chunk->offsets[0] = 0;
      chunk->numbered = P == 1;
      if (P != 1) { chunk->tag_off.resize(n_rec); chunk->tag_len.resize(n_rec); }
      const int TP = n_rec >= 4096 ? T : 1;
      std::vector<int> bad((size_t)TP, 0);
      std::vector<int64_t> part_sum((size_t)TP + 1, 0);
      gm_parallel_for(TP, [&](int t) {
         const size_t a = n_rec * (size_t)t / (size_t)TP, b = n_rec * (size_t)(t + 1) / (size_t)TP;
         int64_t sum = 0;
         for (size_t r = a; r < b; r++) {
            const size_t len = line_len(P * r + seq_line);
            sum += (int64_t)len;
            chunk->offsets[r + 1] = sum;                          // relative to the slice; shifted below
            if (P != 1) {
               const size_t tl = line_len(P * r);
               chunk->tag_off[r] = (uint32_t)(line_beg(P * r) + (tl > 0 ? 1 : 0));       // lines[0] + 1
               chunk->tag_len[r] = (uint32_t)(tl > 0 ? tl - 1 : 0);
            }
            if (P == 4 && line_len(P * r + 3) != len) bad[(size_t)t] = 1;                // src/seqread.c:47-49
         }
         part_sum[(size_t)t + 1] = sum;
      });
      for (int x : bad) GM_FAIL_IF(x, "inconsistent length of sequence quality.");
      for (int t = 0; t < TP; t++) part_sum[(size_t)t + 1] += part_sum[(size_t)t];
      GM_FAIL_IF(!chunk->text.resize((size_t)part_sum[(size_t)TP]), "out of memory.");
      gm_parallel_for(TP, [&](int t) {
         const size_t a = n_rec * (size_t)t / (size_t)TP, b = n_rec * (size_t)(t + 1) / (size_t)TP;
         const int64_t shift = part_sum[(size_t)t];
         for (size_t r = a; r < b; r++) {
            const int64_t beg = r == a ? shift : chunk->offsets[r];                // offsets[r] of r > a is shifted already
            chunk->offsets[r + 1] += shift;
            memcpy(chunk->text.data() + beg, base + line_beg(P * r + seq_line), (size_t)(chunk->offsets[r + 1] - beg));
         }
      });
      chunk->t_read = t_read;
      chunk->t_parse = now_s() - t1 + (t1 - t0 - t_read);
      return (int64_t)n_rec;
   }
failure_return:
   return -1;
}

// The next chunk as the block of the file it is cut from, not parsed: the cut is the one gm_reads_next makes (the first
// record boundary with max_bytes or more before it, src/io.c:196-198), found from the number of newlines before
// max_bytes (counted while the block is read) and a short walk from there.  The block goes to the device as it is and
// is parsed there (gm_parse.hpp).  chunk->desc = the description GPUMAP_SRC_FILE asks for.
int64_t gm_reads_next_block(gm_read_stream *s, size_t max_bytes, gm_read_chunk *chunk)
{
   GM_FAIL_IF(s == nullptr || chunk == nullptr, "NULL argument.");
   GM_FAIL_IF(max_bytes > ((size_t)1 << 30), "chunks hold at most 1 GiB.");
   chunk->raw.clear(); chunk->first_line = s->line; chunk->t_read = chunk->t_parse = 0;
   for (int64_t &d : chunk->desc) d = 0;
   if (s->eof) return 0;                                         // src/io.c:128-130
   {
      gm_bytes &raw = chunk->raw;
      const size_t slack = (size_t)1 << 20;
      const int T = gm_host_threads();
      const double t0 = now_s();
      const size_t limit = std::max<size_t>(max_bytes, 1) - 1;   // newlines below this position end lines that end before max_bytes
      size_t k = 0;
      GM_FAIL_IF(!raw.resize(s->carry.size()), "out of memory.");
      if (s->carry.size()) memcpy(raw.data(), s->carry.data(), s->carry.size());
      k += count_newlines(raw.data(), std::min(raw.size(), limit));
      s->carry.clear();
      GM_FAIL_IF(fill(s, raw, std::max<size_t>(max_bytes, 1) + slack, T, limit, &k) != 0, "error while reading the file.");
      if (s->parser_lines == 0)                                   // io_stream_set_parser, src/io.c:232-266
         s->parser_lines = raw.empty() ? 1 : (raw.data()[0] == '@' ? 4 : (raw.data()[0] == '>' ? 2 : 1));
      const size_t P = (size_t)s->parser_lines;
      const size_t need = ((k + 1 + P - 1) / P) * P;             // lines of the chunk: whole records that reach max_bytes
      size_t found = 0, q = std::min(limit, raw.size()), used = 0, n_lines = 0;
      bool at_end = false;
      for (;;) {
         while (found < need - k) {
            const char *nl = q < raw.size() ? (const char *)memchr(raw.data() + q, '\n', raw.size() - q) : nullptr;
            if (!nl) break;
            found++;
            q = (size_t)(nl - raw.data()) + 1;
         }
         if (found == need - k) { used = q; n_lines = need; break; }
         if (s->file_end) {
            n_lines = k + found + ((!raw.empty() && raw.data()[raw.size() - 1] != '\n') ? 1 : 0);   // a last line without newline
            used = raw.size(); at_end = true;
            break;
         }
         q = raw.size();
         GM_FAIL_IF(fill(s, raw, raw.size() + std::max(slack, raw.size() / 2), T) != 0, "error while reading the file.");
      }
      if (at_end) s->eof = true;                                  // getline returned -1 inside this call
      GM_FAIL_IF(!s->carry.resize(raw.size() - used), "out of memory.");
      if (raw.size() > used) memcpy(s->carry.data(), raw.data() + used, raw.size() - used);
      raw.n = used;
      const size_t n_rec = n_lines / P;                           // an incomplete trailing record is dropped
      s->line += (int64_t)n_lines;
      chunk->desc[0] = (int64_t)used; chunk->desc[1] = (int64_t)P; chunk->desc[2] = (int64_t)n_lines; chunk->desc[3] = 0;
      chunk->desc[4] = chunk->first_line;
      chunk->numbered = P == 1;
      chunk->t_read = now_s() - t0;
      return (int64_t)n_rec;
   }
failure_return:
   return -1;
}

// Host-only helper: parses `path` exactly as gpumap_mapper does and writes "tag<TAB>letters" lines,
// with a "#chunk" line between chunks.  Returns the number of reads or -1.  Lets the read pipeline be
// checked against the reference's io tests (test/test_io.c, test/examples/io_input.*) without a GPU.
extern "C" int64_t gpumap_reads_dump(const char *path, size_t max_bytes, const char *out_path)
{
   gm_read_stream *s = gm_reads_open(path);
   if (!s) return -1;
   FILE *fo = fopen(out_path, "w");
   if (!fo) { gm_reads_close(s); return -1; }
   gm_read_chunk chunk;
   int64_t n, total = 0;
   char num[32];
   while ((n = gm_reads_next(s, max_bytes, &chunk)) > 0) {
      if (total) fprintf(fo, "#chunk\n");
      for (int64_t i = 0; i < n; i++) {
         size_t tl;
         const char *tag = chunk.tag((size_t)i, num, &tl);
         fwrite(tag, 1, tl, fo);
         fputc('\t', fo);
         fwrite(chunk.text.data() + chunk.offsets[(size_t)i], 1, (size_t)(chunk.offsets[(size_t)i + 1] - chunk.offsets[(size_t)i]), fo);
         fputc('\n', fo);
      }
      total += n;
   }
   fclose(fo);
   gm_reads_close(s);
   return n < 0 ? -1 : total;
}

// Host-only helper: cuts `path` into the blocks gpumap_mapper sends to the device (gm_reads_next_block) and writes, per
// block, a line "#block <records> <bytes> <lines per record> <lines> <first line>" followed by the block's bytes.
// Returns the number of reads or -1.  Lets the cut be checked against the line-by-line model without a GPU.
extern "C" int64_t gpumap_reads_dump_blocks(const char *path, size_t max_bytes, const char *out_path)
{
   gm_read_stream *s = gm_reads_open(path);
   if (!s) return -1;
   FILE *fo = fopen(out_path, "w");
   if (!fo) { gm_reads_close(s); return -1; }
   gm_read_chunk chunk;
   int64_t n, total = 0;
   while ((n = gm_reads_next_block(s, max_bytes, &chunk)) > 0) {
      fprintf(fo, "#block %ld %ld %ld %ld %ld\n", (long)n, (long)chunk.desc[0], (long)chunk.desc[1], (long)chunk.desc[2], (long)chunk.desc[4]);
      fwrite(chunk.raw.data(), 1, chunk.raw.size(), fo);
      total += n;
   }
   fclose(fo);
   gm_reads_close(s);
   return n < 0 ? -1 : total;
}
