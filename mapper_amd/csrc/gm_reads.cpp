// gm_reads.cpp -- chunked read input, behaviour of io_stream_read_seq (src/io.c:113-266) and
// seqread_parse_fastq / _fasta (src/seqread.c:119-164), without one malloc per line: letters of a
// chunk go into one flat buffer with offsets, ready for a single upload.
//   first character '@' : FASTQ, 4 lines per read, tag = line 1 without '@', quality length checked
//                 '>' : FASTA, 2 lines per read (one sequence line, as in the reference)
//                 else: raw,   1 line per read, tag = its 1-based line number (src/io.c:171-174)
#include <cstdlib>
#include <cstring>
#include "gm_host.hpp"

struct gm_read_stream {
   FILE   *fin = nullptr;
   int64_t line = 1;          // src/io.c:61
   int     parser_lines = 0;  // 0 = not yet detected
   bool    eof = false;
   char   *buf = nullptr;
   size_t  cap = 0;
};

gm_read_stream *gm_reads_open(const char *path)
{
   gm_read_stream *s = nullptr;
   FILE *f = nullptr;
   GM_FAIL_IF(path == nullptr, "argument 'path' is NULL.");
   f = fopen(path, "r");
   GM_FAIL_IF(f == nullptr, "cannot open file.");
   s = new gm_read_stream();
   s->fin = f;
   return s;
failure_return:
   return nullptr;
}

void gm_reads_close(gm_read_stream *s)
{
   if (!s) return;
   if (s->fin) fclose(s->fin);
   free(s->buf);
   delete s;
}

int64_t gm_reads_next(gm_read_stream *s, size_t max_bytes, gm_read_chunk *chunk)
{
   GM_FAIL_IF(s == nullptr || chunk == nullptr, "NULL argument.");
   chunk->text.clear(); chunk->offsets.assign(1, 0); chunk->tags.clear();
   if (s->eof) return 0;                                         // src/io.c:128-130
   if (s->parser_lines == 0) {                                   // io_stream_set_parser, src/io.c:232-266
      int c = fgetc(s->fin);
      s->parser_lines = c == '@' ? 4 : (c == '>' ? 2 : 1);
      if (c != EOF) ungetc(c, s->fin);
   }
   {
      size_t bytes_cnt = 0, lineno = 0, seq_len = 0;
      std::string tag;
      ssize_t bytes;
      while ((bytes = getline(&s->buf, &s->cap, s->fin)) > 0) {
         if (s->buf[bytes - 1] == '\n') s->buf[--bytes] = 0;
         size_t len = strlen(s->buf);                             // strdup semantics: stops at a NUL
         size_t fline = lineno % (size_t)s->parser_lines;
         if (s->parser_lines == 1) {
            chunk->tags.push_back(std::to_string(s->line));
            chunk->text.insert(chunk->text.end(), s->buf, s->buf + len);
            chunk->offsets.push_back((int64_t)chunk->text.size());
         } else if (fline == 0) {
            tag.assign(len > 0 ? s->buf + 1 : s->buf);            // lines[0] + 1
         } else if (fline == 1) {
            chunk->text.insert(chunk->text.end(), s->buf, s->buf + len);
            seq_len = len;
            if (s->parser_lines == 2) { chunk->tags.push_back(tag); chunk->offsets.push_back((int64_t)chunk->text.size()); }
         } else if (fline == 3) {
            GM_FAIL_IF(len != seq_len, "inconsistent length of sequence quality.");   // src/seqread.c:47-49
            chunk->tags.push_back(tag);
            chunk->offsets.push_back((int64_t)chunk->text.size());
         }
         bytes_cnt += (size_t)bytes;
         s->line++;
         lineno++;
         if (bytes_cnt >= max_bytes && lineno % (size_t)s->parser_lines == 0) break;   // src/io.c:196-198
      }
      if (bytes == -1) s->eof = true;
      // an incomplete trailing record is dropped, as the reference never hands it to the parser
      chunk->text.resize((size_t)chunk->offsets.back());
   }
   return (int64_t)chunk->tags.size();
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
   while ((n = gm_reads_next(s, max_bytes, &chunk)) > 0) {
      if (total) fprintf(fo, "#chunk\n");
      for (int64_t i = 0; i < n; i++) {
         fprintf(fo, "%s\t", chunk.tags[(size_t)i].c_str());
         fwrite(chunk.text.data() + chunk.offsets[(size_t)i], 1, (size_t)(chunk.offsets[(size_t)i + 1] - chunk.offsets[(size_t)i]), fo);
         fputc('\n', fo);
      }
      total += n;
   }
   fclose(fo);
   gm_reads_close(s);
   return n < 0 ? -1 : total;
}
