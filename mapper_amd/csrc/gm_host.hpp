// gm_host.hpp -- host-side declarations shared by the translation units of libgpumap.so.
#pragma once
#include <cerrno>
#include <cstdint>
#include <cstdio>
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

// FASTQ / FASTA / raw reader (gm_reads.cpp).
struct gm_read_chunk {
   std::vector<char>        text;     // concatenated read letters
   std::vector<int64_t>     offsets;  // n+1
   std::vector<std::string> tags;
};
struct gm_read_stream;
gm_read_stream *gm_reads_open(const char *path);
// Returns number of reads (0 at EOF, -1 on error); at most `max_bytes` of input per chunk.
int64_t gm_reads_next(gm_read_stream *s, size_t max_bytes, gm_read_chunk *chunk);
void gm_reads_close(gm_read_stream *s);
