// gm_index.cpp -- host-side index handle: reading the reference's on-disk index and building
// one from FASTA.  One-off host work, off the query hot path; the file formats are the drop-in
// contract and are reproduced byte for byte.
//
//   gpumap_index_read   ~ index_read   (src/index.c:7-96)
//   gpumap_index_build  ~ index_build  (src/index.c:99-162): read_fasta (:242-331),
//                         txt_commit_seq / txt_commit_rc (src/index_txt.c:224-343),
//                         sar_build + compact_array (src/index_sar.c:19-70,283-328),
//                         bwt_build_opt (src/index_bwt.c:460-591), *_file_write.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <cstdlib>
#include <cstring>
#include <thread>
#include "gm_host.hpp"

namespace {

const char *DNA_ALPHABET[5] = {"Aa", "Cc", "Gg", "Tt", "Nn"};   // src/index_sym.c:11
const uint8_t DNA_COMPLEMENT[6] = {3, 2, 1, 0, 4, 5};           // src/index_sym.c:12, identity base :210-212
constexpr int SYM_COUNT = 5;
constexpr int DEFAULT_SYM = 4;                                    // src/index_sym.c:13
constexpr int64_t MARK_INTV = 14;                                 // BWT_OCC_MARK_INTV_DEF, src/index_bwt.h:26

int map_file(const std::string &path, gm_mapping *m)
{
   int fd = open(path.c_str(), O_RDONLY);
   if (fd < 0) return -1;
   struct stat sb;
   if (fstat(fd, &sb) != 0) { close(fd); return -1; }
   m->len = (size_t)sb.st_size;
   m->ptr = mmap(nullptr, m->len, PROT_READ, MAP_SHARED, fd, 0);
   close(fd);
   if (m->ptr == MAP_FAILED) { m->ptr = nullptr; return -1; }
   return 0;
}

int write_all(int fd, const void *buf, size_t len)
{
   const char *p = (const char *)buf;
   while (len > 0) {
      ssize_t w = write(fd, p, len > ((size_t)1 << 30) ? ((size_t)1 << 30) : len);
      if (w < 0) return -1;
      p += w; len -= (size_t)w;
   }
   return 0;
}

void fill_sym_tables(gpumap_index *ix)
{
   memset(ix->sym_table, DEFAULT_SYM, 256);                      // src/index_sym.c:126
   for (int i = 0; i < SYM_COUNT; i++)
      for (const char *c = DNA_ALPHABET[i]; *c; c++) ix->sym_table[(uint8_t)*c] = (uint8_t)i;
   memcpy(ix->com_table, DNA_COMPLEMENT, 6);
   memcpy(ix->canon, "ACGTN", 6);
}

}  // namespace

extern "C" void gpumap_index_free(gpumap_index *ix)
{
   if (!ix) return;
   for (gm_mapping *m : {&ix->m_sym, &ix->m_txt, &ix->m_sar, &ix->m_bwt})
      if (m->ptr) munmap(m->ptr, m->len);
   delete ix;
}

extern "C" int gpumap_index_view_of(const gpumap_index *ix, gpumap_index_view *view)
{
   GM_FAIL_IF(ix == nullptr, "argument 'index' is NULL.");
   GM_FAIL_IF(view == nullptr, "argument 'view' is NULL.");
   *view = ix->view;
   return 0;
failure_return:
   return -1;
}

extern "C" gpumap_index *gpumap_index_read(const char *filename_base)
{
   gpumap_index *ix = nullptr;
   GM_FAIL_IF(filename_base == nullptr, "argument 'filename_base' is NULL.");
   ix = new gpumap_index();
   ix->base = filename_base;
   memset(&ix->view, 0, sizeof ix->view);
   GM_FAIL_IF(map_file(ix->base + ".sym", &ix->m_sym) != 0, "cannot map '.sym' file.");
   GM_FAIL_IF(map_file(ix->base + ".txt", &ix->m_txt) != 0, "cannot map '.txt' file.");
   GM_FAIL_IF(map_file(ix->base + ".sar", &ix->m_sar) != 0, "cannot map '.sar' file.");
   GM_FAIL_IF(map_file(ix->base + ".bwt", &ix->m_bwt) != 0, "cannot map '.bwt' file.");
   {
      gpumap_index_view &v = ix->view;
      // .sym: u64 magic, u8 sym_count, char canon[count+1], u8 table[256], u8 com[count+1]
      const uint8_t *s = (const uint8_t *)ix->m_sym.ptr;
      uint64_t magic;
      GM_FAIL_IF(ix->m_sym.len < 9, "'sym' file is too small.");
      memcpy(&magic, s, 8);
      GM_FAIL_IF(magic != GM_MAGIC_SYM, "unrecognized 'sym' file format (magicno).");
      v.sym_count = s[8];
      GM_FAIL_IF(v.sym_count != SYM_COUNT, "only the 5-symbol DNA alphabet is supported.");
      GM_FAIL_IF(ix->m_sym.len < (size_t)(9 + 2 * (v.sym_count + 1) + 256), "'sym' file is too small.");
      memcpy(ix->canon, s + 9, v.sym_count + 1);
      memcpy(ix->sym_table, s + 9 + v.sym_count + 1, 256);
      memcpy(ix->com_table, s + 9 + v.sym_count + 1 + 256, v.sym_count + 1);
      v.sym_table = ix->sym_table;
      v.com_table = ix->com_table;

      // .txt: u64 magic, i64 txt_len, seq_cnt, wil_cnt, rc_flag, seq_len[], seq_beg[], names, text
      GM_FAIL_IF(ix->m_txt.len <= 40, "'txt' file size is too small.");
      const int64_t *t = (const int64_t *)ix->m_txt.ptr;
      GM_FAIL_IF((uint64_t)t[0] != GM_MAGIC_TXT, "unrecognized 'txt' file format (magicno).");
      v.txt_len = t[1]; v.seq_cnt = t[2]; v.wil_cnt = t[3]; v.rc_flag = t[4];
      GM_FAIL_IF(v.txt_len < 1, "incorrect 'txt' file format (txt_len < 1).");
      GM_FAIL_IF(v.seq_cnt < 1, "incorrect 'txt' file format (seq_cnt < 1).");
      GM_FAIL_IF(v.rc_flag < 0 || v.rc_flag > 1, "incorrect 'txt' file format (rc_flag).");
      GM_FAIL_IF(ix->m_txt.len < (size_t)(40 + 16 * v.seq_cnt + v.txt_len), "'txt' file is truncated.");
      v.seq_len = t + 5;
      v.seq_beg = v.seq_len + v.seq_cnt;
      const char *p = (const char *)(v.seq_beg + v.seq_cnt);
      const char *file_end = (const char *)ix->m_txt.ptr + ix->m_txt.len;
      for (int64_t i = 0; i < v.seq_cnt; i++) {
         ix->name_ptrs.push_back(p);
         size_t l = strnlen(p, (size_t)(file_end - p));
         GM_FAIL_IF(p + l >= file_end, "'txt' file is truncated (names).");
         p += l + 1;
      }
      v.seq_name = ix->name_ptrs.data();
      GM_FAIL_IF(p + v.txt_len > file_end, "'txt' file is truncated (text).");
      v.text = (const uint8_t *)p;

      // .sar: u64 magic, i64 sar_bits, sar_len, txt_len, i64 sa[sar_len]
      GM_FAIL_IF(ix->m_sar.len < 32, "'sar' file size is too small.");
      const int64_t *a = (const int64_t *)ix->m_sar.ptr;
      GM_FAIL_IF((uint64_t)a[0] != GM_MAGIC_SAR, "unrecognized 'sar' file format (magicno).");
      v.sar_bits = a[1]; v.sar_len = a[2];
      GM_FAIL_IF(a[3] != v.txt_len, "'sar' and 'txt' files disagree on the text length.");
      GM_FAIL_IF(v.sar_bits < 1 || v.sar_bits > 63, "incorrect 'sar' file format (sar_bits).");
      GM_FAIL_IF(ix->m_sar.len < (size_t)(32 + 8 * v.sar_len), "'sar' file is truncated.");
      GM_FAIL_IF((v.txt_len * v.sar_bits + 63) / 64 > v.sar_len, "'sar' file is truncated (sar_len).");
      v.sa = a + 4;

      // .bwt: u64 magic, i64 occ_length, mark_intv, word_size, mark_bits, u64 c[6], u64 occ[]
      GM_FAIL_IF(ix->m_bwt.len < 48 + 40, "'bwt' file is too small.");
      const uint64_t *b = (const uint64_t *)ix->m_bwt.ptr;
      GM_FAIL_IF(b[0] != GM_MAGIC_BWT, "unrecognized 'bwt' file format (magicno).");
      v.occ_length = (int64_t)b[1]; v.occ_mark_intv = (int64_t)b[2]; v.occ_word_size = (int64_t)b[3];
      GM_FAIL_IF(v.occ_word_size != 64, "unsupported 'bwt' word size.");
      GM_FAIL_IF(v.occ_mark_intv < 1, "incorrect 'bwt' file format (mark_intv).");
      GM_FAIL_IF(ix->m_bwt.len < (size_t)(40 + 8 * (v.sym_count + 1) + 8 * v.occ_length), "'bwt' file is truncated.");
      v.c = b + 5;
      v.occ = v.c + v.sym_count + 1;
   }
   return ix;
failure_return:
   gpumap_index_free(ix);
   return nullptr;
}

// ------------------------------------------------------------------------------------------------
// Build
// ------------------------------------------------------------------------------------------------

namespace {

// read_fasta (src/index.c:242-331) + txt_append / txt_commit_seq (src/index_txt.c:153-283).
// Header: name runs from the first non-space after '>' to the next space.  Sequence lines lose
// only their trailing '\n'; every other character that is not ACGTacgt becomes N.
int load_fasta(const char *path, gpumap_index *ix)
{
   FILE *f = fopen(path, "r");
   char *line = nullptr; size_t cap = 0; ssize_t len;
   bool have_name = false;
   std::string name;
   auto commit = [&]() -> int {
      if (name.empty()) return -1;
      for (auto &n : ix->own_names) if (n == name) return -1;    // duplicate sequence name
      int64_t beg = ix->own_seq_beg.empty() ? 0 : ix->own_seq_beg.back() + ix->own_seq_len.back();
      ix->own_text.push_back(SYM_COUNT);                          // separator (wildcard)
      ix->own_seq_beg.push_back(beg);
      ix->own_seq_len.push_back((int64_t)ix->own_text.size() - beg);
      ix->own_names.push_back(name);
      return 0;
   };
   GM_FAIL_IF(f == nullptr, "cannot open FASTA file.");
   {
      int c0 = fgetc(f);
      GM_FAIL_IF(c0 != '>', "incorrect input format (fgetc).");
      ungetc(c0, f);
   }
   while ((len = getline(&line, &cap, f)) != -1) {
      if (line[0] == '>') {
         if (have_name) GM_FAIL_IF(commit() != 0, "empty or duplicate sequence name.");
         line[len - 1] = 0;
         int beg = 1;
         while (line[beg] == ' ') beg++;
         int k = beg;
         while (line[k] != ' ' && line[k] != 0) k++;
         GM_FAIL_IF(beg == k, "found empty sequence name.");
         name.assign(line + beg, (size_t)(k - beg));
         have_name = true;
      } else {
         if (len > 0 && line[len - 1] == '\n') line[--len] = 0;
         size_t l = strlen(line);                                  // txt_append stops at a NUL
         for (size_t i = 0; i < l; i++) ix->own_text.push_back(ix->sym_table[(uint8_t)line[i]]);
      }
   }
   GM_FAIL_IF(!have_name || commit() != 0, "empty or duplicate sequence name.");
   free(line);
   fclose(f);
   return 0;
failure_return:
   free(line);
   if (f) fclose(f);
   return -1;
}

// txt_commit_rc (src/index_txt.c:286-343): text becomes F $ revcomp(F) $, where F ends without its
// last separator; separators map to separators.  64-bit loop counters (the reference's `int i`,
// :322, overflows past 2^31 symbols).
void append_revcomp(gpumap_index *ix, int64_t *wil_cnt)
{
   std::vector<uint8_t> &t = ix->own_text;
   int64_t tlen = (int64_t)t.size();
   if (t[tlen - 1] < SYM_COUNT) { t.push_back(SYM_COUNT); (*wil_cnt)++; } else tlen--;
   t.resize((size_t)(2 * tlen + 1));
   for (int64_t i = 1; i <= tlen; i++) {
      uint8_t s = t[tlen - i];
      if (s < SYM_COUNT) t[tlen + i] = DNA_COMPLEMENT[s];
      else { t[tlen + i] = SYM_COUNT; (*wil_cnt)++; }
   }
   if (t.back() < SYM_COUNT) { t.push_back(SYM_COUNT); (*wil_cnt)++; }
}

// compact_array (src/index_sar.c:283-328): entries of `bits` bits, LSB first.  The reference packs
// in place and keeps one word past the last one it filled; that word still holds the unpacked
// entry that lived there (or 0 when the stream ends on a word boundary).  Reproduced so that the
// .sar file is byte-identical.
void pack_suffix_array(const std::vector<int64_t> &sa, int64_t n, int64_t *bits_out,
                       std::vector<int64_t> *packed)
{
   int bits = 0;
   while ((uint64_t)n > ((uint64_t)1 << bits)) bits++;
   if (bits == 0) bits = 0;   // n == 1: zero-bit entries; the reference's mask shift is undefined here
   const uint64_t total_bits = (uint64_t)n * (uint64_t)bits;
   const int64_t word = (int64_t)(total_bits / 64);          // last word index the packer touched
   const int lastbit = (int)(total_bits % 64);
   const int64_t words = word + 1 + (lastbit > 0 ? 1 : 0);
   packed->assign((size_t)words, 0);
   uint64_t *w = (uint64_t *)packed->data();
   for (int64_t i = 0; i < n; i++) {
      uint64_t v = (uint64_t)sa[i];
      uint64_t b = (uint64_t)i * (uint64_t)bits;
      w[b >> 6] |= v << (b & 63);
      if ((b & 63) + bits > 64) w[(b >> 6) + 1] |= v >> (64 - (b & 63));
   }
   if (lastbit > 0 && word + 1 < n) w[word + 1] = (uint64_t)sa[word + 1];   // stale unpacked entry
   *bits_out = bits;
}

// bwt_build_opt (src/index_bwt.c:460-591): interleaved [mark(5)] {[word(5)] x mark_intv} ...,
// MSB-first bit planes, BWT[i] = text[SA[i]-1] (cyclic), separators set no bit; C[0] = wildcards.
void build_occ(const std::vector<uint8_t> &text, const std::vector<int64_t> &sa, int64_t wil_cnt,
               std::vector<uint64_t> *occ, std::vector<uint64_t> *c)
{
   const int64_t n = (int64_t)text.size();
   const int64_t mark_bits = MARK_INTV * 64;
   const int64_t n_intv = (n + mark_bits - 1) / mark_bits;
   const int64_t n_word = n_intv * MARK_INTV, n_mark = n_intv + 1;
   occ->assign((size_t)((n_word + n_mark) * SYM_COUNT), 0);
   uint64_t *o = occ->data();
   uint64_t abs_cnt[SYM_COUNT] = {0, 0, 0, 0, 0};
   for (int64_t w = 0; w < n_word; w++) {
      uint64_t plane[SYM_COUNT] = {0, 0, 0, 0, 0};
      const int64_t lo = w * 64, hi = std::min<int64_t>(lo + 64, n);
      for (int64_t i = lo; i < hi; i++) {
         int64_t p = sa[i];
         uint8_t s = text[p > 0 ? p - 1 : n - 1];
         if (s < SYM_COUNT) { plane[s] |= (uint64_t)1 << (63 - (i - lo)); abs_cnt[s]++; }
      }
      uint64_t *dst = o + (w + w / MARK_INTV + 1) * SYM_COUNT;
      for (int j = 0; j < SYM_COUNT; j++) dst[j] = plane[j];
      if ((w + 1) % MARK_INTV == 0) {
         // mark after a full interval holds the counts so far (final mark = totals)
         uint64_t *mk = o + ((w + 1) / MARK_INTV) * (MARK_INTV + 1) * SYM_COUNT;
         for (int j = 0; j < SYM_COUNT; j++) mk[j] = abs_cnt[j];
      }
   }
   c->assign(SYM_COUNT + 1, 0);
   (*c)[0] = (uint64_t)wil_cnt;
   for (int j = 1; j <= SYM_COUNT; j++) (*c)[j] = (*c)[j - 1] + abs_cnt[j - 1];
}

int write_index_files(const gpumap_index *ix, int64_t wil_cnt, int64_t sar_bits)
{
   const gpumap_index_view &v = ix->view;
   int fd = -1;
   uint64_t magic;
   // .sym (src/index_sym.c:369-429)
   fd = creat((ix->base + ".sym").c_str(), 0644);
   GM_FAIL_IF(fd < 0, "cannot create '.sym' file.");
   {
      magic = GM_MAGIC_SYM;
      uint8_t cnt = SYM_COUNT;
      GM_FAIL_IF(write_all(fd, &magic, 8) || write_all(fd, &cnt, 1) || write_all(fd, ix->canon, SYM_COUNT + 1) ||
                 write_all(fd, ix->sym_table, 256) || write_all(fd, ix->com_table, SYM_COUNT + 1), "write error.");
   }
   close(fd);
   // .txt (src/index_txt.c:577-661)
   fd = creat((ix->base + ".txt").c_str(), 0644);
   GM_FAIL_IF(fd < 0, "cannot create '.txt' file.");
   {
      int64_t hdr[5] = {(int64_t)GM_MAGIC_TXT, v.txt_len, v.seq_cnt, wil_cnt, 1};
      GM_FAIL_IF(write_all(fd, hdr, sizeof hdr) || write_all(fd, v.seq_len, 8 * v.seq_cnt) ||
                 write_all(fd, v.seq_beg, 8 * v.seq_cnt), "write error.");
      for (auto &nm : ix->own_names) GM_FAIL_IF(write_all(fd, nm.c_str(), nm.size() + 1), "write error.");
      GM_FAIL_IF(write_all(fd, v.text, (size_t)v.txt_len), "write error.");
   }
   close(fd);
   // .sar (src/index_sar.c:165-219)
   fd = creat((ix->base + ".sar").c_str(), 0644);
   GM_FAIL_IF(fd < 0, "cannot create '.sar' file.");
   {
      int64_t hdr[4] = {(int64_t)GM_MAGIC_SAR, sar_bits, v.sar_len, v.txt_len};
      GM_FAIL_IF(write_all(fd, hdr, sizeof hdr) || write_all(fd, v.sa, 8 * (size_t)v.sar_len), "write error.");
   }
   close(fd);
   // .bwt (src/index_bwt.c:702-766)
   fd = creat((ix->base + ".bwt").c_str(), 0644);
   GM_FAIL_IF(fd < 0, "cannot create '.bwt' file.");
   {
      int64_t hdr[5] = {(int64_t)GM_MAGIC_BWT, v.occ_length, v.occ_mark_intv, 64, v.occ_mark_intv * 64};
      GM_FAIL_IF(write_all(fd, hdr, sizeof hdr) || write_all(fd, v.c, 8 * (SYM_COUNT + 1)) ||
                 write_all(fd, v.occ, 8 * (size_t)v.occ_length), "write error.");
   }
   close(fd);
   return 0;
failure_return:
   if (fd >= 0) close(fd);
   return -1;
}

}  // namespace

// Builds the in-memory index from text codes that already hold F$...$ for every sequence
// (separators included), i.e. the state after the last txt_commit_seq.  Shared by the FASTA path
// and by gpumap_index_build_codes (synthetic genomes handed over as codes).
static int finish_build(gpumap_index *ix, bool write_files)
{
   int64_t wil_cnt = (int64_t)ix->own_seq_len.size();
   int64_t sar_bits = 0;
   int threads = (int)std::thread::hardware_concurrency();
   if (const char *e = getenv("GPUMAP_BUILD_THREADS")) threads = atoi(e);
   if (threads < 1) threads = 1;
   if (threads > 16) threads = 16;
   GM_FAIL_IF(ix->own_text.empty(), "text has length 0.");
   append_revcomp(ix, &wil_cnt);
   {
      const int64_t n = (int64_t)ix->own_text.size();
      std::vector<uint8_t> shifted((size_t)n);
      for (int64_t i = 0; i < n; i++) shifted[i] = (uint8_t)((ix->own_text[i] + 1) % (SYM_COUNT + 1));  // src/index_sar.c:48-51
      std::vector<int64_t> sa((size_t)n);
      GM_FAIL_IF(gm_suffix_sort(shifted.data(), n, sa.data(), threads) != 0, "suffix sort failed.");
      shifted.clear(); shifted.shrink_to_fit();
      build_occ(ix->own_text, sa, wil_cnt, &ix->own_occ, &ix->own_c);
      pack_suffix_array(sa, n, &sar_bits, &ix->own_sa);
   }
   {
      gpumap_index_view &v = ix->view;
      memset(&v, 0, sizeof v);
      v.occ = ix->own_occ.data(); v.occ_length = (int64_t)ix->own_occ.size();
      v.occ_mark_intv = MARK_INTV; v.occ_word_size = 64; v.c = ix->own_c.data();
      v.sa = ix->own_sa.data(); v.sar_bits = sar_bits; v.sar_len = (int64_t)ix->own_sa.size();
      v.txt_len = (int64_t)ix->own_text.size(); v.seq_cnt = (int64_t)ix->own_seq_len.size();
      v.seq_len = ix->own_seq_len.data(); v.seq_beg = ix->own_seq_beg.data();
      for (auto &nm : ix->own_names) ix->name_ptrs.push_back(nm.c_str());
      v.seq_name = ix->name_ptrs.data();
      v.wil_cnt = wil_cnt; v.rc_flag = 1; v.text = ix->own_text.data();
      v.sym_count = SYM_COUNT; v.sym_table = ix->sym_table; v.com_table = ix->com_table;
   }
   if (write_files) GM_FAIL_IF(write_index_files(ix, wil_cnt, sar_bits) != 0, "cannot write index files.");
   return 0;
failure_return:
   return -1;
}

extern "C" gpumap_index *gpumap_index_build(const char *ref_fasta, const char *filename_base)
{
   gpumap_index *ix = nullptr;
   GM_FAIL_IF(ref_fasta == nullptr, "argument 'ref_txt' is NULL.");
   GM_FAIL_IF(filename_base == nullptr, "argument 'filename_base' is NULL.");
   ix = new gpumap_index();
   ix->base = filename_base;
   fill_sym_tables(ix);
   GM_FAIL_IF(load_fasta(ref_fasta, ix) != 0, "cannot read FASTA file.");
   GM_FAIL_IF(finish_build(ix, true) != 0, "index construction failed.");
   return ix;
failure_return:
   gpumap_index_free(ix);
   return nullptr;
}

// Index of sequences handed over as symbol codes (0..4), one after another: seq_lens[i] letters
// each, named names[i].  filename_base may be NULL (nothing is written).  Used for synthetic
// genomes, which never exist as FASTA text.
extern "C" gpumap_index *gpumap_index_build_codes(const uint8_t *codes, const int64_t *seq_lens,
                                                   const char *const *names, int64_t seq_cnt,
                                                   const char *filename_base)
{
   gpumap_index *ix = nullptr;
   GM_FAIL_IF(codes == nullptr, "argument 'codes' is NULL.");
   GM_FAIL_IF(seq_lens == nullptr || names == nullptr || seq_cnt < 1, "invalid sequence list.");
   ix = new gpumap_index();
   ix->base = filename_base ? filename_base : "";
   fill_sym_tables(ix);
   {
      int64_t total = 0;
      for (int64_t i = 0; i < seq_cnt; i++) total += seq_lens[i] + 1;
      ix->own_text.reserve((size_t)(2 * total + 2));
      const uint8_t *p = codes;
      for (int64_t i = 0; i < seq_cnt; i++) {
         int64_t beg = (int64_t)ix->own_text.size();
         for (int64_t k = 0; k < seq_lens[i]; k++) ix->own_text.push_back(p[k] < SYM_COUNT ? p[k] : DEFAULT_SYM);
         p += seq_lens[i];
         ix->own_text.push_back(SYM_COUNT);
         ix->own_seq_beg.push_back(beg);
         ix->own_seq_len.push_back(seq_lens[i] + 1);
         GM_FAIL_IF(names[i] == nullptr || names[i][0] == 0, "argument 'seqname' is an empty String.");
         ix->own_names.push_back(names[i]);
      }
   }
   GM_FAIL_IF(finish_build(ix, filename_base != nullptr) != 0, "index construction failed.");
   return ix;
failure_return:
   gpumap_index_free(ix);
   return nullptr;
}

// Suffix array of a text given as symbol codes 0..5 (5 = separator), in the order sar_build gives
// (src/index_sar.c:46-54: symbols shifted by one so that separators sort first).  Exposed so that
// the reference's suffix-array known-answer tests (test/test_index_sar.c:8-126) can be run against
// the sorter alone.
extern "C" int gpumap_suffix_sort(const uint8_t *codes, int64_t n, int64_t *sa)
{
   GM_FAIL_IF(codes == nullptr || sa == nullptr || n < 1, "invalid arguments.");
   {
      std::vector<uint8_t> shifted((size_t)n);
      for (int64_t i = 0; i < n; i++) shifted[i] = (uint8_t)((codes[i] + 1) % (SYM_COUNT + 1));
      return gm_suffix_sort(shifted.data(), n, sa, 4);
   }
failure_return:
   return -1;
}

// txt_pos_to_str (src/index_txt.c:480-511).
extern "C" int gpumap_pos_to_str(const gpumap_index *ix, int64_t pos, char *buf, size_t bufsz)
{
   GM_FAIL_IF(ix == nullptr, "argument 'index' is NULL.");
   GM_FAIL_IF(pos < 0, "index must be positive.");
   GM_FAIL_IF(pos >= ix->view.txt_len, "index out of bounds.");
   {
      const gpumap_index_view &v = ix->view;
      int strand = 0;
      if (v.rc_flag) {
         strand = pos >= v.txt_len / 2;
         if (strand) pos = v.txt_len - 2 - pos;
      }
      int64_t lo = 0, hi = v.seq_cnt;                      // seq_bisect (:770-787)
      while (hi - lo >= 2) { int64_t mid = (lo + hi) / 2; if (pos < v.seq_beg[mid]) hi = mid; else lo = mid; }
      return snprintf(buf, bufsz, "%s:%ld:%c", v.seq_name[lo], (long)(pos - v.seq_beg[lo] + 1), strand ? '-' : '+');
   }
failure_return:
   return -1;
}

// ann_query (src/index_ann.c:278-347) on a table of gpumap_ann_build (= ann->info): distance to the closest
// neighbours, their number (as the reference decodes its classes, :332-337), the mismatch positions.
extern "C" int gpumap_ann_query(const uint8_t *table, int64_t size, int kmer, int64_t pos, int32_t *dist, int32_t *neigh_cnt,
                                int32_t *align_cnt, int32_t *align_pos)
{
   GM_FAIL_IF(table == nullptr, "argument 'ann' is NULL.");
   GM_FAIL_IF(pos < 0, "argument 'pos' must be positive.");
   GM_FAIL_IF(pos >= size * 2, "index 'pos' out of bounds.");
   {
      int strand = 0;
      if (pos > size) { strand = 1; pos = (size - 1) * 2 - pos; }          // :305-308
      GM_FAIL_IF(pos < 0 || pos >= size, "index 'pos' out of bounds.");    // (the reference reads outside its table here)
      const uint8_t info = table[pos];
      int32_t n_aln = 0;
      if ((info >> 6) & 1)
         for (int i = 0; i < kmer && pos + i < size; i++)
            if ((table[pos + i] >> 7) & 1) { if (align_pos) align_pos[n_aln] = strand ? kmer - 1 - i : i; n_aln++; }
      const int cnt = info & 0x0F;
      if (dist) *dist = cnt ? ((info >> 4) & 3) + 1 : 0;
      if (align_cnt) *align_cnt = n_aln;
      if (neigh_cnt) *neigh_cnt = cnt <= 10 ? cnt : cnt == 0x0B ? 15 : cnt == 0x0C ? 40 : cnt == 0x0D ? 75 : cnt == 0x0E ? 300 : 1000;
   }
   return 0;
failure_return:
   return -1;
}

// ann_file_write (src/index_ann.c:384-434): magic, kmer, tau, size, the table; the file index_read looks for is
// <base>.ann.<kmer>.<tau> (src/index.c:68-83, :201).
extern "C" int gpumap_ann_file_write(const char *filename, const uint8_t *table, int64_t size, int kmer, int tau)
{
   FILE *f = nullptr;
   GM_FAIL_IF(filename == nullptr, "argument 'filename' is NULL.");
   GM_FAIL_IF(table == nullptr, "argument 'ann' is NULL.");
   f = fopen(filename, "wb");
   GM_FAIL_IF(f == nullptr, "cannot create the file.");
   {
      const int64_t head[4] = {(int64_t)0x0fcb0fcb00050001ll, kmer, tau, size};      // ANN_FILE_MAGICNO, src/index_ann.h:22
      GM_FAIL_IF(fwrite(head, 8, 4, f) != 4 || (size > 0 && fwrite(table, 1, (size_t)size, f) != (size_t)size), "write error.");
   }
   fclose(f);
   return 0;
failure_return:
   if (f) fclose(f);
   return -1;
}
