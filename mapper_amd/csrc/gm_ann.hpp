// gm_ann.hpp -- kernels of gpumap_ann_build: the neighbour annotation of the reference's index (ann_build,
// src/index_ann.c:56-257) on the device.  The heavy part, one blocksc_trail search per distinct k-mer of the text
// (ann_build_mt, :500-609), is k_search<true> (gm_search.hpp) over chunks of k-mers; what is here is what stands
// around it:
//   k_ann_classify   the walk of ann_build_mt / neigh_next (:552-572, :732-830) over the suffix array, all positions
//                    at once: which entries start the interval of a k-mer that is searched, which are flagged
//                    ANN_NO_INFO, which are never visited
//   k_ann_gather     the letters of a chunk of k-mers, as a batch of queries
//   k_ann_updates    hits_push (:920-1071) as data: every hit of a k-mer is an update of its neighbour's record, every
//                    k-mer with hits an update of its own; the reference applies them one at a time under a mutex, here
//                    they are sorted by record and folded (k_ann_fold): a record keeps the smallest distance seen, the
//                    number of updates at that distance (saturating) and the union of their mismatch positions, 0xFF
//                    once there are more than three (aln_merge, :833-879).  The fold does not depend on the order
//   k_ann_state / k_ann_push   the compression (:200-241): a record covers its own suffix-array entry and the untouched
//                    entries behind it; neigh_push (:664-729) for each of them, as atomic ORs into the table
#pragma once
#include "gm_kernels.hpp"

namespace gm {

constexpr uint8_t ANN_CLS_NONE  = 0;     // not the start of a visited interval (or a suffix that starts with a separator: never visited)
constexpr uint8_t ANN_CLS_FLAG  = 1;     // visited and flagged ANN_NO_INFO (src/index_ann.c:559-561)
constexpr uint8_t ANN_CLS_QUERY = 2;     // the start of the interval of a k-mer that is searched
constexpr uint8_t ANN_CLS_EMPTY = 4;     // (ORed in later) ANN_NO_INFO unless the entry holds a record (:943-947, :950-957)
constexpr int ANN_ALN = 3;               // mismatch positions a record keeps: max(3, tau) with tau < 4 (:78, :86)

// A record: count (16 bits) | distance << 16 | three positions << 24 (1-based, 0 = none, all 0xFF = too many).
__device__ __forceinline__ uint32_t ann_rec_cnt(uint64_t r) { return (uint32_t)(r & 0xffff); }
__device__ __forceinline__ uint32_t ann_rec_tau(uint64_t r) { return (uint32_t)((r >> 16) & 0xff); }
__device__ __forceinline__ uint32_t ann_rec_aln(uint64_t r) { return (uint32_t)((r >> 24) & 0xffffff); }
__device__ __forceinline__ uint64_t ann_rec_make(uint32_t cnt, uint32_t tau, uint32_t aln)
{
   return (uint64_t)cnt | ((uint64_t)tau << 16) | ((uint64_t)aln << 24);
}

// The union of two sorted lists of up to three positions; all 0xFF when either is or when it has more than three
// members (aln_merge; a list of 0xFF merges to 0xFF there too: its entries do not fit).
__device__ __forceinline__ uint32_t ann_aln_union(uint32_t a, uint32_t b)
{
   if (a == 0xffffffu || b == 0xffffffu) return 0xffffffu;
   uint32_t out[2 * ANN_ALN];
   int k = 0, i = 0, j = 0;
   uint32_t av[ANN_ALN] = {a & 0xff, (a >> 8) & 0xff, (a >> 16) & 0xff}, bv[ANN_ALN] = {b & 0xff, (b >> 8) & 0xff, (b >> 16) & 0xff};
   while (i < ANN_ALN && av[i] && j < ANN_ALN && bv[j]) {
      if (av[i] == bv[j]) { out[k++] = av[i]; i++; j++; }
      else if (av[i] < bv[j]) out[k++] = av[i++];
      else out[k++] = bv[j++];
   }
   while (i < ANN_ALN && av[i]) out[k++] = av[i++];
   while (j < ANN_ALN && bv[j]) out[k++] = bv[j++];
   if (k > ANN_ALN) return 0xffffffu;
   uint32_t r = 0;
   for (int t = 0; t < k; t++) r |= out[t] << (8 * t);
   return r;
}

// aln_positions (:882-917): the set bits of the first nbits bits, 1-based, from the other end when `reverse`.
__device__ __forceinline__ uint32_t ann_aln_positions(const uint64_t bits[4], int nbits, bool reverse)
{
   uint32_t r = 0;
   int a = 0;
   for (int i = 0; i < nbits; i++)
      if ((bits[i >> 6] >> (i & 63)) & 1) {
         if (a == ANN_ALN) return 0xffffffu;
         r |= (uint32_t)(reverse ? nbits - i : i + 1) << (8 * a);
         a++;
      }
   return r;
}

// The fold of two updates of one record (both with count >= 1).
__device__ __forceinline__ uint64_t ann_fold(uint64_t x, uint64_t y)
{
   const uint32_t tx = ann_rec_tau(x), ty = ann_rec_tau(y);
   if (tx < ty) return x;
   if (ty < tx) return y;
   const uint32_t c = ann_rec_cnt(x) + ann_rec_cnt(y);
   return ann_rec_make(c < 0xfffeu ? c : 0xfffeu, tx, ann_aln_union(ann_rec_aln(x), ann_rec_aln(y)));
}

// One thread per suffix-array entry, entries first .. last - 1 (a launch of 2^32 threads and more is cut short by the
// runtime without an error: the host launches pieces).
__global__ void __launch_bounds__(256) k_ann_classify(IndexDev ix, const uint8_t *text, int kmer, int tau, uint8_t *cls, int64_t first, int64_t last)
{
   const int64_t i = first + blockIdx.x * (int64_t)256 + threadIdx.x;
   if (i >= last) return;
   const int64_t p = sa_entry(ix.sa, ix.sar_bits, i);
   uint8_t c = ANN_CLS_NONE;
   if (text[p] < SYMS) {                                  // the jobs cover the suffixes that start with one of the five symbols (:116-132, threads = 1)
      bool valid = p + kmer <= ix.n;                      // :765
      int n_cnt = 0;
      for (int t = 0; valid && t < kmer; t++) {           // :773-777
         const int s = text[p + t];
         valid = s < SYMS;
         n_cnt += s == UNKNOWN;
      }
      if (!valid) c = ANN_CLS_FLAG;
      else {
         bool start = i == 0;
         if (!start) {
            const int64_t q = sa_entry(ix.sa, ix.sar_bits, i - 1);
            start = q + kmer > ix.n;
            for (int t = 0; !start && t < kmer; t++) start = text[q + t] != text[p + t];
         }
         if (start) c = n_cnt > tau ? ANN_CLS_FLAG : ANN_CLS_QUERY;      // :807
      }
   }
   cls[i] = c;
}

// The k-mers of a chunk as rows of codes (the batch's layout is made by k_encode from these and their offsets).  One
// thread per four bytes of the rows: neighbouring threads read the same suffix-array entry and the same window of the
// text (cache hits) and the rows are written as whole words.  (A thread per k-mer with byte stores at a stride of kmer
// took 1.7 ms per 4 M k-mers, a seventh of gpumap_ann_build.)  `codes` holds a multiple of four bytes.
__global__ void __launch_bounds__(256) k_ann_gather(IndexDev ix, const uint8_t *text, const int64_t *q_sa, int64_t n_q, int kmer,
                                                    uint8_t *codes, int64_t *offsets)
{
   const int64_t t = blockIdx.x * (int64_t)256 + threadIdx.x;
   if (t <= n_q) offsets[t] = t * kmer;
   const int64_t total = n_q * kmer, b0 = 4 * t;
   if (b0 >= total) return;
   uint32_t word = 0;
   int64_t q = b0 / kmer;
   int off = (int)(b0 - q * kmer);
   int64_t p = sa_entry(ix.sa, ix.sar_bits, q_sa[q]);
   for (int k = 0; k < 4 && b0 + k < total; k++) {
      word |= (uint32_t)text[p + off] << (8 * k);
      if (++off == kmer && b0 + k + 1 < total) { off = 0; q++; p = sa_entry(ix.sa, ix.sar_bits, q_sa[q]); }
   }
   reinterpret_cast<uint32_t *>(codes)[t] = word;
}

// hits_push as updates.  One thread per k-mer of the chunk: its hits are hit_*[lo .. hi) (sorted by k-mer).
//   out_key[h] / out_val[h], h in lo..hi: the update of the neighbour of hit h (key = record << 8 | distance; ~0: none, the self hit)
//   out_key[n_hits + q] / out_val: the update of the k-mer's own record (distance = the smallest among its hits)
//   cls: ANN_CLS_EMPTY ORed in (32-bit atomics on the bytes' words) where the reference writes ANN_NO_INFO
__global__ void __launch_bounds__(256) k_ann_updates(int64_t n_q, int kmer, int tau, const int64_t *q_fp, const int64_t *q_rp,
                                                     const int64_t *hit_read, const int64_t *hit_fp, const int64_t *hit_rp,
                                                     const int64_t *hit_score, const uint64_t *hit_align, int64_t n_hits,
                                                     unsigned long long *out_key, unsigned long long *out_val, uint32_t *cls_words)
{
   const int64_t q = blockIdx.x * (int64_t)256 + threadIdx.x;
   if (q >= n_q) return;
   // the hits of k-mer q: lower bounds of q and q + 1 in hit_read
   int64_t lo = 0, hi = n_hits;
   { int64_t a = 0, b = n_hits; while (a < b) { const int64_t m = (a + b) >> 1; if (hit_read[m] < q) a = m + 1; else b = m; } lo = a; }
   { int64_t a = lo, b = n_hits; while (a < b) { const int64_t m = (a + b) >> 1; if (hit_read[m] <= q) a = m + 1; else b = m; } hi = a; }
   const int64_t fp = q_fp[q], rp = q_rp[q];
   const int64_t low = fp < rp ? fp : rp;
   if (fp > rp) atomicOr(cls_words + (fp >> 2), (uint32_t)ANN_CLS_EMPTY << (8 * (fp & 3)));        // :943-947
   unsigned long long own_key = ~0ull, own_val = 0;
   if (hi - lo < 2) {                                                                               // :950-957
      atomicOr(cls_words + (low >> 2), (uint32_t)ANN_CLS_EMPTY << (8 * (low & 3)));
      for (int64_t h = lo; h < hi; h++) out_key[h] = ~0ull;
   } else {
      int best = tau + 1, nh = 0;
      uint64_t qal[4] = {0, 0, 0, 0};
      for (int64_t h = lo; h < hi; h++) {
         if (hit_fp[h] == fp) { out_key[h] = ~0ull; continue; }                                     // the self hit (:969)
         const int64_t pf = hit_fp[h], pr = hit_rp[h];
         const bool nrev = pr < pf;
         const int s = (int)hit_score[h];
         uint64_t al[4] = {hit_align[4 * h], hit_align[4 * h + 1], hit_align[4 * h + 2], hit_align[4 * h + 3]};
         out_key[h] = ((unsigned long long)(nrev ? pr : pf) << 8) | (unsigned long long)s;
         out_val[h] = ann_rec_make(1, (uint32_t)s, ann_aln_positions(al, kmer, nrev));
         if (s == best) { for (int w = 0; w < 4; w++) qal[w] |= al[w]; nh++; }                      // :1014-1026
         else if (s < best) { for (int w = 0; w < 4; w++) qal[w] = al[w]; best = s; nh = 1; }
      }
      if (nh > 0) {                                                                                 // :1030-1061
         own_key = ((unsigned long long)low << 8) | (unsigned long long)best;
         own_val = ann_rec_make((uint32_t)(nh < 0xfffe ? nh : 0xfffe), (uint32_t)best, ann_aln_positions(qal, kmer, rp < fp));
      }
   }
   out_key[n_hits + q] = own_key;
   out_val[n_hits + q] = own_val;
}

// The sorted updates folded into the records: the thread of the first update of a record folds the updates of that
// record at the smallest distance (they come first) and what the record holds already.
__global__ void __launch_bounds__(256) k_ann_fold(const unsigned long long *key, const unsigned long long *val, int64_t n, uint64_t *rec)
{
   const int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x;
   if (i >= n) return;
   const unsigned long long k = key[i];
   if (k == ~0ull) return;
   if (i > 0 && (key[i - 1] >> 8) == (k >> 8)) return;
   uint64_t acc = val[i];
   for (int64_t j = i + 1; j < n && key[j] == k; j++) acc = ann_fold(acc, val[j]);
   const uint64_t old = rec[k >> 8];
   rec[k >> 8] = ann_rec_cnt(old) ? ann_fold(old, acc) : acc;
}

// What the compression sees at entry i (:209, :217): a record, ANN_NO_INFO, or nothing; owner candidate = i unless nothing.
__global__ void __launch_bounds__(256) k_ann_state(const uint64_t *rec, const uint8_t *cls, int64_t first, int64_t last, int64_t *owner)
{
   const int64_t i = first + blockIdx.x * (int64_t)256 + threadIdx.x;
   if (i >= last) return;
   const bool something = ann_rec_cnt(rec[i]) != 0 || (cls[i] & (ANN_CLS_FLAG | ANN_CLS_EMPTY)) != 0;
   owner[i] = something ? i : -1;
}

// The running maximum carried from one piece of the scan into the next.
__global__ void k_ann_carry(int64_t *owner_piece)
{
   if (owner_piece[-1] > owner_piece[0]) owner_piece[0] = owner_piece[-1];
}

__device__ __forceinline__ void ann_or_byte(uint32_t *table_words, int64_t size, int64_t at, uint32_t v)
{
   if (at >= 0 && at < size) atomicOr(table_words + (at >> 2), v << (8 * (at & 3)));       // (outside: the reference writes out of bounds there)
}

// neigh_push for every suffix-array entry that a record covers (owner = inclusive running maximum of k_ann_state's).
__global__ void __launch_bounds__(256) k_ann_push(IndexDev ix, const uint64_t *rec, const int64_t *owner, int kmer, uint32_t *table_words,
                                                  int64_t first, int64_t last)
{
   const int64_t i = first + blockIdx.x * (int64_t)256 + threadIdx.x;
   if (i >= last) return;
   const int64_t o = owner[i];
   if (o < 0) return;
   const uint64_t r = rec[o];
   const uint32_t cnt = ann_rec_cnt(r);
   if (cnt == 0) return;                                   // ANN_NO_INFO ends the run of the record before it
   const int64_t half = ix.n / 2;
   const int64_t p = sa_entry(ix.sa, ix.sar_bits, i);
   const bool rev = p >= half;
   const int64_t at = rev ? ix.n - p - kmer : p;           // :228-232
   uint32_t v = cnt <= 10 ? cnt : cnt <= 20 ? 0x0Bu : cnt <= 50 ? 0x0Cu : cnt <= 100 ? 0x0Du : cnt <= 500 ? 0x0Eu : 0x0Fu;   // :692-697
   v |= ((ann_rec_tau(r) - 1) & 3u) << 4;                  // :700
   const uint32_t aln = ann_rec_aln(r);
   const bool have = (aln & 0xff) != 0xff;                 // :703
   if (have) v |= 1u << 6;
   ann_or_byte(table_words, half, at, v);
   if (have)
      for (int t = 0; t < ANN_ALN; t++) {
         uint32_t a = (aln >> (8 * t)) & 0xff;
         if (!a) break;
         if (rev) a = (uint32_t)kmer + 1 - a;              // :711-713
         ann_or_byte(table_words, half, at + a - 1, 0x80u);
      }
}

}  // namespace gm
