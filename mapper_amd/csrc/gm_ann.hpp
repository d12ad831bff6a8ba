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
// depth = letters of the prefixes the reference cuts its jobs at (:116-121: 1 for threads = 1, the smallest d with
// 5^d >= 5 * threads): a job is the interval of one prefix of `depth` symbols with at most tau N (jobs_by_prefix,
// :612-661); suffixes with a separator or more than tau N among their first `depth` symbols lie in no job.
__global__ void __launch_bounds__(256) k_ann_classify(IndexDev ix, const uint8_t *text, int kmer, int tau, int depth, uint8_t *cls, int64_t first, int64_t last)
{
   const int64_t i = first + blockIdx.x * (int64_t)256 + threadIdx.x;
   if (i >= last) return;
   const int64_t p = sa_entry(ix.sa, ix.sar_bits, i);
   uint8_t c = ANN_CLS_NONE;
   bool in_job = p + depth <= ix.n;
   {
      int n_head = 0;
      for (int t = 0; in_job && t < depth; t++) { const int s = text[p + t]; in_job = s < SYMS; n_head += s == UNKNOWN; }
      in_job = in_job && n_head <= tau;                   // :629
   }
   if (in_job) {                                          // the jobs cover the suffixes that start with one of the five symbols (:116-132, threads = 1)
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

// ------------------------------------------------------------------------------------------------
// The reference's reuse of the last k-mer's search (`trail`: src/index_ann.c:552-570, src/blocksearch.c:41-43, :63-66,
// :73-86, :156-157).  ann_build_mt keeps, per job and strand class, the last k-mer it searched (query_1 / query_2) and a
// tree of hit stacks; blocksc_trail recomputes tree->next_l->stack (the hits of the left block, filtered by the bound)
// only when the new k-mer shares fewer than pos_r letters with the last one, and blocksearch_trail_rec recomputes the
// leaf tree->next_l->next_l->stack only when it shares fewer than the leaf's length.  A k-mer whose search returns early
// (:63-66) replaces the last k-mer but leaves the tree as it is, and a k-mer with N runs its left block with a smaller
// budget: the next k-mer then extends hits that were made for other letters or another budget.  Where no k-mer of the
// text holds an N none of this shows (every reuse is of hits that a fresh search would find again); where some do, the
// reference's table is what this plan reproduces: per k-mer of a chunk, in suffix-array order,
//    prev   the last k-mer of its class before it                                     (inclusive scan, shifted)
//    trail  the letters it shares with prev (0 across a job's border: the state is per job)
//    wl     it recomputes the left block's hits (searched and trail < pos_r); wll: it recomputes the first leaf
//    ol     the last k-mer of its class up to itself with wl; oll: the same with wll   (inclusive scans)
// and the search of k-mer i is run as the reference's state has it: letters [0, pos_r) and the left block's budget and
// bound of ol, the first leaf's letters of oll where that leaf is searched on its own, the extension's letters and
// budget of i.  Steps are counted where the reference executes them (not in blocks it took from the tree).  The state
// that crosses from one chunk of k-mers to the next is AnnCarry.
// ------------------------------------------------------------------------------------------------
struct AnnCarry {
   int64_t prev[2];          // per strand class: index (in the list of all k-mers) of the last k-mer of the class, -1 = none
   int64_t ol[2];            // of the k-mer whose left block's hits the tree holds
   int64_t ol_bound[2];      // its bound (:76) and
   int64_t oll[2];           // of the k-mer whose first leaf's hits the tree holds
   int32_t ol_tl[2];         // the budget of its left block (:70)
};

struct AnnPlanArgs {
   IndexDev ix;
   const uint8_t *text;
   const int64_t *q_sa;      // the suffix-array entries of ALL k-mers searched, in order
   int64_t c0, nq;           // the chunk: k-mers c0 .. c0 + nq - 1 are queries 0 .. nq - 1 of the batch
   int kmer, tau, depth, stride, n_nat;
   const int64_t *q_fp, *q_rp;
   uint8_t *reads;           // the batch's arrays, with room for 2 * nq queries
   uint16_t *read_len, *read_row;
   int64_t *sc_bound;
   uint8_t *score_add, *sc_flags;
   uint32_t *alias;
   unsigned long long *s1, *s2a, *s2b;
   AnnCarry *carry;
   unsigned int *n_extra;
};

struct MaxHalves {
   __host__ __device__ unsigned long long operator()(unsigned long long a, unsigned long long b) const
   {
      const unsigned long long lo = (a & 0xffffffffull) > (b & 0xffffffffull) ? (a & 0xffffffffull) : (b & 0xffffffffull);
      const unsigned long long hi = (a >> 32) > (b >> 32) ? (a >> 32) : (b >> 32);
      return (hi << 32) | lo;
   }
};

__device__ __forceinline__ uint32_t ann_half(unsigned long long v, int cls) { return (uint32_t)(cls ? v >> 32 : v & 0xffffffffull); }
__device__ __forceinline__ unsigned long long ann_in_half(uint32_t v, int cls) { return cls ? (unsigned long long)v << 32 : (unsigned long long)v; }
__device__ __forceinline__ int ann_lcp(const uint8_t *a, const uint8_t *b, int k) { int t = 0; while (t < k && a[t] == b[t]) t++; return t; }
__device__ __forceinline__ const uint8_t *ann_letters(const AnnPlanArgs &P, int64_t g) { return P.text + sa_entry(P.ix.sa, P.ix.sar_bits, P.q_sa[g]); }
// the budget of the left block of a searched query from its row of the phase tables (k_sc_prepare)
__device__ __forceinline__ int ann_row_tl(int row, int tau) { return sc_tau_l((row >> 1) % (tau + 1), row & 1); }

__global__ void __launch_bounds__(256) k_ann_plan_class(AnnPlanArgs P)
{
   const int64_t r = blockIdx.x * (int64_t)256 + threadIdx.x;
   if (r >= P.nq) return;
   P.s1[r] = ann_in_half((uint32_t)r + 1, P.q_fp[r] >= P.q_rp[r]);
}

__global__ void __launch_bounds__(256) k_ann_plan_trail(AnnPlanArgs P)
{
   const int64_t r = blockIdx.x * (int64_t)256 + threadIdx.x;
   if (r >= P.nq) return;
   const int cls = P.q_fp[r] >= P.q_rp[r];
   const int row = P.read_row[r];
   const uint32_t s = r > 0 ? ann_half(P.s1[r - 1], cls) : 0u;
   const int64_t prev = s ? P.c0 + s - 1 : P.carry->prev[cls];
   int trail = 0;
   if (prev >= 0) {
      trail = ann_lcp(ann_letters(P, prev), ann_letters(P, P.c0 + r), P.kmer);       // :786-789
      if (trail < P.depth) trail = 0;                                                 // another job: fresh queries (:528-529)
   }
   const int pos_r = sc_pos_r(P.kmer, cls), leaf = (pos_r >> 1) + (pos_r & 1);
   const bool wl = row != NO_ROW && trail < pos_r;                                    // src/blocksearch.c:73
   const bool wll = wl && ann_row_tl(row, P.tau) == 1 && trail < leaf;                // :156
   P.s2a[r] = wl ? ann_in_half((uint32_t)r + 1, cls) : 0ull;
   P.s2b[r] = wll ? ann_in_half((uint32_t)r + 1, cls) : 0ull;
}

__global__ void __launch_bounds__(256) k_ann_plan_apply(AnnPlanArgs P)
{
   const int64_t r = blockIdx.x * (int64_t)256 + threadIdx.x;
   if (r >= P.nq) return;
   P.alias[r] = (uint32_t)r;
   P.sc_flags[r] = 3;                                     // both leaves (those the row has), every step counts
   const int row = P.read_row[r];
   if (row == NO_ROW) return;
   const int cls = P.q_fp[r] >= P.q_rp[r];
   const int k = P.kmer, pos_r = sc_pos_r(k, cls), leaf = (pos_r >> 1) + (pos_r & 1);
   const int64_t g = P.c0 + r;
   const uint8_t *qi = ann_letters(P, g);
   const uint32_t xl = ann_half(P.s2a[r], cls), xll = ann_half(P.s2b[r], cls);
   int64_t ol = xl ? P.c0 + xl - 1 : P.carry->ol[cls];
   int64_t oll = xll ? P.c0 + xll - 1 : P.carry->oll[cls];
   const uint8_t *qo = nullptr, *qoo = nullptr;
   if (ol >= 0) { qo = ann_letters(P, ol); if (ann_lcp(qo, qi, P.depth) < P.depth) ol = -1; }       // of another job
   if (ol < 0) { P.read_row[r] = NO_ROW; return; }                                     // the tree's left stack is empty: no hits
   if (oll >= 0) { qoo = ann_letters(P, oll); if (ann_lcp(qoo, qi, P.depth) < P.depth) oll = -1; }
   const int64_t o_bound = xl ? P.sc_bound[xl - 1] : P.carry->ol_bound[cls];
   const int o_tl = xl ? ann_row_tl(P.read_row[xl - 1], P.tau) : P.carry->ol_tl[cls];
   const bool self = ol == g;
   const uint32_t scp = 3u * (uint32_t)o_tl + 1u;          // the bounded extension's phase (build_sc_phase_table)
   uint32_t leaves = 1, count_from = self ? 0u : scp;
   bool split = false;
   if (o_tl == 1) {
      leaves = 3;
      if (oll != ol) {                                     // the left block's search took its first leaf's hits from the tree
         if (self) count_from = 1;
         if (oll < 0) leaves = 2;                          // (none there)
         else if (ann_lcp(qoo, qo, leaf) < leaf) split = true;                         // hits of other letters
      }
   }
   int new_row = row;
   if (!self) {
      for (int t = 0; t < pos_r; t++) P.reads[r * (int64_t)P.stride + t] = qo[t];
      new_row = P.n_nat + 2 * row + o_tl;
      P.read_row[r] = (uint16_t)new_row;
      P.sc_bound[r] = o_bound;
   }
   if (!split) { P.sc_flags[r] = (uint8_t)(leaves | (count_from << 2)); return; }
   // two queries: the second leaf and what follows it as above; the first leaf, with the letters it was searched with, apart
   P.sc_flags[r] = (uint8_t)(2u | ((self ? 0u : scp) << 2));
   const int64_t q = P.nq + (int64_t)atomicAdd(P.n_extra, 1u);
   uint8_t *dst = P.reads + q * (int64_t)P.stride;
   for (int t = 0; t < P.stride; t++) dst[t] = t < leaf ? qoo[t] : (t < pos_r ? qo[t] : (t < k ? qi[t] : (uint8_t)0xff));
   P.read_len[q] = (uint16_t)k;
   P.read_row[q] = (uint16_t)new_row;
   P.sc_bound[q] = o_bound;
   P.score_add[q] = P.score_add[r];
   P.sc_flags[q] = (uint8_t)(1u | ((self ? 1u : scp) << 2));
   P.alias[q] = (uint32_t)r;
}

__global__ void k_ann_plan_carry(AnnPlanArgs P)
{
   if (P.nq < 1) return;
   for (int cls = 0; cls < 2; cls++) {
      const uint32_t s = ann_half(P.s1[P.nq - 1], cls), x = ann_half(P.s2a[P.nq - 1], cls), y = ann_half(P.s2b[P.nq - 1], cls);
      if (s) P.carry->prev[cls] = P.c0 + s - 1;
      if (x) {
         P.carry->ol[cls] = P.c0 + x - 1;
         P.carry->ol_bound[cls] = P.sc_bound[x - 1];
         P.carry->ol_tl[cls] = ann_row_tl(P.read_row[x - 1], P.tau);
      }
      if (y) P.carry->oll[cls] = P.c0 + y - 1;
   }
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
