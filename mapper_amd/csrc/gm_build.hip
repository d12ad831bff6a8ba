// gm_build.hip -- index construction on the device.
//
// The reference builds its index on the host: txt_commit_rc (src/index_txt.c:286-343), sar_build =
// divsufsort + compact_array (src/index_sar.c:19-70, :283-328) and bwt_build_opt
// (src/index_bwt.c:460-591); hours for a GRCh38-size text.  Here the same arrays are produced in
// HBM, bit for bit (tests compare them with the host builder, whose files are byte-identical to the
// reference's), in seconds:
//
//   k_build_text        text = F $ revcomp(F) $                         (txt_commit_rc)
//   suffix_sort_device  prefix doubling on rocPRIM radix sorts          (sar_build: any correct
//                       suffix sorter gives the same array, the order is src/index_sar.c:46-54)
//   k_bwt_planes, k_intv_counts, k_write_marks   the reference's Occ array   (bwt_build_opt)
//   k_pack_sa           bit-packed suffix array                         (compact_array)
//
// and then repacked into rank blocks by the same kernels gpumap_create uses.  One-off work, off the
// query hot path; bandwidth-bound streaming and sorting, nothing here is tuned beyond that.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include "gm_devctx.hpp"

using namespace gm;
typedef unsigned long long u64;

namespace {

constexpr int K0 = 21;                 // symbols in the initial key, 3 bits each (shifted symbol + 1; 0 = past the end)
constexpr int64_t MARK_INTV = 14;      // BWT_OCC_MARK_INTV_DEF, src/index_bwt.h:26

__device__ __forceinline__ int shifted(uint8_t s) { return s == SYMS ? 0 : s + 1; }   // src/index_sar.c:48-51

// ---------------------------------------------------------------------------------------------- text
__global__ void k_build_text(const uint8_t *codes, const int64_t *seq_beg, int64_t seq_cnt, int64_t tlen0, uint8_t *text)
{
   const int64_t n = 2 * tlen0;
   for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < n; q += (int64_t)gridDim.x * blockDim.x) {
      int64_t p = q < tlen0 ? q : 2 * (tlen0 - 1) - q;      // reverse-complement half mirrors around tlen0 - 1
      uint8_t s;
      if (q == n - 1) s = SYMS;
      else {
         int64_t lo = 0, hi = seq_cnt;                      // last sequence starting at or before p
         while (hi - lo >= 2) { int64_t mid = (lo + hi) / 2; if (p < seq_beg[mid]) hi = mid; else lo = mid; }
         int64_t end = (lo + 1 < seq_cnt ? seq_beg[lo + 1] : tlen0) - 1;    // its separator
         if (p == end) s = SYMS;
         else { s = codes[p - lo]; if (s >= SYMS) s = UNKNOWN; if (q >= tlen0) s = (uint8_t)complement(s); }
      }
      text[q] = s;
   }
}

// ---------------------------------------------------------------------------------------------- suffix sort
__device__ __forceinline__ u64 key_of(const uint8_t *text, int64_t n, int64_t i, int syms)
{
   u64 k = 0;
   for (int t = 0; t < syms; t++) {
      int64_t p = i + t;
      u64 v = p < n ? (u64)(shifted(text[p]) + 1) : 0;
      k |= v << (60 - 3 * t);
   }
   return k;
}

__global__ void k_class_hist(const uint8_t *text, int64_t n, int P, u64 *hist /* 8^P */)
{
   __shared__ unsigned int h[64];
   if (threadIdx.x < 64) h[threadIdx.x] = 0;
   __syncthreads();
   for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
      atomicAdd(&h[key_of(text, n, i, P) >> (63 - 3 * P)], 1u);
   __syncthreads();
   if (threadIdx.x < 64 && h[threadIdx.x]) atomicAdd(hist + threadIdx.x, (u64)h[threadIdx.x]);
}

// Members of class `cls` (leading P symbols) with their 21-symbol keys, in no particular order.
__global__ void __launch_bounds__(1024) k_class_gather(const uint8_t *text, int64_t n, int P, int cls, u64 *keys, u64 *idx, u64 *cursor)
{
   __shared__ unsigned int wcnt[16];
   __shared__ u64 base;
   const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
   const int64_t span = (int64_t)gridDim.x * blockDim.x;
   for (int64_t i0 = blockIdx.x * (int64_t)blockDim.x; i0 < n; i0 += span) {
      int64_t i = i0 + threadIdx.x;
      bool in = i < n && (int)(key_of(text, n, i, P) >> (63 - 3 * P)) == cls;
      u64 m = __ballot(in);
      if (lane == 0) wcnt[wid] = (unsigned int)__popcll(m);
      __syncthreads();
      if (threadIdx.x == 0) {
         unsigned int tot = 0;
         for (int w = 0; w < 16; w++) { unsigned int c = wcnt[w]; wcnt[w] = tot; tot += c; }
         base = tot ? atomicAdd(cursor, (u64)tot) : 0;
      }
      __syncthreads();
      if (in) {
         u64 at = base + wcnt[wid] + (u64)__popcll(m & ((1ull << lane) - 1ull));
         keys[at] = key_of(text, n, i, K0);
         idx[at] = (u64)i;
      }
      __syncthreads();
   }
}

// hs[j] = j where a run of equal keys starts, else 0 (a running maximum then gives each j its run start)
__global__ void k_run_heads(const u64 *keys, int64_t m, u64 *hs)
{
   for (int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; j < m; j += (int64_t)gridDim.x * blockDim.x)
      hs[j] = (j == 0 || keys[j] != keys[j - 1]) ? (u64)j : 0;
}

// Segment [off, off + m) of the suffix array after the initial sort: SA, rank (= SA index of the
// first member of the run) and the unresolved flag (run longer than one).
__global__ void k_segment_apply(const u64 *idx, const u64 *rs, int64_t m, int64_t off, u64 *sa, u64 *rk, uint8_t *unres)
{
   for (int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; j < m; j += (int64_t)gridDim.x * blockDim.x) {
      u64 r = rs[j];
      bool head = r == (u64)j, next_head = j + 1 >= m || rs[j + 1] == (u64)(j + 1);
      sa[off + j] = idx[j];
      rk[idx[j]] = (u64)off + r;
      unres[off + j] = !(head && next_head);
   }
}

// Stable compaction of the flagged positions: out = in[x] (or x itself when in == nullptr).
// (A launch holds fewer than 2^32 threads, texts more than 2^32 symbols: chunks of 1024 are looped over.)
__global__ void __launch_bounds__(1024) k_flag_count(const uint8_t *flag, int64_t n, int64_t n_chunks, u64 *cnt)
{
   for (int64_t ch = blockIdx.x; ch < n_chunks; ch += gridDim.x) {
      int64_t x = ch * 1024 + threadIdx.x;
      int c = __syncthreads_count(x < n && flag[x]);
      if (threadIdx.x == 0) cnt[ch] = (u64)c;
   }
}

__global__ void __launch_bounds__(1024) k_flag_write(const uint8_t *flag, int64_t n, int64_t n_chunks, const u64 *off, const u64 *in, u64 *out)
{
   __shared__ unsigned int wcnt[16];
   const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
   for (int64_t ch = blockIdx.x; ch < n_chunks; ch += gridDim.x) {
      int64_t x = ch * 1024 + threadIdx.x;
      bool f = x < n && flag[x];
      u64 m = __ballot(f);
      if (lane == 0) wcnt[wid] = (unsigned int)__popcll(m);
      __syncthreads();
      if (f) {
         unsigned int before = 0;
         for (int w = 0; w < wid; w++) before += wcnt[w];
         out[off[ch] + before + (u64)__popcll(m & ((1ull << lane) - 1ull))] = in ? in[x] : (u64)x;
      }
      __syncthreads();
   }
}

// One doubling round over the unresolved entries.  pos[x] = their slots in the suffix array, increasing.
__global__ void k_round_keys(const u64 *pos, int64_t U, const u64 *sa, const u64 *rk, int64_t n, int64_t h,
                             u64 *sfx, u64 *k2, u64 *ghead)
{
   for (int64_t x = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; x < U; x += (int64_t)gridDim.x * blockDim.x) {
      u64 p = pos[x], s = sa[p];
      sfx[x] = s;
      k2[x] = (int64_t)s + h < n ? rk[s + h] + 1 : 0;
      ghead[x] = rk[s] == p ? 1 : 0;         // the rank of a group is the slot of its first member
   }
}

__global__ void k_compose_keys(const u64 *ord_incl, const u64 *k2, int64_t U, int kb, u64 *key)
{
   for (int64_t x = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; x < U; x += (int64_t)gridDim.x * blockDim.x)
      key[x] = ((ord_incl[x] - 1) << kb) | k2[x];
}

__global__ void k_iota(u64 *v, int64_t U)
{
   for (int64_t x = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; x < U; x += (int64_t)gridDim.x * blockDim.x) v[x] = (u64)x;
}

__global__ void k_gather_u64(const u64 *src, const u64 *perm, int64_t U, u64 *dst)
{
   for (int64_t x = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; x < U; x += (int64_t)gridDim.x * blockDim.x) dst[x] = src[perm[x]];
}

// two-pass form: heads from (ord, k2) seen through the permutation
__global__ void k_run_heads2(const u64 *ord_incl, const u64 *k2, const u64 *perm, int64_t U, u64 *hs)
{
   for (int64_t y = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; y < U; y += (int64_t)gridDim.x * blockDim.x) {
      bool head = y == 0 || ord_incl[perm[y]] != ord_incl[perm[y - 1]] || k2[perm[y]] != k2[perm[y - 1]];
      hs[y] = head ? (u64)y : 0;
   }
}

__global__ void k_round_apply(const u64 *pos, const u64 *sfx_sorted, const u64 *rs, int64_t U, u64 *sa, u64 *rk, uint8_t *unres)
{
   for (int64_t y = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; y < U; y += (int64_t)gridDim.x * blockDim.x) {
      u64 r = rs[y];
      bool head = r == (u64)y, next_head = y + 1 >= U || rs[y + 1] == (u64)(y + 1);
      sa[pos[y]] = sfx_sorted[y];
      rk[sfx_sorted[y]] = pos[r];
      unres[y] = !(head && next_head);
   }
}

// ---------------------------------------------------------------------------------------------- Occ, reference layout
// One wave per 64-position word: BWT[i] = text[SA[i] - 1] (cyclic), five MSB-first bit planes at
// (w + w / 14 + 1) * 5 (src/index_bwt.c:525-564).  Separators and the padding past n set no bit.
__global__ void __launch_bounds__(256) k_bwt_planes(const uint8_t *text, const u64 *sa, int64_t n, int64_t n_word, u64 *occ)
{
   const int lane = threadIdx.x & 63;
   const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6, n_wave = ((int64_t)gridDim.x * blockDim.x) >> 6;
   for (int64_t w = wave; w < n_word; w += n_wave) {
      int64_t i = w * 64 + lane;
      int s = SYMS;
      if (i < n) { u64 p = sa[i]; s = text[p ? p - 1 : (u64)n - 1]; }
      u64 plane = 0;
      for (int j = 0; j < SYMS; j++) {
         u64 m = __ballot(s == j);
         if (lane == j) plane = m;
      }
      if (lane < SYMS) occ[(w + w / MARK_INTV + 1) * SYMS + lane] = __brevll(plane);
   }
}

__global__ void k_intv_counts(const u64 *occ, int64_t n_intv, u64 *cnt /* [5][n_intv] */)
{
   int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
   if (t >= n_intv * SYMS) return;
   int64_t k = t / SYMS; int j = (int)(t - k * SYMS);
   u64 c = 0;
   for (int64_t w = k * MARK_INTV; w < (k + 1) * MARK_INTV; w++) c += (u64)__popcll(occ[(w + w / MARK_INTV + 1) * SYMS + j]);
   cnt[(int64_t)j * n_intv + k] = c;
}

// mark k + 1 = counts in BWT[0, 896 (k + 1)); mark 0 stays zero (:558-564)
__global__ void k_write_marks(const u64 *cum /* [5][n_intv] inclusive */, int64_t n_intv, u64 *occ)
{
   int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
   if (t >= n_intv * SYMS) return;
   int64_t k = t / SYMS; int j = (int)(t - k * SYMS);
   occ[(k + 1) * (MARK_INTV + 1) * SYMS + j] = cum[(int64_t)j * n_intv + k];
}

// ---------------------------------------------------------------------------------------------- packed suffix array
// compact_array (src/index_sar.c:283-328): entry i in bits [i b, (i + 1) b) of the stream, LSB first.
__global__ void k_pack_sa(const u64 *sa, int64_t n, int bits, int64_t words, u64 *out)
{
   for (int64_t w = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; w < words; w += (int64_t)gridDim.x * blockDim.x) {
      u64 lo = (u64)w * 64, v = 0;
      for (int64_t e = (int64_t)(lo / (u64)bits); e < n && (u64)e * (u64)bits < lo + 64; e++) {
         u64 b = (u64)e * (u64)bits;
         v |= b >= lo ? sa[e] << (b - lo) : sa[e] >> (lo - b);
      }
      out[w] = v;
   }
}

// Debug aid (GPUMAP_BUILD_VERIFY=samples): neighbours in the suffix array must be in suffix order.
__global__ void k_verify_order(const uint8_t *text, const u64 *sa, int64_t n, int64_t samples, u64 *bad)
{
   int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
   if (t >= samples) return;
   u64 z = (u64)t * 0x9E3779B97F4A7C15ull + 0x1234567ull;
   z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
   int64_t i = (int64_t)(z % (u64)(n - 1));
   int64_t a = (int64_t)sa[i], b = (int64_t)sa[i + 1];
   if (a < 0 || a >= n || b < 0 || b >= n || a == b) { atomicAdd(bad, 1ull); return; }
   for (;;) {
      if (a >= n) return;                       // the shorter suffix sorts first
      if (b >= n) { atomicAdd(bad, 1ull); return; }
      int x = shifted(text[a]), y = shifted(text[b]);
      if (x < y) return;
      if (x > y) { atomicAdd(bad, 1ull); return; }
      a++; b++;
   }
}

struct Arena {       // device allocations of the build, released together
   std::vector<void *> p;
   template <class T> hipError_t get(T **out, size_t count)
   {
      void *q = nullptr;
      hipError_t e = hipMalloc(&q, std::max<size_t>(count, 1) * sizeof(T));
      if (e == hipSuccess) { p.push_back(q); *out = (T *)q; }
      return e;
   }
   void drop(void *q)
   {
      for (auto &x : p) if (x == q) { (void)hipFree(q); x = nullptr; }
   }
   ~Arena() { for (void *q : p) if (q) (void)hipFree(q); }
};

inline unsigned grid_for(int64_t items, int block) { return (unsigned)std::min<int64_t>(cdiv(std::max<int64_t>(items, 1), block), 1 << 20); }
inline int bits_for(u64 v) { int b = 0; while (b < 64 && (v >> b)) b++; return b; }

struct MaxOp { __host__ __device__ u64 operator()(u64 a, u64 b) const { return a > b ? a : b; } };

// Running maximum (in place allowed).
hipError_t scan_max(u64 *in, u64 *out, int64_t m, DevBuf &tmp, hipStream_t st)
{
   size_t bytes = 0;
   hipError_t e = rocprim::inclusive_scan(nullptr, bytes, in, out, (size_t)m, MaxOp(), st);
   if (e != hipSuccess) return e;
   if ((e = tmp.ensure(bytes)) != hipSuccess) return e;
   return rocprim::inclusive_scan(tmp.p, bytes, in, out, (size_t)m, MaxOp(), st);
}
hipError_t scan_sum(u64 *in, u64 *out, int64_t m, bool inclusive, DevBuf &tmp, hipStream_t st)
{
   size_t bytes = 0;
   hipError_t e = inclusive ? rocprim::inclusive_scan(nullptr, bytes, in, out, (size_t)m, rocprim::plus<u64>(), st)
                            : rocprim::exclusive_scan(nullptr, bytes, in, out, (u64)0, (size_t)m, rocprim::plus<u64>(), st);
   if (e != hipSuccess) return e;
   if ((e = tmp.ensure(bytes)) != hipSuccess) return e;
   return inclusive ? rocprim::inclusive_scan(tmp.p, bytes, in, out, (size_t)m, rocprim::plus<u64>(), st)
                    : rocprim::exclusive_scan(tmp.p, bytes, in, out, (u64)0, (size_t)m, rocprim::plus<u64>(), st);
}

// Sorts (keys, vals) by bits [0, end_bit); the sorted arrays end up in *keys / *vals (buffers may swap).
hipError_t sort_pairs(u64 **keys, u64 **keys_alt, u64 **vals, u64 **vals_alt, int64_t m, int end_bit, DevBuf &tmp, hipStream_t st)
{
   if (m >= ((int64_t)1 << 31)) return hipErrorInvalidValue;
   if (end_bit < 1) end_bit = 1;
   rocprim::double_buffer<u64> dk(*keys, *keys_alt), dv(*vals, *vals_alt);
   size_t bytes = 0;
   hipError_t e = rocprim::radix_sort_pairs(nullptr, bytes, dk, dv, (size_t)m, 0, (unsigned)end_bit, st);
   if (e != hipSuccess) return e;
   if ((e = tmp.ensure(bytes)) != hipSuccess) return e;
   e = rocprim::radix_sort_pairs(tmp.p, bytes, dk, dv, (size_t)m, 0, (unsigned)end_bit, st);
   *keys = dk.current(); *keys_alt = dk.alternate(); *vals = dv.current(); *vals_alt = dv.alternate();
   return e;
}

// Positions x (or in[x]) with flag[x] != 0, in increasing x.
int compact(const uint8_t *flag, int64_t n, const u64 *in, u64 **out, int64_t *count, Arena &A, DevBuf &tmp, hipStream_t st)
{
   const int64_t nb = cdiv(std::max<int64_t>(n, 1), 1024);
   u64 *cnt = nullptr, *off = nullptr;
   u64 last[2] = {0, 0};
   *out = nullptr; *count = 0;
   HIP_TRY(A.get(&cnt, (size_t)nb));
   HIP_TRY(A.get(&off, (size_t)nb));
   hipLaunchKernelGGL(k_flag_count, dim3(grid_for(nb, 1)), dim3(1024), 0, st, flag, n, nb, cnt);
   HIP_TRY(hipGetLastError());
   HIP_TRY(scan_sum(cnt, off, nb, false, tmp, st));
   HIP_TRY(hipMemcpyAsync(&last[0], off + nb - 1, 8, hipMemcpyDeviceToHost, st));
   HIP_TRY(hipMemcpyAsync(&last[1], cnt + nb - 1, 8, hipMemcpyDeviceToHost, st));
   HIP_TRY(hipStreamSynchronize(st));
   *count = (int64_t)(last[0] + last[1]);
   HIP_TRY(A.get(out, (size_t)*count));
   if (*count > 0) hipLaunchKernelGGL(k_flag_write, dim3(grid_for(nb, 1)), dim3(1024), 0, st, flag, n, nb, (const u64 *)off, in, *out);
   HIP_TRY(hipGetLastError());
   HIP_TRY(hipStreamSynchronize(st));
   A.drop(cnt); A.drop(off);
   return 0;
failure_return:
   return -1;
}

// Suffix array of text[0, n) (codes 0..5, separator smallest) into sa[n].
int suffix_sort_device(const uint8_t *text, int64_t n, u64 *sa, hipStream_t st, bool verbose)
{
   Arena A;
   DevBuf tmp;
   u64 *rk = nullptr, *hist = nullptr, *cursor = nullptr, *pos = nullptr;
   uint8_t *unres = nullptr;
   int64_t U = 0;
   const int P = getenv("GPUMAP_BUILD_P") ? std::min(2, std::max(1, atoi(getenv("GPUMAP_BUILD_P")))) : (n >= ((int64_t)1 << 30) ? 2 : 1);
   const int n_class = 1 << (3 * P);
   const bool force_two_pass = getenv("GPUMAP_BUILD_TWO_PASS") != nullptr;
   u64 h_hist[64];
   HIP_TRY(A.get(&rk, (size_t)n));
   HIP_TRY(A.get(&unres, (size_t)n));
   HIP_TRY(A.get(&hist, 64));
   HIP_TRY(A.get(&cursor, 1));
   HIP_TRY(hipMemsetAsync(hist, 0, 64 * 8, st));
   hipLaunchKernelGGL(k_class_hist, dim3(grid_for(n, 1024)), dim3(1024), 0, st, text, n, P, hist);
   HIP_TRY(hipMemcpyAsync(h_hist, hist, 8 * n_class, hipMemcpyDeviceToHost, st));
   HIP_TRY(hipStreamSynchronize(st));
   {
      // ---- initial order: class by class (leading P symbols), radix sort on the 21-symbol key
      int64_t biggest = 0;
      for (int c = 0; c < n_class; c++) biggest = std::max<int64_t>(biggest, (int64_t)h_hist[c]);
      GM_FAIL_IF(biggest >= ((int64_t)1 << 31), "a class of suffixes is too large for one device sort.");
      u64 *k_a = nullptr, *k_b = nullptr, *v_a = nullptr, *v_b = nullptr;
      HIP_TRY(A.get(&k_a, (size_t)biggest)); HIP_TRY(A.get(&k_b, (size_t)biggest));
      HIP_TRY(A.get(&v_a, (size_t)biggest)); HIP_TRY(A.get(&v_b, (size_t)biggest));
      int64_t off = 0;
      for (int c = 0; c < n_class; c++) {
         const int64_t m = (int64_t)h_hist[c];
         if (m == 0) continue;
         HIP_TRY(hipMemsetAsync(cursor, 0, 8, st));
         hipLaunchKernelGGL(k_class_gather, dim3(grid_for(n, 1024)), dim3(1024), 0, st, text, n, P, c, k_a, v_a, cursor);
         HIP_TRY(hipGetLastError());
         u64 *keys = k_a, *keys_alt = k_b, *vals = v_a, *vals_alt = v_b;
         HIP_TRY(sort_pairs(&keys, &keys_alt, &vals, &vals_alt, m, 63 - 3 * P, tmp, st));
         hipLaunchKernelGGL(k_run_heads, dim3(grid_for(m, 256)), dim3(256), 0, st, (const u64 *)keys, m, keys_alt);
         HIP_TRY(scan_max(keys_alt, keys_alt, m, tmp, st));
         hipLaunchKernelGGL(k_segment_apply, dim3(grid_for(m, 256)), dim3(256), 0, st, (const u64 *)vals, (const u64 *)keys_alt, m, off, sa, rk, unres);
         HIP_TRY(hipGetLastError());
         off += m;
      }
      HIP_TRY(hipStreamSynchronize(st));
      GM_FAIL_IF(off != n, "internal error: classes do not cover the text.");
      A.drop(k_a); A.drop(k_b); A.drop(v_a); A.drop(v_b);
   }
   GM_FAIL_IF(compact(unres, n, nullptr, &pos, &U, A, tmp, st) != 0, "compaction failed.");
   A.drop(unres); unres = nullptr;
   // ---- doubling rounds over the entries still tied
   for (int64_t h = K0; U > 0; h *= 2) {
      if (verbose) fprintf(stderr, "[gpumap build] h = %ld: %ld suffixes tied\n", (long)h, (long)U);
      GM_FAIL_IF(h > 2 * n, "internal error: suffix sort does not converge.");
      u64 *sfx = nullptr, *k2 = nullptr, *ord = nullptr, *key = nullptr, *alt_k = nullptr, *alt_v = nullptr, *hs = nullptr, *npos = nullptr;
      uint8_t *flag = nullptr;
      int64_t U2 = 0;
      u64 n_groups = 0;
      HIP_TRY(A.get(&sfx, (size_t)U)); HIP_TRY(A.get(&k2, (size_t)U)); HIP_TRY(A.get(&ord, (size_t)U));
      HIP_TRY(A.get(&key, (size_t)U)); HIP_TRY(A.get(&alt_k, (size_t)U)); HIP_TRY(A.get(&alt_v, (size_t)U));
      HIP_TRY(A.get(&flag, (size_t)U));
      hipLaunchKernelGGL(k_round_keys, dim3(grid_for(U, 256)), dim3(256), 0, st, (const u64 *)pos, U, (const u64 *)sa, (const u64 *)rk, n, h, sfx, k2, ord);
      HIP_TRY(scan_sum(ord, ord, U, true, tmp, st));            // ord[x] = 1-based ordinal of x's group
      HIP_TRY(hipMemcpyAsync(&n_groups, ord + U - 1, 8, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
      const int kb = bits_for((u64)n), gb = bits_for(n_groups);
      u64 *sorted_sfx = nullptr;
      if (gb + kb <= 64 && !force_two_pass) {
         hipLaunchKernelGGL(k_compose_keys, dim3(grid_for(U, 256)), dim3(256), 0, st, (const u64 *)ord, (const u64 *)k2, U, kb, key);
         u64 *keys = key, *keys_alt = alt_k, *vals = sfx, *vals_alt = alt_v;
         HIP_TRY(sort_pairs(&keys, &keys_alt, &vals, &vals_alt, U, gb + kb, tmp, st));
         hs = keys_alt;
         hipLaunchKernelGGL(k_run_heads, dim3(grid_for(U, 256)), dim3(256), 0, st, (const u64 *)keys, U, hs);
         sorted_sfx = vals;
      } else {
         // least significant first: by k2, then (stable) by group
         u64 *perm = nullptr, *perm_alt = alt_v, *kk = nullptr, *kk_alt = alt_k;
         HIP_TRY(A.get(&perm, (size_t)U)); HIP_TRY(A.get(&kk, (size_t)U));
         hipLaunchKernelGGL(k_iota, dim3(grid_for(U, 256)), dim3(256), 0, st, perm, U);
         HIP_TRY(hipMemcpyAsync(kk, k2, 8 * (size_t)U, hipMemcpyDeviceToDevice, st));
         HIP_TRY(sort_pairs(&kk, &kk_alt, &perm, &perm_alt, U, kb, tmp, st));
         hipLaunchKernelGGL(k_gather_u64, dim3(grid_for(U, 256)), dim3(256), 0, st, (const u64 *)ord, (const u64 *)perm, U, kk);
         HIP_TRY(sort_pairs(&kk, &kk_alt, &perm, &perm_alt, U, gb, tmp, st));
         hs = kk_alt;
         hipLaunchKernelGGL(k_run_heads2, dim3(grid_for(U, 256)), dim3(256), 0, st, (const u64 *)ord, (const u64 *)k2, (const u64 *)perm, U, hs);
         hipLaunchKernelGGL(k_gather_u64, dim3(grid_for(U, 256)), dim3(256), 0, st, (const u64 *)sfx, (const u64 *)perm, U, key);
         sorted_sfx = key;
      }
      HIP_TRY(scan_max(hs, hs, U, tmp, st));
      hipLaunchKernelGGL(k_round_apply, dim3(grid_for(U, 256)), dim3(256), 0, st, (const u64 *)pos, (const u64 *)sorted_sfx, (const u64 *)hs, U, sa, rk, flag);
      HIP_TRY(hipGetLastError());
      GM_FAIL_IF(compact(flag, U, pos, &npos, &U2, A, tmp, st) != 0, "compaction failed.");
      for (void *q : A.p) if (q && q != rk && q != npos && q != hist && q != cursor) { (void)hipFree(q); }
      for (auto &q : A.p) if (q && q != rk && q != npos && q != hist && q != cursor) q = nullptr;
      pos = npos; U = U2;
   }
   tmp.release();
   return 0;
failure_return:
   tmp.release();
   return -1;
}

}  // namespace

extern "C" gpumap_ctx *gpumap_build_device(const uint8_t *codes, int codes_on_device, const int64_t *seq_lens,
                                           int64_t seq_cnt, int device, int keep_reference)
{
   gpumap_ctx *ctx = nullptr;
   Arena A;
   DevBuf tmp;
   int count = 0;
   const bool verbose = getenv("GPUMAP_BUILD_VERBOSE") != nullptr;
   GM_FAIL_IF(codes == nullptr || seq_lens == nullptr || seq_cnt < 1, "invalid arguments.");
   HIP_TRY(hipGetDeviceCount(&count));
   GM_FAIL_IF(device < 0 || device >= count, "no such HIP device.");
   {
      std::vector<int64_t> seq_beg((size_t)seq_cnt);
      int64_t tlen0 = 0, total_codes = 0;
      for (int64_t i = 0; i < seq_cnt; i++) {
         GM_FAIL_IF(seq_lens[i] < 0, "negative sequence length.");
         seq_beg[i] = tlen0; tlen0 += seq_lens[i] + 1; total_codes += seq_lens[i];
      }
      const int64_t n = 2 * tlen0;
      GM_FAIL_IF(num_supers(n) > MAX_SB_LDS, "text too long for the staged superblock table.");
      ctx = new gpumap_ctx();
      {  // sym tables of sym_new_dna (src/index_sym.c:11-13, :121-132)
         memset(ctx->sym_table, UNKNOWN, 256);
         const char *al[5] = {"Aa", "Cc", "Gg", "Tt", "Nn"};
         for (int i = 0; i < 5; i++) for (const char *c = al[i]; *c; c++) ctx->sym_table[(uint8_t)*c] = (uint8_t)i;
         const uint8_t com[8] = {3, 2, 1, 0, 4, 5, 0, 0};
         memcpy(ctx->com_table, com, 8);
      }
      GM_FAIL_IF(gm_ctx_common_init(ctx, device) != 0, "cannot initialise the device context.");
      hipStream_t st = ctx->stream;
      // ---- text
      const uint8_t *d_codes = codes;
      uint8_t *own_codes = nullptr, *text = nullptr;
      int64_t *d_seq_beg = nullptr;
      if (!codes_on_device) {
         HIP_TRY(A.get(&own_codes, (size_t)total_codes));
         HIP_TRY(hipMemcpyAsync(own_codes, codes, (size_t)total_codes, hipMemcpyHostToDevice, st));      // (on the kernels' stream: see gm_ctx_common_init)
         d_codes = own_codes;
      }
      HIP_TRY(A.get(&d_seq_beg, (size_t)seq_cnt));
      HIP_TRY(hipMemcpyAsync(d_seq_beg, seq_beg.data(), 8 * (size_t)seq_cnt, hipMemcpyHostToDevice, st));
      HIP_TRY(hipMalloc((void **)&text, (size_t)n + 16));                    // k_verify reads whole words
      HIP_TRY(hipMemsetAsync(text + n, SYMS, 16, st));
      ctx->d_text = text;                                   // released by gpumap_destroy (or below)
      hipLaunchKernelGGL(k_build_text, dim3(grid_for(n, 256)), dim3(256), 0, st, d_codes, (const int64_t *)d_seq_beg, seq_cnt, tlen0, text);
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipStreamSynchronize(st));
      if (own_codes) A.drop(own_codes);
      // ---- suffix array
      u64 *sa = nullptr;
      HIP_TRY(A.get(&sa, (size_t)n));
      GM_FAIL_IF(suffix_sort_device(text, n, sa, st, verbose) != 0, "suffix sort failed.");
      if (const char *e = getenv("GPUMAP_BUILD_VERIFY")) {
         const int64_t samples = std::max<int64_t>(1, atoll(e));
         u64 *bad = nullptr, h_bad = 0;
         HIP_TRY(A.get(&bad, 1));
         HIP_TRY(hipMemsetAsync(bad, 0, 8, st));
         hipLaunchKernelGGL(k_verify_order, dim3((unsigned)cdiv(samples, 256)), dim3(256), 0, st, (const uint8_t *)text, (const u64 *)sa, n, samples, bad);
         HIP_TRY(hipMemcpyAsync(&h_bad, bad, 8, hipMemcpyDeviceToHost, st));
         HIP_TRY(hipStreamSynchronize(st));
         fprintf(stderr, "[gpumap build] verify: %llu of %ld sampled neighbours out of order\n", h_bad, (long)samples);
         GM_FAIL_IF(h_bad != 0, "suffix array is not sorted.");
      }
      // ---- Occ array in the reference's layout, C[]
      const int64_t n_intv = cdiv(n, MARK_INTV * 64), n_word = n_intv * MARK_INTV;
      const int64_t occ_words = (n_word + n_intv + 1) * SYMS;
      u64 *ref_occ = nullptr, *cnt = nullptr, *d_c = nullptr;
      HIP_TRY(hipMalloc((void **)&ref_occ, 8 * (size_t)occ_words));
      ctx->d_ref_occ = ref_occ; ctx->ref_occ_words = occ_words;
      HIP_TRY(hipMemsetAsync(ref_occ, 0, 8 * (size_t)occ_words, st));
      HIP_TRY(A.get(&cnt, (size_t)(n_intv * SYMS)));
      HIP_TRY(A.get(&d_c, 8));
      hipLaunchKernelGGL(k_bwt_planes, dim3(grid_for(n_word * 64, 256)), dim3(256), 0, st, (const uint8_t *)text, (const u64 *)sa, n, n_word, ref_occ);
      hipLaunchKernelGGL(k_intv_counts, dim3((unsigned)cdiv(n_intv * SYMS, 256)), dim3(256), 0, st, (const u64 *)ref_occ, n_intv, cnt);
      HIP_TRY(hipGetLastError());
      for (int j = 0; j < SYMS; j++) HIP_TRY(scan_sum(cnt + j * n_intv, cnt + j * n_intv, n_intv, true, tmp, st));
      hipLaunchKernelGGL(k_write_marks, dim3((unsigned)cdiv(n_intv * SYMS, 256)), dim3(256), 0, st, (const u64 *)cnt, n_intv, ref_occ);
      HIP_TRY(hipGetLastError());
      {
         u64 tot[SYMS];
         for (int j = 0; j < SYMS; j++) HIP_TRY(hipMemcpyAsync(&tot[j], cnt + (j + 1) * n_intv - 1, 8, hipMemcpyDeviceToHost, st));
         HIP_TRY(hipStreamSynchronize(st));
         ctx->wil_cnt = 2 * seq_cnt;                                       // src/index_txt.c:314-334
         ctx->c_host[0] = (u64)ctx->wil_cnt;                               // src/index_bwt.c:575-582
         for (int j = 1; j <= SYMS; j++) ctx->c_host[j] = ctx->c_host[j - 1] + tot[j - 1];
         GM_FAIL_IF((int64_t)ctx->c_host[SYMS] != n, "internal error: symbol counts do not add up to the text length.");
         HIP_TRY(hipMemcpyAsync(d_c, ctx->c_host, 8 * (SYMS + 1), hipMemcpyHostToDevice, st));
      }
      // ---- packed suffix array (with the stale word the in-place packer leaves behind, see gm_index.cpp)
      {
         int bits = 0;
         while ((u64)n > ((u64)1 << bits)) bits++;
         const u64 total_bits = (u64)n * (u64)bits;
         const int64_t word = (int64_t)(total_bits / 64);
         const int lastbit = (int)(total_bits % 64);
         const int64_t words = word + 1 + (lastbit > 0 ? 1 : 0);
         ctx->ix.n = n; ctx->ix.sar_bits = bits; ctx->sar_len = words;
         ctx->sa_bytes = (words + 1) * 8;
         HIP_TRY(hipMalloc(&ctx->d_sa, (size_t)ctx->sa_bytes));
         HIP_TRY(hipMemsetAsync(ctx->d_sa, 0, (size_t)ctx->sa_bytes, st));
         hipLaunchKernelGGL(k_pack_sa, dim3(grid_for(words, 256)), dim3(256), 0, st, (const u64 *)sa, n, bits, words, (u64 *)ctx->d_sa);
         HIP_TRY(hipGetLastError());
         if (lastbit > 0 && word + 1 < n)
            HIP_TRY(hipMemcpyAsync((u64 *)ctx->d_sa + word + 1, sa + word + 1, 8, hipMemcpyDeviceToDevice, st));
         HIP_TRY(hipStreamSynchronize(st));
      }
      A.drop(sa);
      // ---- rank blocks
      GM_FAIL_IF(gm_ctx_repack(ctx, (const uint64_t *)ref_occ, (const uint64_t *)d_c, MARK_INTV, seq_beg.data(), seq_cnt) != 0, "cannot repack the Occ table.");
      if (!keep_reference) {
         HIP_TRY(hipFree(ctx->d_ref_occ)); ctx->d_ref_occ = nullptr; ctx->ref_occ_words = 0;      // the text stays: k_verify reads it
      }
   }
   tmp.release();
   return ctx;
failure_return:
   tmp.release();
   gpumap_destroy(ctx);
   return nullptr;
}

extern "C" int gpumap_reference_sizes(const gpumap_ctx *ctx, int64_t sizes[6])
{
   GM_FAIL_IF(ctx == nullptr || sizes == nullptr, "NULL argument.");
   sizes[0] = ctx->ix.n; sizes[1] = ctx->ref_occ_words; sizes[2] = ctx->sar_len; sizes[3] = ctx->ix.sar_bits;
   sizes[4] = ctx->wil_cnt; sizes[5] = ctx->d_text ? ctx->ix.n : 0;
   return 0;
failure_return:
   return -1;
}

extern "C" int gpumap_reference_fetch(gpumap_ctx *ctx, uint64_t *occ, uint64_t c[6], int64_t *sa, uint8_t *text)
{
   GM_FAIL_IF(ctx == nullptr, "argument 'ctx' is NULL.");
   HIP_TRY(hipSetDevice(ctx->device));
   if (occ) {
      GM_FAIL_IF(ctx->d_ref_occ == nullptr, "the context does not keep the reference Occ array.");
      HIP_TRY(hipMemcpyAsync(occ, ctx->d_ref_occ, 8 * (size_t)ctx->ref_occ_words, hipMemcpyDeviceToHost, ctx->stream));
   }
   if (c) memcpy(c, ctx->c_host, 8 * (SYMS + 1));
   if (sa) HIP_TRY(hipMemcpyAsync(sa, ctx->d_sa, 8 * (size_t)ctx->sar_len, hipMemcpyDeviceToHost, ctx->stream));
   if (text) {
      GM_FAIL_IF(ctx->d_text == nullptr, "the context does not keep the text.");
      HIP_TRY(hipMemcpyAsync(text, ctx->d_text, (size_t)ctx->ix.n, hipMemcpyDeviceToHost, ctx->stream));
   }
   HIP_TRY(hipStreamSynchronize(ctx->stream));
   return 0;
failure_return:
   return -1;
}
