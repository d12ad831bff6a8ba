// sort_bench.hip -- the ordering stage's alternatives on keys shaped like a chunk's (read << 37 | position << 4 | score):
//   full      one radix sort over bits [5, 61)            (7 passes: what the library did in round 2)
//   seg       radix sort over the top 16 bits of the read (2 passes), then hipcub's segmented sort of the buckets
//   byread    radix sort over the read's 24 bits (3 passes): the hits of a read together, not ordered
//   lds       radix sort over the top 16 bits (2 passes), then the library's k_order_segments (a workgroup per bucket, in LDS)
// tools/_build/sort_bench N_KEYS N_READS [heavy | real]
//   heavy: a quarter of the hits on a thousand reads;  real: one hit per read, the rest of the hits on 0.8 % of the reads
//   (65 .. 1024 each: the histogram of a GRCh38-size chunk, profiles/r3_hits_hist.jsonl)
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include <algorithm>
#include <cstring>
#include "../mapper_amd/csrc/gm_kernels.hpp"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ void k_bounds(const unsigned long long *keys, long n, int shift, int n_buckets, int *off)
{
   const int b = blockIdx.x * blockDim.x + threadIdx.x;
   if (b > n_buckets) return;
   long lo = 0, hi = n;                                   // first key with (key >> shift) >= b
   while (lo < hi) { const long m = (lo + hi) >> 1; if ((long)(keys[m] >> shift) < b) lo = m + 1; else hi = m; }
   off[b] = (int)lo;
}
int main(int argc, char **argv)
{
   const long n = argc > 1 ? atol(argv[1]) : 41700000, R = argc > 2 ? atol(argv[2]) : 12500000;
   const bool heavy = argc > 3 && !strcmp(argv[3], "heavy"), real = argc > 3 && !strcmp(argv[3], "real");
   std::mt19937_64 rng(7);
   std::vector<unsigned long long> k((size_t)n);
   if (real) {
      long i = 0;
      for (; i < R && i < n; i++) k[(size_t)i] = ((unsigned long long)i << 37) | ((rng() & ((1ull << 33) - 1)) << 4) | (rng() % 3);
      while (i < n) {
         const unsigned long long read = rng() % R;
         long c = 65 + (long)(rng() % 460);                // mean 294
         for (; c > 0 && i < n; c--, i++) k[(size_t)i] = (read << 37) | ((rng() & ((1ull << 33) - 1)) << 4) | (rng() % 3);
      }
      std::shuffle(k.begin(), k.end(), rng);
   } else
   for (long i = 0; i < n; i++) {
      unsigned long long read = heavy && (rng() % 4 == 0) ? (rng() % 1000) * 12345 % R : rng() % R;      // heavy: a quarter of the hits on a thousand reads
      k[(size_t)i] = (read << 37) | ((rng() & ((1ull << 33) - 1)) << 4) | (rng() % 3);
   }
   unsigned long long *d0, *d1, *src;
   CK(hipMalloc(&d0, 8 * (size_t)n)); CK(hipMalloc(&d1, 8 * (size_t)n)); CK(hipMalloc(&src, 8 * (size_t)n));
   CK(hipMemcpy(src, k.data(), 8 * (size_t)n, hipMemcpyHostToDevice));
   void *tmp = nullptr; size_t tmp_bytes = 0;
   int *off; CK(hipMalloc(&off, 4 * 70000));
   hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
   std::vector<unsigned long long> want = k, got((size_t)n);
   std::sort(want.begin(), want.end());
   uint32_t *seg_first; CK(hipMalloc(&seg_first, 4 * 70000));
   for (const char *mode : {"full", "seg", "byread", "lds"}) {
      for (int rep = 0; rep < 3; rep++) {
         CK(hipMemcpy(d0, src, 8 * (size_t)n, hipMemcpyDeviceToDevice));
         hipcub::DoubleBuffer<unsigned long long> db(d0, d1);
         size_t need = 0;
         CK(hipEventRecord(e0, 0));
         if (!strcmp(mode, "full")) {
            CK(hipcub::DeviceRadixSort::SortKeys(nullptr, need, db, (int)n, 5, 61, 0));
            if (need > tmp_bytes) { if (tmp) CK(hipFree(tmp)); CK(hipMalloc(&tmp, need)); tmp_bytes = need; }
            CK(hipcub::DeviceRadixSort::SortKeys(tmp, need, db, (int)n, 5, 61, 0));
         } else if (!strcmp(mode, "byread")) {
            CK(hipcub::DeviceRadixSort::SortKeys(nullptr, need, db, (int)n, 37, 61, 0));
            if (need > tmp_bytes) { if (tmp) CK(hipFree(tmp)); CK(hipMalloc(&tmp, need)); tmp_bytes = need; }
            CK(hipcub::DeviceRadixSort::SortKeys(tmp, need, db, (int)n, 37, 61, 0));
         } else if (!strcmp(mode, "lds")) {
            CK(hipcub::DeviceRadixSort::SortKeys(nullptr, need, db, (int)n, 45, 61, 0));
            if (need > tmp_bytes) { if (tmp) CK(hipFree(tmp)); CK(hipMalloc(&tmp, need)); tmp_bytes = need; }
            CK(hipcub::DeviceRadixSort::SortKeys(tmp, need, db, (int)n, 45, 61, 0));
            const long nb = ((R - 1) >> 8) + 1;
            hipLaunchKernelGGL(gm::k_seg_first, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, (const unsigned long long *)db.Current(), (int64_t)n, 45, (int64_t)nb, seg_first);
            hipLaunchKernelGGL((gm::k_order_segments<4>), dim3((unsigned)nb), dim3(256), 0, 0, db.Current(), db.Alternate(), (const uint32_t *)seg_first, 4, 45, -1, false, gm::SegOut{nullptr, 0, 0, 37, 8, nullptr});
            hipLaunchKernelGGL((gm::k_order_segments<16>), dim3((unsigned)nb), dim3(256), 0, 0, db.Current(), db.Alternate(), (const uint32_t *)seg_first, 4, 45, 1024, false, gm::SegOut{nullptr, 0, 0, 37, 8, nullptr});
            hipLaunchKernelGGL((gm::k_order_segments<32>), dim3((unsigned)nb), dim3(256), 0, 0, db.Current(), db.Alternate(), (const uint32_t *)seg_first, 4, 45, 4096, true, gm::SegOut{nullptr, 0, 0, 37, 8, nullptr});
         } else {
            CK(hipcub::DeviceRadixSort::SortKeys(nullptr, need, db, (int)n, 45, 61, 0));
            if (need > tmp_bytes) { if (tmp) CK(hipFree(tmp)); CK(hipMalloc(&tmp, need)); tmp_bytes = need; }
            CK(hipcub::DeviceRadixSort::SortKeys(tmp, need, db, (int)n, 45, 61, 0));
            const int nb = (int)((R >> 8) + 1);
            hipLaunchKernelGGL(k_bounds, dim3((nb + 256) / 256), dim3(256), 0, 0, db.Current(), n, 45, nb, off);
            CK(hipcub::DeviceSegmentedRadixSort::SortKeys(nullptr, need, db, (int)n, nb, off, off + 1, 4, 45, 0));
            if (need > tmp_bytes) { if (tmp) CK(hipFree(tmp)); CK(hipMalloc(&tmp, need)); tmp_bytes = need; }
            CK(hipcub::DeviceSegmentedRadixSort::SortKeys(tmp, need, db, (int)n, nb, off, off + 1, 4, 45, 0));
         }
         CK(hipEventRecord(e1, 0));
         CK(hipEventSynchronize(e1));
         float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
         CK(hipMemcpy(got.data(), db.Current(), 8 * (size_t)n, hipMemcpyDeviceToHost));
         long bad = 0;
         if (strcmp(mode, "byread")) for (long i = 0; i < n; i++) bad += got[(size_t)i] != want[(size_t)i];
         else for (long i = 1; i < n; i++) bad += (got[(size_t)i] >> 37) < (got[(size_t)i - 1] >> 37);
         printf("{\"mode\": \"%s\", \"keys\": %ld, \"reads\": %ld, \"shape\": \"%s\", \"ms\": %.3f, \"wrong\": %ld}\n", mode, n, R, heavy ? "heavy" : real ? "real" : "uniform", ms, bad);
         fflush(stdout);
      }
   }
   return 0;
}
