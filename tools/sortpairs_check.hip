// Check of hipcub::DeviceRadixSort with begin_bit > 0 (ROCm 7.2.0, gfx950).  Found with it: with end_bit = 64 and
// begin_bit > 0 SortPairs / SortKeys return unsorted keys (and pairs torn apart) for about 10^4 .. 10^6 items; with
// end_bit < 64, or begin_bit = 0, or a few thousand or several million items they are right.  The library only uses
// begin_bit > 0 with end_bit < 64 (gm_device.hip, "order" and the interval sort of gpumap_ann_build).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <cstdio>
#include <vector>
#include <random>
#define CK(x) do { if ((x) != hipSuccess) { printf("HIP error at line %d\n", __LINE__); return 1; } } while (0)
int main()
{
   const int sizes[] = {1000, 5000, 12486, 38012, 100000, 300000, 1000000, 2500000, 5000000};
   const int ranges[][2] = {{0, 64}, {8, 64}, {40, 64}, {5, 61}, {7, 55}, {39, 63}, {39, 62}, {1, 63}};
   long total_bad = 0;
   for (int n : sizes)
      for (auto &r : ranges) {
         const int begin = r[0], end = r[1];
         std::mt19937_64 rng(n * 131 + begin * 7 + end);
         std::vector<unsigned long long> k(n);
         std::vector<unsigned int> v(n);
         const unsigned long long mask = end == 64 ? ~0ull : ((1ull << end) - 1ull);
         for (int i = 0; i < n; i++) { k[i] = (rng() % 20 == 0 ? ~0ull : rng()) & mask; v[i] = i; }
         unsigned long long *dk0, *dk1; unsigned int *dv0, *dv1;
         CK(hipMalloc(&dk0, 8 * (size_t)n)); CK(hipMalloc(&dk1, 8 * (size_t)n)); CK(hipMalloc(&dv0, 4 * (size_t)n)); CK(hipMalloc(&dv1, 4 * (size_t)n));
         long bad[2][2] = {{0, 0}, {0, 0}};
         for (int pairs = 0; pairs < 2; pairs++) {
            CK(hipMemcpy(dk0, k.data(), 8 * (size_t)n, hipMemcpyHostToDevice)); CK(hipMemcpy(dv0, v.data(), 4 * (size_t)n, hipMemcpyHostToDevice));
            hipcub::DoubleBuffer<unsigned long long> dk(dk0, dk1);
            hipcub::DoubleBuffer<unsigned int> dv(dv0, dv1);
            size_t tmp = 0; void *t = nullptr;
            if (pairs) CK(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp, dk, dv, n, begin, end, 0));
            else CK(hipcub::DeviceRadixSort::SortKeys(nullptr, tmp, dk, n, begin, end, 0));
            CK(hipMalloc(&t, tmp));
            if (pairs) CK(hipcub::DeviceRadixSort::SortPairs(t, tmp, dk, dv, n, begin, end, 0));
            else CK(hipcub::DeviceRadixSort::SortKeys(t, tmp, dk, n, begin, end, 0));
            CK(hipDeviceSynchronize());
            std::vector<unsigned long long> ok(n); std::vector<unsigned int> ov(n);
            CK(hipMemcpy(ok.data(), dk.Current(), 8 * (size_t)n, hipMemcpyDeviceToHost));
            if (pairs) CK(hipMemcpy(ov.data(), dv.Current(), 4 * (size_t)n, hipMemcpyDeviceToHost));
            for (int i = 0; i < n; i++) {
               if (i && (ok[i - 1] >> begin) > (ok[i] >> begin)) bad[pairs][0]++;
               if (pairs && k[ov[i]] != ok[i]) bad[pairs][1]++;
            }
            CK(hipFree(t));
         }
         printf("n %8d bits [%2d, %2d): keys out of order %ld | pairs out of order %ld, torn %ld\n", n, begin, end, bad[0][0], bad[1][0], bad[1][1]);
         if (end < 64 || begin == 0) total_bad += bad[0][0] + bad[1][0] + bad[1][1];
         CK(hipFree(dk0)); CK(hipFree(dk1)); CK(hipFree(dv0)); CK(hipFree(dv1));
      }
   printf("wrong results in the ranges the library uses: %ld\n", total_bad);
   return total_bad != 0;
}
