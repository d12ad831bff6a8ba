// copy_bench: device -> page-locked host copies by a kernel of G workgroups against hipMemcpyAsync (which on this
// platform is a kernel over the whole device).  Prints GB/s per variant.  tools/copy_bench [MiB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int NT, int UNROLL>
__global__ void __launch_bounds__(256) k_copy(u32x4 *dst, const u32x4 *src, int64_t n16)
{
   const int64_t stride = (int64_t)gridDim.x * 256;
   int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x;
   for (; i + (UNROLL - 1) * stride < n16; i += UNROLL * stride) {
      u32x4 v[UNROLL];
#pragma unroll
      for (int k = 0; k < UNROLL; k++) v[k] = src[i + k * stride];
#pragma unroll
      for (int k = 0; k < UNROLL; k++) { if (NT) __builtin_nontemporal_store(v[k], dst + i + k * stride); else dst[i + k * stride] = v[k]; }
   }
   for (; i < n16; i += stride) dst[i] = src[i];
}

// a workgroup copies contiguous 4-KB pieces (each wave 1 KB contiguous per store instruction)
template <int UNROLL>
__global__ void __launch_bounds__(256) k_copy_blocked(u32x4 *dst, const u32x4 *src, int64_t n16)
{
   const int64_t per = 256 * UNROLL;
   for (int64_t base = blockIdx.x * per; base < n16; base += (int64_t)gridDim.x * per) {
      u32x4 v[UNROLL];
#pragma unroll
      for (int k = 0; k < UNROLL; k++) { int64_t i = base + k * 256 + threadIdx.x; if (i < n16) v[k] = src[i]; }
#pragma unroll
      for (int k = 0; k < UNROLL; k++) { int64_t i = base + k * 256 + threadIdx.x; if (i < n16) __builtin_nontemporal_store(v[k], dst + i); }
   }
}

int main(int argc, char **argv)
{
   const size_t mib = argc > 1 ? atoi(argv[1]) : 512;
   const size_t bytes = mib << 20;
   void *d = nullptr, *h = nullptr, *hd = nullptr;
   CK(hipMalloc(&d, bytes)); CK(hipMemset(d, 1, bytes));
   CK(hipHostMalloc(&h, bytes, hipHostMallocDefault));
   CK(hipHostGetDevicePointer(&hd, h, 0));
   hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
   hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
   auto timeit = [&](const char *name, int g, auto launch) -> int {
      launch(); CK(hipStreamSynchronize(st));
      CK(hipEventRecord(e0, st));
      for (int r = 0; r < 3; r++) launch();
      CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
      float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("{\"variant\": \"%s\", \"workgroups\": %d, \"GBps\": %.2f}\n", name, g, 3.0 * bytes / (ms * 1e-3) / 1e9);
      return 0;
   };
   const int64_t n16 = (int64_t)(bytes / 16);
   if (timeit("hipMemcpyAsync", 0, [&] { (void)hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, st); })) return 1;
   for (int g : {8, 16, 32, 64, 128, 256, 1024, 4096}) {
      if (timeit("strided_nt_x4", g, [&] { hipLaunchKernelGGL((k_copy<1, 4>), dim3(g), dim3(256), 0, st, (u32x4 *)hd, (const u32x4 *)d, n16); })) return 1;
      if (timeit("strided_plain_x4", g, [&] { hipLaunchKernelGGL((k_copy<0, 4>), dim3(g), dim3(256), 0, st, (u32x4 *)hd, (const u32x4 *)d, n16); })) return 1;
      if (timeit("strided_nt_x1", g, [&] { hipLaunchKernelGGL((k_copy<1, 1>), dim3(g), dim3(256), 0, st, (u32x4 *)hd, (const u32x4 *)d, n16); })) return 1;
      if (timeit("blocked_nt_x4", g, [&] { hipLaunchKernelGGL((k_copy_blocked<4>), dim3(g), dim3(256), 0, st, (u32x4 *)hd, (const u32x4 *)d, n16); })) return 1;
      if (timeit("blocked_nt_x8", g, [&] { hipLaunchKernelGGL((k_copy_blocked<8>), dim3(g), dim3(256), 0, st, (u32x4 *)hd, (const u32x4 *)d, n16); })) return 1;
   }
   return 0;
}
