// gather_bench.hip -- the "HBM gather roofline" microbenchmark: R independent
// DEPENDENT chains of random cache-line reads over a footprint of the index's size, in the access
// shapes the search kernel could use.  Prints one JSON line per (shape, footprint, occupancy).
//
//   quad64   4 lanes x 16 B = one 64-B line per chain step   (what k_search does), 16 chains / wave
//   quad64x2 same, two independent lines per step            (sp-1 and ep of one extension)
//   oct128   8 lanes x 16 B = one 128-B line per step, 8 chains / wave
//   lane64   1 lane  x 64 B (4 dwordx4) per step, 64 chains / wave
//   lane16   1 lane  x 16 B per step, 64 chains / wave
//
// build: hipcc --offload-arch=gfx950 -O3 -o gather_bench tools/gather_bench.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x)
{
   x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
   return x;
}

__global__ void k_fill(uint4 *buf, uint64_t n16)
{
   uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
   uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
   for (; i < n16; i += stride) {
      uint32_t a = mix((uint32_t)i), b = mix((uint32_t)(i >> 32) + a);
      buf[i] = make_uint4(a, b, a ^ 0x9e3779b9U, b + a);
   }
}

template <int GROUP>   // lanes per chain: 4 or 8; one 16-B load per lane
__device__ __forceinline__ uint32_t group_xor(uint32_t x)
{
   x ^= (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0xB1, 0xF, 0xF, true);
   x ^= (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x4E, 0xF, 0xF, true);
   if (GROUP == 8) x ^= (uint32_t)__shfl_xor((int)x, 4);
   return x;
}

// mode 0 quad64, 1 quad64x2, 2 oct128, 3 lane64, 4 lane16
template <int MODE>
__global__ void __launch_bounds__(256) k_chase(const uint4 *buf, uint64_t n_lines64, int steps, uint32_t *sink)
{
   const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
   const int lane = threadIdx.x & 63;
   uint32_t acc = 0;
   if (MODE == 0 || MODE == 1) {
      const int sub = lane & 3;
      uint32_t s = mix((gid >> 2) * 2654435761U + 12345U);
      for (int it = 0; it < steps; it++) {
         uint64_t line = ((uint64_t)s * n_lines64) >> 32;
         uint4 v = buf[line * 4 + sub];
         uint32_t x = v.x ^ v.y ^ v.z ^ v.w;
         if (MODE == 1) {
            uint64_t line2 = ((uint64_t)mix(s + 7U) * n_lines64) >> 32;
            uint4 w = buf[line2 * 4 + sub];
            x ^= w.x + w.y + w.z + w.w;
         }
         x = group_xor<4>(x);
         s = mix(s ^ x);
         acc += x;
      }
   } else if (MODE == 2) {
      const int sub = lane & 7;
      uint32_t s = mix((gid >> 3) * 2654435761U + 12345U);
      const uint64_t n128 = n_lines64 >> 1;
      for (int it = 0; it < steps; it++) {
         uint64_t line = ((uint64_t)s * n128) >> 32;
         uint4 v = buf[line * 8 + sub];
         uint32_t x = group_xor<8>(v.x ^ v.y ^ v.z ^ v.w);
         s = mix(s ^ x);
         acc += x;
      }
   } else if (MODE == 3) {
      uint32_t s = mix(gid * 2654435761U + 12345U);
      for (int it = 0; it < steps; it++) {
         uint64_t line = ((uint64_t)s * n_lines64) >> 32;
         uint4 a = buf[line * 4 + 0], b = buf[line * 4 + 1], c = buf[line * 4 + 2], d = buf[line * 4 + 3];
         uint32_t x = a.x ^ a.w ^ b.y ^ b.z ^ c.x ^ c.w ^ d.y ^ d.z;
         s = mix(s ^ x);
         acc += x;
      }
   } else {
      uint32_t s = mix(gid * 2654435761U + 12345U);
      for (int it = 0; it < steps; it++) {
         uint64_t line = ((uint64_t)s * n_lines64) >> 32;
         uint4 a = buf[line * 4 + (s & 3)];
         uint32_t x = a.x ^ a.y ^ a.z ^ a.w;
         s = mix(s ^ x);
         acc += x;
      }
   }
   if (acc == 0x12345678U) sink[0] = acc;   // keep the chain alive
}

template <int MODE>
static void run(const char *name, const uint4 *buf, uint64_t bytes, int waves_per_cu, int steps, int cus, uint32_t *sink,
                int bytes_per_touch, int lanes_per_chain, int touches_per_step)
{
   int blocks = cus * waves_per_cu / 4;
   hipEvent_t e0, e1;
   CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
   hipLaunchKernelGGL(k_chase<MODE>, dim3(blocks), dim3(256), 0, 0, buf, bytes / 64, steps / 4, sink);   // warm-up
   CK(hipDeviceSynchronize());
   CK(hipEventRecord(e0));
   hipLaunchKernelGGL(k_chase<MODE>, dim3(blocks), dim3(256), 0, 0, buf, bytes / 64, steps, sink);
   CK(hipEventRecord(e1));
   CK(hipEventSynchronize(e1));
   float ms = 0;
   CK(hipEventElapsedTime(&ms, e0, e1));
   double chains = (double)blocks * 256 / lanes_per_chain;
   double touches = chains * steps * touches_per_step;
   printf("{\"shape\": \"%s\", \"footprint_MiB\": %.0f, \"waves_per_cu\": %d, \"chains\": %.0f, \"steps\": %d, \"ms\": %.3f, "
          "\"Gtouch_per_s\": %.2f, \"GB_per_s\": %.1f, \"ns_per_step\": %.1f}\n",
          name, bytes / 1048576.0, waves_per_cu, chains, steps, ms, touches / ms / 1e6,
          touches * bytes_per_touch / ms / 1e6, ms * 1e6 / steps);
   fflush(stdout);
   CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}

int main(int argc, char **argv)
{
   if (argc > 1 && !strcmp(argv[1], "quick")) {
      // gather_bench quick <footprint MiB> [waves per CU] [steps]: one line, the quad64 shape (what k_search does)
      // over one footprint -- the ceiling bench.py quotes beside the search kernel's request rate
      const uint64_t mib = argc > 2 ? strtoull(argv[2], nullptr, 10) : 4096;
      const int waves = argc > 3 ? atoi(argv[3]) : 32, qsteps = argc > 4 ? atoi(argv[4]) : 2000;
      hipDeviceProp_t prop;
      CK(hipGetDeviceProperties(&prop, 0));
      uint4 *buf; uint32_t *sink;
      CK(hipMalloc(&buf, mib << 20));
      CK(hipMalloc(&sink, 64));
      hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, buf, (mib << 20) / 16);
      CK(hipDeviceSynchronize());
      run<0>("quad64", buf, mib << 20, waves, qsteps, prop.multiProcessorCount, sink, 64, 4, 1);
      return 0;
   }
   int steps = argc > 1 ? atoi(argv[1]) : 2000;
   hipDeviceProp_t prop;
   CK(hipGetDeviceProperties(&prop, 0));
   int cus = prop.multiProcessorCount;
   const int big_gib = argc > 2 ? atoi(argv[2]) : 0;      // second argument: also footprints up to that many GiB (TLB reach)
   uint64_t max_bytes = (uint64_t)(big_gib > 4 ? big_gib : 4) << 30;
   uint4 *buf; uint32_t *sink;
   CK(hipMalloc(&buf, max_bytes));
   CK(hipMalloc(&sink, 64));
   hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, buf, max_bytes / 16);
   CK(hipDeviceSynchronize());
   printf("{\"device\": \"%s\", \"cus\": %d}\n", prop.name, cus);
   const uint64_t foot[] = {(uint64_t)16 << 20, (uint64_t)192 << 20, (uint64_t)1 << 30, (uint64_t)4 << 30};
   const int occ[] = {8, 16, 32};
   for (uint64_t f : foot)
      for (int w : occ) {
         run<0>("quad64", buf, f, w, steps, cus, sink, 64, 4, 1);
         run<1>("quad64x2", buf, f, w, steps, cus, sink, 64, 4, 2);
         run<2>("oct128", buf, f, w, steps, cus, sink, 128, 8, 1);
         run<3>("lane64", buf, f, w, steps, cus, sink, 64, 1, 1);
         run<4>("lane16", buf, f, w, steps, cus, sink, 16, 1, 1);
      }
   for (uint64_t f = (uint64_t)8 << 30; f <= max_bytes; f <<= 1) {      // beyond 4 GiB: 32 waves per CU only
      run<0>("quad64", buf, f, 32, steps, cus, sink, 64, 4, 1);
      run<3>("lane64", buf, f, 32, steps, cus, sink, 64, 1, 1);
      run<4>("lane16", buf, f, 32, steps, cus, sink, 16, 1, 1);
   }
   return 0;
}
