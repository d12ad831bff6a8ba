// gm_multi.cpp -- several GPUs behind the C-ABI: the index is built or uploaded ONCE, on the first
// member's GPU; every other GPU receives the device arrays (rank blocks, suffix array, tables, text) by one
// ncclBroadcast each over xGMI (RCCL: one communicator per GPU, ncclCommInitAll, the broadcasts of all ranks in
// one group); members that share a GPU with an earlier member adopt that member's arrays.  Reads are split into
// contiguous ranges in input order, one host thread per member maps its range, the hit lists are concatenated in
// member order: the output is that of one GPU.  No collective runs per batch.
// librccl is loaded with dlopen when the first broadcast is needed: a process with one GPU never loads it.
// Host code only (HIP runtime API, no kernels).
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include "gm_host.hpp"

#define HIP_TRY(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
   std::fprintf(stderr, "%s\t(%s:%d) error: %s: %s\n", __func__, __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
   errno = 1; goto failure_return; } } while (0)

namespace {
// The few entry points of rccl.h this file calls (ncclResult_t = int, ncclSuccess = 0; ncclUint8 = 1).
struct Rccl {
   void *lib = nullptr;
   int (*CommInitAll)(void **comms, int ndev, const int *devlist) = nullptr;
   int (*CommDestroy)(void *comm) = nullptr;
   int (*GroupStart)() = nullptr;
   int (*GroupEnd)() = nullptr;
   int (*Broadcast)(const void *send, void *recv, size_t count, int datatype, int root, void *comm, hipStream_t stream) = nullptr;
   const char *(*GetErrorString)(int) = nullptr;
   bool load()
   {
      if (lib) return true;
      for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
         lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
         if (lib) break;
      }
      if (!lib) return false;
      CommInitAll = (decltype(CommInitAll))dlsym(lib, "ncclCommInitAll");
      CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
      GroupStart = (decltype(GroupStart))dlsym(lib, "ncclGroupStart");
      GroupEnd = (decltype(GroupEnd))dlsym(lib, "ncclGroupEnd");
      Broadcast = (decltype(Broadcast))dlsym(lib, "ncclBroadcast");
      GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
      return CommInitAll && CommDestroy && GroupStart && GroupEnd && Broadcast && GetErrorString;
   }
};
Rccl g_rccl;
constexpr int NCCL_UINT8 = 1;

#define NCCL_TRY(call) do { int r_ = (call); if (r_ != 0) { \
   std::fprintf(stderr, "%s\t(%s:%d) error: %s: %s\n", __func__, __FILE__, __LINE__, #call, g_rccl.GetErrorString(r_)); \
   errno = 1; goto failure_return; } } while (0)

struct DeviceCopy {            // the arrays of the index on one GPU other than the root's
   int device = -1;
   void *buf[4] = {nullptr, nullptr, nullptr, nullptr};
};
}  // namespace

struct gpumap_multi {
   std::vector<gpumap_ctx *> members;
   std::vector<bool> owned;               // members this object created (the root may be the caller's)
   std::vector<DeviceCopy> copies;
   bool used_rccl = false;
   double bcast_ms = 0;                   // set-up times: the broadcast of the arrays, and per member its own tables (jump table, bit planes)
   std::vector<double> member_ms;
};

extern "C" void gpumap_multi_destroy(gpumap_multi *m)
{
   if (!m) return;
   for (size_t i = m->members.size(); i-- > 0;)
      if (m->owned[i]) gpumap_destroy(m->members[i]);
   for (auto &c : m->copies) {
      (void)hipSetDevice(c.device);
      for (void *p : c.buf) if (p) (void)hipFree(p);
   }
   delete m;
}

// One ncclBroadcast per array from rank 0 (the root member's GPU) to the GPUs of `copies`; ranks = distinct GPUs.
static int broadcast_arrays(const gpumap_devindex &root, int root_device, std::vector<DeviceCopy> &copies)
{
   const int n = (int)copies.size() + 1;
   std::vector<int> devs(1, root_device);
   std::vector<void *> comms((size_t)n, nullptr);
   std::vector<hipStream_t> streams((size_t)n, nullptr);
   const void *src[4] = {root.occ_blocks, root.sa_words, root.tables, root.text};
   const int64_t bytes[4] = {root.occ_bytes, root.sa_bytes, root.tables_bytes, root.text_bytes};
   bool ok = false;
   GM_FAIL_IF(!g_rccl.load(), "cannot load librccl.so (needed to replicate the index over several GPUs).");
   for (auto &c : copies) devs.push_back(c.device);
   for (int r = 0; r < n; r++) {
      HIP_TRY(hipSetDevice(devs[(size_t)r]));
      HIP_TRY(hipStreamCreateWithFlags(&streams[(size_t)r], hipStreamNonBlocking));
   }
   NCCL_TRY(g_rccl.CommInitAll(comms.data(), n, devs.data()));
   for (int a = 0; a < 4; a++) {
      if (bytes[a] == 0) continue;
      // in pieces of 1 GiB: the suffix array of a GRCh38-size text is 25.6 GB
      for (int64_t lo = 0; lo < bytes[a]; lo += (int64_t)1 << 30) {
         const size_t cnt = (size_t)std::min<int64_t>((int64_t)1 << 30, bytes[a] - lo);
         NCCL_TRY(g_rccl.GroupStart());
         for (int r = 0; r < n; r++) {
            HIP_TRY(hipSetDevice(devs[(size_t)r]));
            char *mine = r == 0 ? (char *)const_cast<void *>(src[a]) : (char *)copies[(size_t)r - 1].buf[a];
            NCCL_TRY(g_rccl.Broadcast(mine + lo, mine + lo, cnt, NCCL_UINT8, 0, comms[(size_t)r], streams[(size_t)r]));
         }
         NCCL_TRY(g_rccl.GroupEnd());
      }
   }
   for (int r = 0; r < n; r++) {
      HIP_TRY(hipSetDevice(devs[(size_t)r]));
      HIP_TRY(hipStreamSynchronize(streams[(size_t)r]));
   }
   ok = true;
failure_return:
   for (int r = 0; r < n; r++) {
      if (comms[(size_t)r]) (void)g_rccl.CommDestroy(comms[(size_t)r]);
      if (streams[(size_t)r]) { (void)hipSetDevice(devs[(size_t)r]); (void)hipStreamDestroy(streams[(size_t)r]); }
   }
   return ok ? 0 : -1;
}

extern "C" gpumap_multi *gpumap_multi_from_ctx(gpumap_ctx *root, int root_device, int n_members, const int *devices)
{
   gpumap_multi *m = nullptr;
   gpumap_devindex dev;
   std::vector<uint8_t> header(gpumap_devindex_header_size());
   int count = 0;
   GM_FAIL_IF(root == nullptr, "argument 'root' is NULL.");
   GM_FAIL_IF(n_members < 1 || n_members > 64, "n_members out of range (1..64).");
   HIP_TRY(hipGetDeviceCount(&count));
   for (int i = 0; i < n_members; i++) {
      const int d = devices ? devices[i] : i;
      GM_FAIL_IF(d < 0 || d >= count, "no such HIP device.");
   }
   GM_FAIL_IF((devices ? devices[0] : 0) != root_device, "the first member lives on the root context's device.");
   GM_FAIL_IF(gpumap_devindex_export(root, &dev, header.data()) != 0, "cannot export the root's index.");
   m = new gpumap_multi();
   m->members.push_back(root);
   m->owned.push_back(false);
   {
      // the GPUs that need a copy, in the order of their first member
      const bool force = getenv("GPUMAP_MULTI_FORCE_RCCL") != nullptr;      // tests on one GPU: a broadcast on a communicator of one rank
      for (int i = 1; i < n_members; i++) {
         const int d = devices ? devices[i] : i;
         bool have = d == root_device;
         for (auto &c : m->copies) have = have || c.device == d;
         if (have) continue;
         DeviceCopy c;
         c.device = d;
         HIP_TRY(hipSetDevice(d));
         const int64_t bytes[4] = {dev.occ_bytes, dev.sa_bytes, dev.tables_bytes, dev.text_bytes};
         m->copies.push_back(c);
         for (int a = 0; a < 4; a++)
            if (bytes[a] > 0) HIP_TRY(hipMalloc(&m->copies.back().buf[a], (size_t)bytes[a]));
      }
      if (!m->copies.empty() || force) {
         const auto t0 = std::chrono::steady_clock::now();
         GM_FAIL_IF(broadcast_arrays(dev, root_device, m->copies) != 0, "cannot broadcast the index.");
         m->bcast_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
         m->used_rccl = true;
      }
      // every member derives its own jump table and bit planes on its GPU (1.5 s + at GRCh38 size): all at once, a thread
      // per member (members that share a GPU queue up on it)
      std::vector<gpumap_ctx *> made((size_t)n_members, nullptr);
      std::vector<std::thread> th;
      m->member_ms.assign((size_t)n_members, 0.0);
      for (int i = 1; i < n_members; i++)
         th.emplace_back([&, i] {
            const int d = devices ? devices[i] : i;
            const auto t0 = std::chrono::steady_clock::now();
            gpumap_devindex mine = dev;
            for (auto &c : m->copies)
               if (c.device == d) { mine.occ_blocks = c.buf[0]; mine.sa_words = c.buf[1]; mine.tables = c.buf[2]; mine.text = c.buf[3]; }
            made[(size_t)i] = gpumap_create_from_device(&mine, header.data(), d);
            m->member_ms[(size_t)i] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
         });
      for (auto &t : th) t.join();
      bool all = true;
      for (int i = 1; i < n_members; i++) {
         if (made[(size_t)i] == nullptr) { all = false; continue; }
         m->members.push_back(made[(size_t)i]);
         m->owned.push_back(true);
      }
      GM_FAIL_IF(!all, "cannot create a member.");
      if (getenv("GPUMAP_VERBOSE")) {
         fprintf(stderr, "gpumap_multi: broadcast %.0f ms; members:", m->bcast_ms);
         for (int i = 1; i < n_members; i++) fprintf(stderr, " %.0f", m->member_ms[(size_t)i]);
         fprintf(stderr, " ms\n");
      }
   }
   return m;
failure_return:
   gpumap_multi_destroy(m);
   return nullptr;
}

extern "C" gpumap_multi *gpumap_create_multi(const gpumap_index_view *view, int n_members, const int *devices)
{
   gpumap_multi *m = nullptr;
   gpumap_ctx *root = nullptr;
   GM_FAIL_IF(n_members < 1, "n_members out of range.");
   root = gpumap_create(view, devices ? devices[0] : 0);
   GM_FAIL_IF(root == nullptr, "cannot create the first member.");
   m = gpumap_multi_from_ctx(root, devices ? devices[0] : 0, n_members, devices);
   GM_FAIL_IF(m == nullptr, "cannot replicate the index.");
   m->owned[0] = true;
   return m;
failure_return:
   gpumap_destroy(root);
   return nullptr;
}

extern "C" int gpumap_multi_size(const gpumap_multi *m) { return m ? (int)m->members.size() : 0; }
extern "C" int gpumap_multi_used_rccl(const gpumap_multi *m) { return m && m->used_rccl ? 1 : 0; }
// Milliseconds of the set-up: i = -1 the broadcast of the arrays, i >= 1 what member i took to derive its own tables.
extern "C" double gpumap_multi_setup_ms(const gpumap_multi *m, int i)
{
   if (!m) return 0.0;
   if (i < 0) return m->bcast_ms;
   return i < (int)m->member_ms.size() ? m->member_ms[(size_t)i] : 0.0;
}
extern "C" gpumap_ctx *gpumap_multi_ctx(const gpumap_multi *m, int i)
{
   return (m && i >= 0 && i < (int)m->members.size()) ? m->members[(size_t)i] : nullptr;
}
extern "C" gpumap_ctx *const *gpumap_multi_ctxs(const gpumap_multi *m) { return m ? m->members.data() : nullptr; }

namespace {
template <class T> T *cat(const std::vector<gpumap_hits> &parts, T *gpumap_hits::*field, int64_t gpumap_hits::*count, int64_t per, int64_t total)
{
   T *out = (T *)malloc(sizeof(T) * (size_t)std::max<int64_t>(total * per, 1));
   if (!out) return nullptr;
   int64_t at = 0;
   for (const auto &p : parts) {
      const int64_t n = p.*count * per;
      if (n > 0) memcpy(out + at, p.*field, sizeof(T) * (size_t)n);
      at += n;
   }
   return out;
}
}  // namespace

// gpumap_map_batch over the members: member g maps the reads [g R / G, (g + 1) R / G) on its own thread; the hit lists
// are concatenated in member order, read indices and CSR offsets moved to the whole batch's numbering.
extern "C" int gpumap_multi_map_batch(gpumap_multi *m, const uint8_t *codes, const int64_t *offsets, int64_t n_reads,
                                      int tau, int want_intervals, gpumap_hits *out)
{
   std::vector<gpumap_hits> parts;
   std::vector<int> rc;
   std::vector<std::thread> th;
   GM_FAIL_IF(out == nullptr, "argument 'out' is NULL.");
   memset(out, 0, sizeof *out);
   GM_FAIL_IF(m == nullptr, "argument 'multi' is NULL.");
   GM_FAIL_IF(n_reads < 0 || offsets == nullptr, "invalid arguments.");
   {
      const int64_t G = (int64_t)m->members.size();
      parts.resize((size_t)G);
      rc.assign((size_t)G, 0);
      for (auto &p : parts) memset(&p, 0, sizeof p);
      for (int64_t g = 0; g < G; g++)
         th.emplace_back([&, g] {
            const int64_t lo = n_reads * g / G, hi = n_reads * (g + 1) / G;
            gpumap_hits &part = parts[(size_t)g];
            rc[(size_t)g] = gpumap_map_batch(m->members[(size_t)g], codes, offsets + lo, hi - lo, tau, want_intervals, &part);
            // a member with option "compact_results" returns keys only: the concatenation below works on the full arrays
            if (rc[(size_t)g] == 0 && part.loc_key != nullptr && part.loc_read == nullptr)
               rc[(size_t)g] = gpumap_hits_expand(m->members[(size_t)g], &part);
         });
      for (auto &t : th) t.join();
      for (int x : rc) GM_FAIL_IF(x != 0, "a member failed.");
      int64_t L = 0, H = 0;
      for (auto &p : parts) { L += p.n_locs; H += p.n_hits; out->n_ext += p.n_ext; out->ms_total = std::max(out->ms_total, p.ms_total); }
      out->n_reads = n_reads; out->n_locs = L; out->n_hits = H;
      out->key_pos_bits = parts[0].key_pos_bits;
      out->status = cat(parts, &gpumap_hits::status, &gpumap_hits::n_reads, 1, n_reads);
      out->read_first_loc = (int64_t *)malloc(8 * (size_t)(n_reads + 1));
      out->loc_read = cat(parts, &gpumap_hits::loc_read, &gpumap_hits::n_locs, 1, L);
      out->loc_pos = cat(parts, &gpumap_hits::loc_pos, &gpumap_hits::n_locs, 1, L);
      out->loc_fwd = cat(parts, &gpumap_hits::loc_fwd, &gpumap_hits::n_locs, 1, L);
      out->loc_seq = cat(parts, &gpumap_hits::loc_seq, &gpumap_hits::n_locs, 1, L);
      out->loc_seqpos = cat(parts, &gpumap_hits::loc_seqpos, &gpumap_hits::n_locs, 1, L);
      out->loc_strand = cat(parts, &gpumap_hits::loc_strand, &gpumap_hits::n_locs, 1, L);
      out->loc_score = cat(parts, &gpumap_hits::loc_score, &gpumap_hits::n_locs, 1, L);
      GM_FAIL_IF(!out->status || !out->read_first_loc || !out->loc_read || !out->loc_pos || !out->loc_fwd || !out->loc_seq ||
                 !out->loc_seqpos || !out->loc_strand || !out->loc_score, "memory error.");
      if (want_intervals) {
         out->hit_read = cat(parts, &gpumap_hits::hit_read, &gpumap_hits::n_hits, 1, H);
         out->hit_fp = cat(parts, &gpumap_hits::hit_fp, &gpumap_hits::n_hits, 1, H);
         out->hit_rp = cat(parts, &gpumap_hits::hit_rp, &gpumap_hits::n_hits, 1, H);
         out->hit_sz = cat(parts, &gpumap_hits::hit_sz, &gpumap_hits::n_hits, 1, H);
         out->hit_dp = cat(parts, &gpumap_hits::hit_dp, &gpumap_hits::n_hits, 1, H);
         out->hit_score = cat(parts, &gpumap_hits::hit_score, &gpumap_hits::n_hits, 1, H);
         out->hit_align = cat(parts, &gpumap_hits::hit_align, &gpumap_hits::n_hits, GPUMAP_ALIGN_WORDS, H);
         GM_FAIL_IF(!out->hit_read || !out->hit_fp || !out->hit_rp || !out->hit_sz || !out->hit_dp || !out->hit_score || !out->hit_align, "memory error.");
      }
      int64_t loc_base = 0, hit_base = 0;
      for (int64_t g = 0; g < G; g++) {
         const gpumap_hits &p = parts[(size_t)g];
         const int64_t lo = n_reads * g / G;
         for (int64_t r = 0; r < p.n_reads; r++) out->read_first_loc[lo + r] = loc_base + p.read_first_loc[r];
         for (int64_t t = 0; t < p.n_locs; t++) out->loc_read[loc_base + t] += lo;
         if (want_intervals) for (int64_t t = 0; t < p.n_hits; t++) out->hit_read[hit_base + t] += lo;
         loc_base += p.n_locs; hit_base += p.n_hits;
      }
      out->read_first_loc[n_reads] = L;
   }
   for (auto &p : parts) gpumap_hits_free(&p);
   return 0;
failure_return:
   for (auto &t : th) if (t.joinable()) t.join();
   for (auto &p : parts) gpumap_hits_free(&p);
   if (out) gpumap_hits_free(out);
   return -1;
}
