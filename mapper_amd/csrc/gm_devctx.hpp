// gm_devctx.hpp -- device context shared by the translation units that launch kernels
// (gm_device.hip: query path; gm_build.hip: index construction on the device).
#pragma once
#include <hip/hip_runtime.h>
#include <cerrno>
#include <cstdio>
#include <mutex>
#include <vector>
#include "gm_host.hpp"
#include "gm_core.hpp"

namespace gm {
struct IndexDev {
   const uint32_t *occ;        // rank blocks, 16 dwords each
   const uint64_t *tables;
   const uint64_t *sa;         // packed suffix array, verbatim
   const int64_t  *seq_beg;
   int64_t n;                  // text length
   int64_t n_blocks, n_super;
   int32_t sar_bits;
   int32_t seq_cnt;
};
constexpr int MAX_SB_LDS = 256;      // superblocks whose counts are staged in LDS (n < 2^33)
constexpr int TAB_C = 0;             // `tables` (64-bit words): [0..4] C[0..4];
constexpr int TAB_SBASE = 8;         // [8 + 5*s + j] absolute count of symbol j in BWT[0, s << 25)
}  // namespace gm

#define HIP_TRY(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
   std::fprintf(stderr, "%s\t(%s:%d) error: %s: %s\n", __func__, __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
   errno = 1; goto failure_return; } } while (0)

struct DevBuf {
   void *p = nullptr;
   size_t cap = 0;
   hipError_t ensure(size_t bytes)
   {
      if (bytes <= cap) return hipSuccess;
      if (p) { hipError_t e = hipFree(p); p = nullptr; cap = 0; if (e != hipSuccess) return e; }
      size_t want = bytes + bytes / 4 + 256;
      hipError_t e = hipMalloc(&p, want);
      if (e != hipSuccess) { e = hipMalloc(&p, bytes); want = bytes; }
      if (e == hipSuccess) cap = want;
      return e;
   }
   void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
   template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

struct DevIndexHeader {          // host blob that travels with an exported device index
   uint64_t magic;
   int64_t  n, n_blocks, n_super, sar_bits, sar_len, seq_cnt, tables_words;
   uint8_t  sym_table[256];
   uint8_t  com_table[8];
};
constexpr uint64_t DEVINDEX_MAGIC = 0x47504d4158444e49ULL;

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

struct gpumap_ctx {
   int device = 0;
   hipStream_t stream = nullptr;
   bool owns_index = true;
   gm::IndexDev ix{};
   void *d_occ = nullptr, *d_tables = nullptr, *d_sa = nullptr;
   int64_t occ_bytes = 0, tables_bytes = 0, sa_bytes = 0;
   int64_t sar_len = 0, tables_words = 0;
   uint8_t sym_table[256];
   uint8_t com_table[8];
   uint8_t *d_sym_table = nullptr;
   // reference-layout arrays kept on the device by gpumap_build_device (keep_reference != 0)
   void *d_ref_occ = nullptr, *d_text = nullptr;
   int64_t ref_occ_words = 0;
   uint64_t c_host[8] = {0, 0, 0, 0, 0, 0, 0, 0};
   int64_t wil_cnt = 0;
   // nodes whose interval holds one suffix are finished against the text (k_verify) when the text is in
   // HBM (d_text) and the caller wants located hits only
   bool sort_all_bits = false;    // option "sort_all_bits": 1 = the radix sort of the located hits runs over every bit of the keys (no k_order_runs)
   int sort_by_read = -1;         // option "sort_by_read": 1 = the sort runs over the read's bits and the hits of a read are ordered in place, 0 = never either, 1 = by read, 2 = by segments of reads (k_order_segments), -1 = by read where nearly every read has one hit, by segments where they are small
   int seed_candidates = 1;       // option "seed_candidates": a seed whose entry of the jump table holds one suffix goes to k_verify at once
   int verify_tile = 0;           // option "verify_tile": entries of the hit list a block of k_verify takes (0: 512)
   int verify = -1;               // option "verify": 0 off, 1 nodes of one suffix, 2 also of two, -1 by the size of the text
   bool owns_text = true;
   // jump table of the leaf phases (k_build_jump): 4^jump_k entries of 16 bytes, nullptr = off
   void *d_jump = nullptr;
   void *d_jump2 = nullptr;       // a second, shorter table for leaf blocks the first is too long for (gpumap_ann_build: the left
   int jump2_k = 0;               // block of a k-mer is half the k-mer; 4^jump2_k entries)
   void *d_text4 = nullptr;       // the text as bit planes (k_pack_planes), built with the jump table when the text is in HBM
   int jump_k = 0;
   // launch shape
   int cu_count = 256;
   int waves_per_cu = 0;              // 0 = default (12)
   int64_t spill_cap = 16384;         // nodes per wave
   // device address, on this context's GPU, of the static scratch array of gpumap_query_all_noalloc (gm_device.hip)
   void *noalloc_scratch = nullptr;
   // staging of the batched operators (gpumap_occ_all ...), on the context's own stream
   DevBuf op_in, op_out;
   DevBuf d_names, d_name_off;        // the sequences' names (gpumap_set_seq_names): hit lines made on the device
   // options read by gpumap_batch_run / gpumap_batch_fetch (batches own their streams, scratch and result arenas)
   bool pinned_results = false;       // "pinned_results": fetch into page-locked arenas of the batch
   bool compact_results = false;      // "compact_results": sorted 8-byte keys + CSR offsets instead of the structure of arrays
   int64_t hit_guess = 0;             // "hit_list_guess": first guess of the hit list's length in entries (0 = by the batch; tests)
   bool ann_trail_reuse = true;       // "ann_trail_reuse": gpumap_ann_build reproduces the reference's reuse of the last k-mer's search (0: every k-mer afresh)
   int ann_threads = 1;               // "ann_threads": the `threads` of the reference's ann_build whose job list is reproduced
   int64_t jump_budget = 0;           // "jump_bytes": most bytes the jump table may take (0 = a third of the free memory)
   std::vector<int64_t> seq_beg_host; // seq_beg[] (for gpumap_hits_expand)
   hipStream_t pipe_post = nullptr;   // what follows the search of a run (locate, order), beside the next chunk's search
   hipStream_t pipe_run = nullptr, pipe_copy = nullptr, pipe_up = nullptr;   // streams shared by the batches of pipelines (gm_ctx_pipe_streams)
   std::mutex spare_mutex;
   std::vector<gpumap_batch *> spare_batches;     // batch objects of finished pipelines, kept for the next one (gm_ctx_take_batch)
   gpumap_batch *map_batch_obj = nullptr;   // gpumap_map_batch with "pinned_results": the batch whose arenas hold the results
};



// Repack of a reference-layout Occ array that already lies in device memory (d_ref_occ, d_c) into
// the rank blocks and tables of ctx (allocated here); also copies seq_beg (host) behind the tables.
// ctx->ix.n, sar fields and d_sa must be set by the caller.  gm_device.hip.
int gm_ctx_repack(gpumap_ctx *ctx, const uint64_t *d_ref_occ, const uint64_t *d_c, int64_t mark_intv,
                  const int64_t *seq_beg_host, int64_t seq_cnt);
int gm_ctx_common_init(gpumap_ctx *ctx, int device);
// (Re)builds the jump table for K-mers of k letters (k < 0: default for the text length; 0: none).
int gm_ctx_build_jump(gpumap_ctx *ctx, int k);
int gm_ctx_build_jump2(gpumap_ctx *ctx, int k);
