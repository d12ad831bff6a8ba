// gm_device.hip -- device context, batches and the C-ABI entry points that launch kernels.
// Host plumbing only; the kernels are in gm_kernels.hpp and the shared integer logic in gm_core.hpp.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <chrono>
#include <vector>
#include "gm_devctx.hpp"
#include "gm_kernels.hpp"
#include "gm_search.hpp"
#include "gm_ann.hpp"
#include "gm_parse.hpp"

using namespace gm;

// One batch of reads and everything its run needs: a batch has its own stream, counters, spill region and
// sort scratch, so that batches of one context can be filled, run and fetched side by side from different
// host threads (gm_pipe.cpp: the copy of one chunk's results overlaps the search of the next chunk).  The
// context is only read while batches run.
struct gpumap_batch {
   gpumap_ctx *ctx = nullptr;
   // Three streams: uploads (copies of the chunk to the device, k_read_lengths), the run's kernels, the copy of
   // the results to the host.  A batch on its own uses its one stream for all three.  The batches of a pipeline
   // share their context's three streams (gm_batch_set_streams): the kernels of consecutive chunks then run
   // back to back on one stream (two persistent k_search launches side by side only slow each other down) while
   // the result copy of the chunk before and the upload of the chunk after run on the other two.  That is why
   // nothing below waits for a whole stream: every wait is for an event this batch recorded.
   hipStream_t stream = nullptr, st_run = nullptr, st_copy = nullptr, st_up = nullptr;
   // Round 3: what follows the search of a run (locate, order, hit lines) goes to a fourth stream, st_post, and the
   // pipeline lets the next chunk's run begin at that point (on_searched): the streaming kernels of the sort then run
   // beside the next chunk's encoding and search, which wait for memory requests most of the time.
   hipStream_t st_post = nullptr;
   std::function<void()> on_searched;
   // ... and the pipeline holds the NEXT run's k_search back until this run's sort has been enqueued (before_search of the
   // next batch waits for on_ordering of this one): k_search is persistent and takes every CU's LDS; a sort that arrives after
   // it has started finds no room until it ends, and the chunks' results then come back in bursts (17.3 instead of 14.8 ms
   // per chunk; which of the two a run got was a matter of a host round trip taking 0.2 or 1 ms)
   std::function<void()> before_search, on_ordering;
   hipEvent_t ev_sync = nullptr, ev_up = nullptr;
   hipEvent_t ev_drain = nullptr;       // on the post stream, before this run's first kernel there: what the runs before left on that stream is done
   int64_t n_reads = 0;
   int stride = 0, maxlen = 1;
   // the letters (or codes) of the reads in device memory: the library's copy of a host buffer (up_src, up_off;
   // indexed relative to offsets[0]) or the caller's own device buffers
   const uint8_t *d_src = nullptr;
   const int64_t *d_off = nullptr;
   bool src_relative = true, src_text = false, encoded = false, have_bytes = false;
   DevBuf up_src, up_off, seen;
   uint8_t *h_small = nullptr;       // page-locked: small read-backs (lengths seen, counters) and uploads (phase tables)
   hipEvent_t ev_seen = nullptr;
   DevBuf reads, reads4, read_len, len_row, row_len, phases, phase_rec, phase_lo, status;
   DevBuf hits, hit_sizes, hit_off, vsteps, longer, keys, keys_alt, read_cnt, read_first;
   DevBuf loc_read, loc_pos, loc_fwd, loc_seq, loc_seqpos, loc_strand, loc_score;
   DevBuf hkeys, hkeys_alt, hidx, hidx_alt, hit_read, hit_fp, hit_rp, hit_sz, hit_dp, hit_score, hit_align;
   DevBuf spill, counters, cub_tmp;
   DevBuf read_row, sc_bound, score_add, row_leaves, row_state, sc_qfp, sc_qrp;      // blocksc_trail (gm_batch_run with GM_RUN_SC)
   // a chunk that came as a block of the read file (GM_SRC_FILE): the ends of its lines, per read where its letters begin
   // and end in the block and where its tag lies; and the hit lines of a run with GM_RESULT_LINES
   DevBuf nl_pos, tile_cnt, tile_off, raw_beg, raw_end, tag_off, tag_len, nul_flag, line_len, lines;
   const int64_t *d_end = nullptr;   // not nullptr: the letters of read r are d_src[d_off[r], d_end[r])
   bool from_file = false;
   int lines_per_rec = 0;
   int64_t first_line = 1, line_bytes = 0;
   bool have_lines = false;
   DevBuf packed, first32, status8;  // k_pack_results: the packed result
   bool have_packed = false;
   DevBuf seg_first;                 // k_seg_first: where the segments of the sort by segments begin
   DevBuf big_start, big_cnt;        // k_order_first: the runs of keys it left to k_order_big
   DevBuf sc_flags, sc_alias;        // gpumap_ann_build with the reference's reuse of searches (sc_plan)
   struct AnnPlanHost *sc_plan = nullptr;     // not owned; set by gpumap_ann_build around its runs
   int extra_factor = 1;             // the per-query arrays hold room for extra_factor * n_reads queries (2 with sc_plan)
   DevBuf stage, stage2[2], iv_idx, iv_dense, blk_cnt, blk_off, blk_cnt2[2], blk_off2[2];      // what the blocks of k_verify / k_verify_group leave (keys, lists) and the scans of their counts
   DevBuf big_idx, big_dense, wide_idx, wide_dense, vsteps2[2];   // candidates of three and four suffixes (k_verify_group)
   const unsigned long long *sorted_keys = nullptr;     // the located hits' keys after the sort (in keys or keys_alt)
   int key_pos_bits = 0;
   bool compact = false;             // the last run left keys + CSR offsets only (no structure of arrays)
   bool length_seen[MAX_READ + 1];
   int64_t n_hits = 0, n_real = 0, n_locs = 0, n_ext = 0;
   int64_t stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};
   int64_t n_big = 0;                // candidates of three and four suffixes in the last run
   int64_t verify_windows = 0;       // suffixes k_verify replayed on the text in the last run
   int64_t spill_chunks = 0;         // chunks of SPILL_CHUNK nodes the deques of the last search moved to global memory
   int attempts = 0;                 // searches the last run took (2: the first guess of the hit list was too short)
   bool have_intervals = false, ran = false;
   int tau = 0;
   float ms[4] = {0, 0, 0, 0};
   float ms_encode = 0;
   hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};      // [4]: between k_search and k_verify, [5]: before the encoding
   // option "pinned_results": results are fetched into page-locked arenas owned by the batch, used in turn
   void *pin_arena[2] = {nullptr, nullptr};
   size_t pin_cap[2] = {0, 0};
   int pin_turn = 0;
   void release()
   {
      for (DevBuf *b : {&up_src, &up_off, &seen, &reads, &reads4, &read_len, &len_row, &row_len, &phases, &phase_rec, &phase_lo, &status, &hits, &hit_sizes, &hit_off, &vsteps, &longer, &keys, &keys_alt,
                        &read_cnt, &read_first, &loc_read, &loc_pos, &loc_fwd, &loc_seq, &loc_seqpos, &loc_strand,
                        &loc_score, &hkeys, &hkeys_alt, &hidx, &hidx_alt, &hit_read, &hit_fp, &hit_rp, &hit_sz,
                        &hit_dp, &hit_score, &hit_align, &spill, &counters, &cub_tmp, &read_row, &sc_bound, &score_add, &row_leaves, &row_state, &sc_qfp, &sc_qrp, &sc_flags, &sc_alias, &big_start, &big_cnt, &nl_pos, &tile_cnt, &tile_off, &raw_beg, &raw_end, &tag_off, &tag_len, &nul_flag, &line_len, &lines,
                        &big_idx, &big_dense, &wide_idx, &wide_dense, &vsteps2[0], &vsteps2[1], &stage, &stage2[0], &stage2[1], &iv_idx, &iv_dense, &blk_cnt, &blk_off,
                        &blk_cnt2[0], &blk_cnt2[1], &blk_off2[0], &blk_off2[1]})
         b->release();
      for (auto &e : ev) if (e) { (void)hipEventDestroy(e); e = nullptr; }
      for (hipEvent_t *e : {&ev_seen, &ev_sync, &ev_up, &ev_drain}) if (*e) { (void)hipEventDestroy(*e); *e = nullptr; }
      for (int k = 0; k < 2; k++) if (pin_arena[k]) { gm_pinned_give(pin_arena[k], pin_cap[k]); pin_arena[k] = nullptr; }
      pin_cap[0] = pin_cap[1] = 0;
      if (h_small) { (void)hipHostFree(h_small); h_small = nullptr; }
      if (stream) { (void)hipStreamDestroy(stream); stream = nullptr; }
   }
};
// What gpumap_ann_build hands to gm_batch_run for the plan of gm_ann.hpp (the reference's reuse of searches).
struct AnnPlanHost {
   const int64_t *q_sa = nullptr;    // device: suffix-array entries of all k-mers searched
   int64_t c0 = 0;                   // first k-mer of the chunk in the batch
   int kmer = 0, depth = 1;
   int64_t n_extra = 0;              // out: queries the plan added to the last chunk
   DevBuf s1, s2a, s2b, carry, n_extra_dev, tmp;
   void release() { for (DevBuf *b : {&s1, &s2a, &s2b, &carry, &n_extra_dev, &tmp}) b->release(); }
};
constexpr int CNT_WORDS = 16;       // [0] cursor [1] hits [2] n_ext [3] error [4..8] stats of k_search (SearchArgs::stats)
constexpr int CNT_SUMS = 192;       // behind them: 64 partial sums of k_verify, 64 + 64 of the two launches of k_verify_group; then single words
constexpr size_t HOST_CNT_BYTES = 2048;   // the counters as read back: CNT_WORDS + 64 + 64 words, the located total, the count of big candidates
constexpr size_t H_SMALL_BYTES = 1 << 17;      // lengths seen | phase tables of up to 256 read lengths | counters

// ------------------------------------------------------------------------------------------------
// Context
// ------------------------------------------------------------------------------------------------

// ------------------------------------------------------------------------------------------------
// Scratch of the allocation-free entry point (gpumap_query_all_noalloc): staging of the intervals and of
// their extensions.  The array lives in the code object, every GPU holds its own copy at its own
// address; `scratch` below is only the host-side stand-in of the symbol, the device address is looked
// up once per device (gm_scratch_address) and only that pointer goes to copies and kernels.
// ------------------------------------------------------------------------------------------------
constexpr size_t NOALLOC_IN_BYTES = 24 * (size_t)GPUMAP_NOALLOC_MAX_QUERIES;       // fp, rp, sz
constexpr size_t NOALLOC_OUT_BYTES = 120 * (size_t)GPUMAP_NOALLOC_MAX_QUERIES;     // 5 x {fp, rp, sz}
static __device__ __align__(256) unsigned char scratch[NOALLOC_IN_BYTES + NOALLOC_OUT_BYTES];   // 2.25 MiB per GPU
constexpr int SCRATCH_MAX_DEVICES = 64;
static void *g_scratch_ptr[SCRATCH_MAX_DEVICES];
static std::mutex g_scratch_mutex;

// Device address of `scratch` on `device`, which must be the current device.  nullptr on failure.
static void *gm_scratch_address(int device)
{
   if (device < 0 || device >= SCRATCH_MAX_DEVICES) return nullptr;
   std::lock_guard<std::mutex> lock(g_scratch_mutex);
   if (g_scratch_ptr[device] == nullptr) {
      void *p = nullptr;
      const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
      if (e != hipSuccess) {
         std::fprintf(stderr, "%s\t(%s:%d) error: hipGetSymbolAddress: %s\n", __func__, __FILE__, __LINE__, hipGetErrorString(e));
         return nullptr;
      }
      g_scratch_ptr[device] = p;
   }
   return g_scratch_ptr[device];
}

static void ctx_free_spare_batches(gpumap_ctx *ctx);

int gm_ctx_common_init(gpumap_ctx *ctx, int device)
{
   hipDeviceProp_t prop;
   HIP_TRY(hipSetDevice(device));
   ctx->noalloc_scratch = gm_scratch_address(device);
   GM_FAIL_IF(ctx->noalloc_scratch == nullptr, "cannot resolve the device address of the static scratch array.");
   HIP_TRY(hipGetDeviceProperties(&prop, device));
   ctx->device = device;
   ctx->cu_count = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
   HIP_TRY(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
   HIP_TRY(hipMalloc((void **)&ctx->d_sym_table, 256));
   // (Every upload and fill of the set-up goes on the stream of the kernels that use it, and the host waits for that stream.
   // The runtime's synchronous calls work on the null stream, which a non-blocking stream is not ordered with: a fill
   // that returns before it is done (hipMemset does) then lands on top of what a kernel on the context's stream has
   // written since.  Seen under rocprofv3's counter collection, where every operation takes its time: the second
   // context of a process searched with a zeroed superblock table.)
   HIP_TRY(hipMemcpyAsync(ctx->d_sym_table, ctx->sym_table, 256, hipMemcpyHostToDevice, ctx->stream));
   HIP_TRY(hipStreamSynchronize(ctx->stream));
   return 0;
failure_return:
   return -1;
}

extern "C" void gpumap_destroy(gpumap_ctx *ctx)
{
   if (!ctx) return;
   (void)hipSetDevice(ctx->device);
   if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
   if (ctx->map_batch_obj) { gpumap_batch_free(ctx->map_batch_obj); ctx->map_batch_obj = nullptr; }
   ctx_free_spare_batches(ctx);
   for (hipStream_t *st : {&ctx->pipe_run, &ctx->pipe_copy, &ctx->pipe_up, &ctx->pipe_post}) if (*st) { (void)hipStreamSynchronize(*st); (void)hipStreamDestroy(*st); *st = nullptr; }
   if (ctx->owns_index) {
      if (ctx->d_occ) (void)hipFree(ctx->d_occ);
      if (ctx->d_tables) (void)hipFree(ctx->d_tables);
      if (ctx->d_sa) (void)hipFree(ctx->d_sa);
   }
   if (ctx->d_jump) (void)hipFree(ctx->d_jump);
   if (ctx->d_jump2) (void)hipFree(ctx->d_jump2);
   if (ctx->d_text4) (void)hipFree(ctx->d_text4);
   if (ctx->d_ref_occ) (void)hipFree(ctx->d_ref_occ);
   if (ctx->d_text && ctx->owns_text) (void)hipFree(ctx->d_text);
   if (ctx->d_sym_table) (void)hipFree(ctx->d_sym_table);
   for (DevBuf *b : {&ctx->op_in, &ctx->op_out, &ctx->d_names, &ctx->d_name_off}) b->release();
   if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
   delete ctx;
}

int gm_ctx_build_jump(gpumap_ctx *ctx, int k)
{
   if (ctx->d_jump) { HIP_TRY(hipFree(ctx->d_jump)); ctx->d_jump = nullptr; ctx->jump_k = 0; }
   if (ctx->d_text != nullptr && ctx->d_text4 == nullptr) {
      // what k_verify compares windows of: bit planes, 12 bytes per 32 symbols, 512 symbols of separators behind the end
      const int64_t units = cdiv(ctx->ix.n, 128) + 4, groups = 4 * units;
      HIP_TRY(hipMalloc(&ctx->d_text4, 48 * (size_t)units));
      hipLaunchKernelGGL(k_pack_planes, dim3((unsigned)std::min<int64_t>(cdiv(groups, 256), 1 << 20)), dim3(256), 0, ctx->stream,
                         (const uint8_t *)ctx->d_text, ctx->ix.n, (uint8_t)SYMS, (uint32_t *)ctx->d_text4, groups);
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipStreamSynchronize(ctx->stream));
   }
   if (k < 0) {
      // as many letters as keep the table at no more than one entry per text position, at most 16 (69 GB),
      // and the table within a third of the memory that is free, or within the byte budget the host
      // application gave (option "jump_bytes" / GPUMAP_JUMP_BYTES)
      size_t free_b = 0, total_b = 0;
      HIP_TRY(hipMemGetInfo(&free_b, &total_b));
      if (ctx->jump_budget == 0 && getenv("GPUMAP_JUMP_BYTES")) ctx->jump_budget = atoll(getenv("GPUMAP_JUMP_BYTES"));
      const size_t budget = ctx->jump_budget > 0 ? std::min<size_t>((size_t)ctx->jump_budget, free_b) : free_b / 3;
      k = 0;
      while (k < 16 && ((int64_t)1 << (2 * (k + 1))) <= ctx->ix.n && ((size_t)16 << (2 * (k + 1))) <= budget) k++;
      if (k < 4) k = 0;
   }
   GM_FAIL_IF(k > 16, "jump_k out of range.");
   if (k == 0) return 0;
   {
      const int64_t entries = (int64_t)1 << (2 * k);
      HIP_TRY(hipMalloc(&ctx->d_jump, 16 * (size_t)entries));
      const int64_t per_launch = (int64_t)1 << 30;                      // a launch holds fewer than 2^32 threads
      for (int64_t first = 0; first < entries; first += per_launch) {
         const int64_t cnt = std::min(per_launch, entries - first);
         hipLaunchKernelGGL(k_build_jump, dim3((unsigned)cdiv(cnt, 256)), dim3(256), 0, ctx->stream, ctx->ix, k, first, (uint4 *)ctx->d_jump);
         HIP_TRY(hipGetLastError());
      }
      HIP_TRY(hipStreamSynchronize(ctx->stream));
      ctx->jump_k = k;
   }
   return 0;
failure_return:
   return -1;
}

// The second jump table (k letters, 0 = none): for searches whose leaf blocks are shorter than the first table's
// letters (gpumap_ann_build).  Kept until the context goes or another length is asked for.
int gm_ctx_build_jump2(gpumap_ctx *ctx, int k)
{
   if (k == ctx->jump2_k) return 0;
   if (ctx->d_jump2) { HIP_TRY(hipFree(ctx->d_jump2)); ctx->d_jump2 = nullptr; ctx->jump2_k = 0; }
   if (k <= 0) return 0;
   GM_FAIL_IF(k > 13, "jump2_k out of range.");
   {
      const int64_t entries = (int64_t)1 << (2 * k);
      HIP_TRY(hipMalloc(&ctx->d_jump2, 16 * (size_t)entries));
      hipLaunchKernelGGL(k_build_jump, dim3((unsigned)cdiv(entries, 256)), dim3(256), 0, ctx->stream, ctx->ix, k, (int64_t)0, (uint4 *)ctx->d_jump2);
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipStreamSynchronize(ctx->stream));
      ctx->jump2_k = k;
   }
   return 0;
failure_return:
   return -1;
}

int gm_ctx_repack(gpumap_ctx *ctx, const uint64_t *d_ref_occ, const uint64_t *d_c, int64_t mark_intv,
                  const int64_t *seq_beg_host, int64_t seq_cnt)
{
   IndexDev &ix = ctx->ix;
   GM_FAIL_IF(num_supers(ix.n) > MAX_SB_LDS, "text too long for the staged superblock table.");
   ix.n_blocks = num_blocks(ix.n);
   ix.n_super = num_supers(ix.n);
   ix.seq_cnt = (int32_t)seq_cnt;
   ctx->seq_beg_host.assign(seq_beg_host, seq_beg_host + seq_cnt);
   ctx->occ_bytes = ix.n_blocks * 64;
   ctx->tables_words = TAB_SBASE + SYMS * ix.n_super + seq_cnt;
   ctx->tables_bytes = ctx->tables_words * 8;
   HIP_TRY(hipMalloc(&ctx->d_occ, (size_t)ctx->occ_bytes));
   HIP_TRY(hipMalloc(&ctx->d_tables, (size_t)ctx->tables_bytes));
   HIP_TRY(hipMemsetAsync(ctx->d_tables, 0, (size_t)ctx->tables_bytes, ctx->stream));
   {
      int64_t n_ref_words = cdiv(ix.n, mark_intv * 64) * mark_intv;
      int64_t g1 = cdiv(std::max<int64_t>(ix.n_super, SYMS), 64);
      hipLaunchKernelGGL(k_super_counts, dim3((unsigned)g1), dim3(64), 0, ctx->stream, d_ref_occ,
                         mark_intv, ix.n, ix.n_super, d_c, (uint64_t *)ctx->d_tables);
      PackArgs P;
      P.ref_occ = d_ref_occ; P.tables = (const uint64_t *)ctx->d_tables; P.blocks = (uint32_t *)ctx->d_occ;
      P.intv = mark_intv; P.n_ref_words = n_ref_words; P.n_blocks = ix.n_blocks;
      hipLaunchKernelGGL(k_pack_blocks, dim3((unsigned)cdiv(ix.n_blocks, 256)), dim3(256), 0, ctx->stream, P);
   }
   HIP_TRY(hipGetLastError());
   // seq_beg sits behind the superblock table
   HIP_TRY(hipMemcpyAsync((uint64_t *)ctx->d_tables + TAB_SBASE + SYMS * ix.n_super, seq_beg_host,
                          8 * (size_t)seq_cnt, hipMemcpyHostToDevice, ctx->stream));
   HIP_TRY(hipStreamSynchronize(ctx->stream));
   ix.occ = (const uint32_t *)ctx->d_occ;
   ix.tables = (const uint64_t *)ctx->d_tables;
   ix.sa = (const uint64_t *)ctx->d_sa;
   ix.seq_beg = (const int64_t *)((const uint64_t *)ctx->d_tables + TAB_SBASE + SYMS * ix.n_super);
   return gm_ctx_build_jump(ctx, getenv("GPUMAP_JUMP_K") ? atoi(getenv("GPUMAP_JUMP_K")) : -1);
failure_return:
   return -1;
}

extern "C" gpumap_ctx *gpumap_create(const gpumap_index_view *v, int device)
{
   gpumap_ctx *ctx = nullptr;
   void *d_ref = nullptr, *d_c = nullptr, *d_seq = nullptr;
   int count = 0;
   GM_FAIL_IF(v == nullptr, "argument 'view' is NULL.");
   GM_FAIL_IF(v->occ == nullptr || v->c == nullptr || v->sa == nullptr, "index view has NULL arrays.");
   GM_FAIL_IF(v->sym_count != SYMS, "only the 5-symbol DNA alphabet is supported.");
   GM_FAIL_IF(v->sym_table == nullptr || v->com_table == nullptr, "index view has no symbol tables.");
   GM_FAIL_IF(v->occ_word_size != 64 || v->occ_mark_intv < 1, "unsupported Occ layout.");
   // 2^33: the superblock counts of up to 256 superblocks are staged in LDS, and the Occ blocks of such a text
   // stay below 4 GiB, which k_search addresses with 32-bit offsets
   GM_FAIL_IF(v->txt_len < 2 || v->txt_len >= ((int64_t)1 << 33), "text length out of range (2 <= txt_len < 2^33).");
   GM_FAIL_IF(v->sar_bits < 1 || v->sar_bits > 40, "sar_bits out of range.");
   GM_FAIL_IF(v->seq_cnt < 1 || v->seq_beg == nullptr, "index view has no sequences.");
   {
      // the Occ array must be exactly as long as bwt_build_opt makes it (src/index_bwt.c:501-507)
      int64_t mark_bits = v->occ_mark_intv * 64;
      int64_t n_intv = cdiv(v->txt_len, mark_bits);
      GM_FAIL_IF(v->occ_length != (n_intv * v->occ_mark_intv + n_intv + 1) * SYMS, "occ_length does not match txt_len.");
      GM_FAIL_IF(cdiv(v->txt_len * v->sar_bits, 64) > v->sar_len, "sar_len too small for txt_len.");
   }
   HIP_TRY(hipGetDeviceCount(&count));
   GM_FAIL_IF(device < 0 || device >= count, "no such HIP device.");
   ctx = new gpumap_ctx();
   memcpy(ctx->sym_table, v->sym_table, 256);
   memcpy(ctx->com_table, v->com_table, SYMS + 1);
   GM_FAIL_IF(gm_ctx_common_init(ctx, device) != 0, "cannot initialise the device context.");
   {
      IndexDev &ix = ctx->ix;
      ix.n = v->txt_len;
      ix.sar_bits = (int32_t)v->sar_bits;
      ctx->sar_len = v->sar_len;
      ctx->sa_bytes = (v->sar_len + 1) * 8;           // one spare word: an entry read may touch word+1
      HIP_TRY(hipMalloc(&ctx->d_sa, (size_t)ctx->sa_bytes));
      HIP_TRY(hipMemsetAsync(ctx->d_sa, 0, (size_t)ctx->sa_bytes, ctx->stream));
      HIP_TRY(hipMemcpyAsync(ctx->d_sa, v->sa, (size_t)v->sar_len * 8, hipMemcpyHostToDevice, ctx->stream));
      if (v->text) {        // the text lets k_verify finish unique nodes; without it the search runs on the Occ table alone
         HIP_TRY(hipMalloc(&ctx->d_text, (size_t)v->txt_len + 16));          // k_verify reads whole words
         HIP_TRY(hipMemsetAsync((char *)ctx->d_text + v->txt_len, SYMS, 16, ctx->stream));
         HIP_TRY(hipMemcpyAsync(ctx->d_text, v->text, (size_t)v->txt_len, hipMemcpyHostToDevice, ctx->stream));
      }
      // K0: upload the reference layout verbatim, repack on the device, drop the copy
      HIP_TRY(hipMalloc(&d_ref, (size_t)v->occ_length * 8));
      HIP_TRY(hipMemcpyAsync(d_ref, v->occ, (size_t)v->occ_length * 8, hipMemcpyHostToDevice, ctx->stream));
      HIP_TRY(hipMalloc(&d_c, 8 * (SYMS + 1)));
      HIP_TRY(hipMemcpyAsync(d_c, v->c, 8 * (SYMS + 1), hipMemcpyHostToDevice, ctx->stream));
      HIP_TRY(hipStreamSynchronize(ctx->stream));
      GM_FAIL_IF(gm_ctx_repack(ctx, (const uint64_t *)d_ref, (const uint64_t *)d_c, v->occ_mark_intv, v->seq_beg, v->seq_cnt) != 0,
                 "cannot repack the Occ table.");
      HIP_TRY(hipFree(d_ref)); d_ref = nullptr;
      HIP_TRY(hipFree(d_c)); d_c = nullptr;
      memcpy(ctx->c_host, v->c, 8 * (SYMS + 1));
      ctx->wil_cnt = v->wil_cnt;
   }
   (void)d_seq;
   return ctx;
failure_return:
   if (d_ref) (void)hipFree(d_ref);
   if (d_c) (void)hipFree(d_c);
   gpumap_destroy(ctx);
   return nullptr;
}

extern "C" gpumap_ctx *gpumap_create_from_files(const char *filename_base, int device)
{
   gpumap_index *ix = gpumap_index_read(filename_base);
   if (!ix) return nullptr;
   gpumap_index_view v;
   gpumap_index_view_of(ix, &v);
   gpumap_ctx *ctx = gpumap_create(&v, device);
   gpumap_index_free(ix);
   return ctx;
}

extern "C" size_t gpumap_devindex_header_size(void) { return sizeof(DevIndexHeader); }

extern "C" int gpumap_devindex_export(const gpumap_ctx *ctx, gpumap_devindex *out, void *header)
{
   GM_FAIL_IF(ctx == nullptr || out == nullptr || header == nullptr, "NULL argument.");
   {
      out->occ_blocks = ctx->d_occ; out->occ_bytes = ctx->occ_bytes;
      out->sa_words = ctx->d_sa; out->sa_bytes = ctx->sa_bytes;
      out->tables = ctx->d_tables; out->tables_bytes = ctx->tables_bytes;
      out->text = ctx->d_text; out->text_bytes = ctx->d_text ? ctx->ix.n + 16 : 0;
      DevIndexHeader h;
      memset(&h, 0, sizeof h);
      h.magic = DEVINDEX_MAGIC;
      h.n = ctx->ix.n; h.n_blocks = ctx->ix.n_blocks; h.n_super = ctx->ix.n_super;
      h.sar_bits = ctx->ix.sar_bits; h.sar_len = ctx->sar_len; h.seq_cnt = ctx->ix.seq_cnt;
      h.tables_words = ctx->tables_words;
      memcpy(h.sym_table, ctx->sym_table, 256);
      memcpy(h.com_table, ctx->com_table, 8);
      memcpy(header, &h, sizeof h);
   }
   return 0;
failure_return:
   return -1;
}

extern "C" gpumap_ctx *gpumap_create_from_device(const gpumap_devindex *dev, const void *header, int device)
{
   gpumap_ctx *ctx = nullptr;
   DevIndexHeader h;
   int count = 0;
   GM_FAIL_IF(dev == nullptr || header == nullptr, "NULL argument.");
   memcpy(&h, header, sizeof h);
   GM_FAIL_IF(h.magic != DEVINDEX_MAGIC, "bad device index header.");
   GM_FAIL_IF(h.n < 2 || h.n_blocks != num_blocks(h.n) || h.n_super != num_supers(h.n), "inconsistent device index header.");
   GM_FAIL_IF(dev->occ_bytes < h.n_blocks * 64 || dev->sa_bytes < (h.sar_len + 1) * 8 ||
              dev->tables_bytes < h.tables_words * 8, "device index buffers are too small.");
   GM_FAIL_IF(dev->occ_blocks == nullptr || dev->sa_words == nullptr || dev->tables == nullptr, "NULL device buffer.");
   HIP_TRY(hipGetDeviceCount(&count));
   GM_FAIL_IF(device < 0 || device >= count, "no such HIP device.");
   ctx = new gpumap_ctx();
   ctx->owns_index = false;
   memcpy(ctx->sym_table, h.sym_table, 256);
   memcpy(ctx->com_table, h.com_table, 8);
   GM_FAIL_IF(gm_ctx_common_init(ctx, device) != 0, "cannot initialise the device context.");
   ctx->d_occ = dev->occ_blocks; ctx->d_sa = dev->sa_words; ctx->d_tables = dev->tables;
   if (dev->text != nullptr && dev->text_bytes >= h.n + 16) { ctx->d_text = dev->text; ctx->owns_text = false; }
   ctx->occ_bytes = dev->occ_bytes; ctx->sa_bytes = dev->sa_bytes; ctx->tables_bytes = dev->tables_bytes;
   ctx->sar_len = h.sar_len; ctx->tables_words = h.tables_words;
   ctx->ix.n = h.n; ctx->ix.n_blocks = h.n_blocks; ctx->ix.n_super = h.n_super;
   ctx->ix.sar_bits = (int32_t)h.sar_bits; ctx->ix.seq_cnt = (int32_t)h.seq_cnt;
   ctx->ix.occ = (const uint32_t *)ctx->d_occ;
   ctx->ix.tables = (const uint64_t *)ctx->d_tables;
   ctx->ix.sa = (const uint64_t *)ctx->d_sa;
   ctx->ix.seq_beg = (const int64_t *)((const uint64_t *)ctx->d_tables + TAB_SBASE + SYMS * h.n_super);
   ctx->seq_beg_host.resize((size_t)h.seq_cnt);
   HIP_TRY(hipMemcpyAsync(ctx->seq_beg_host.data(), ctx->ix.seq_beg, 8 * (size_t)h.seq_cnt, hipMemcpyDeviceToHost, ctx->stream));
   HIP_TRY(hipStreamSynchronize(ctx->stream));
   GM_FAIL_IF(gm_ctx_build_jump(ctx, getenv("GPUMAP_JUMP_K") ? atoi(getenv("GPUMAP_JUMP_K")) : -1) != 0, "cannot build the jump table.");
   return ctx;
failure_return:
   gpumap_destroy(ctx);
   return nullptr;
}

extern "C" int gpumap_set_option(gpumap_ctx *ctx, const char *name, int64_t value)
{
   GM_FAIL_IF(ctx == nullptr || name == nullptr, "NULL argument.");
   if (!strcmp(name, "waves_per_cu")) { GM_FAIL_IF(value < 0 || value > 32, "waves_per_cu out of range."); ctx->waves_per_cu = (int)value; }
   else if (!strcmp(name, "pinned_results")) ctx->pinned_results = value != 0;
   else if (!strcmp(name, "compact_results")) ctx->compact_results = value != 0;
   else if (!strcmp(name, "hit_list_guess")) { GM_FAIL_IF(value < 0, "hit_list_guess out of range."); ctx->hit_guess = value; }
   else if (!strcmp(name, "jump_bytes")) { GM_FAIL_IF(value < 0, "jump_bytes out of range."); ctx->jump_budget = value; HIP_TRY(hipSetDevice(ctx->device)); GM_FAIL_IF(gm_ctx_build_jump(ctx, -1) != 0, "cannot build the jump table."); }
   else if (!strcmp(name, "verify")) { GM_FAIL_IF(value != -1 && value != 0 && value != 1 && value != 2 && value != 4 && value != 8 && value != 16, "verify out of range (-1, 0, 1, 2, 4, 8, 16)."); ctx->verify = (int)value; }
   else if (!strcmp(name, "sort_all_bits")) ctx->sort_all_bits = value != 0;
   else if (!strcmp(name, "sort_by_read")) { GM_FAIL_IF(value < -1 || value > 2, "sort_by_read out of range (-1, 0, 1, 2)."); ctx->sort_by_read = (int)value; }
   else if (!strcmp(name, "seed_candidates")) { GM_FAIL_IF(value < 0 || value > 1, "seed_candidates out of range (0, 1)."); ctx->seed_candidates = (int)value; }
   else if (!strcmp(name, "verify_tile")) { GM_FAIL_IF(value < 0 || value > VERIFY_TILE_MAX, "verify_tile out of range (0 = default, at most 1024)."); ctx->verify_tile = (int)value; }
   else if (!strcmp(name, "trim")) { HIP_TRY(hipSetDevice(ctx->device)); ctx_free_spare_batches(ctx); }
   else if (!strcmp(name, "ann_trail_reuse")) ctx->ann_trail_reuse = value != 0;
   else if (!strcmp(name, "ann_threads")) { GM_FAIL_IF(value < 1, "argument 'threads' must be greater than 0."); ctx->ann_threads = (int)value; }
   else if (!strcmp(name, "jump_k")) { HIP_TRY(hipSetDevice(ctx->device)); GM_FAIL_IF(gm_ctx_build_jump(ctx, (int)value) != 0, "cannot build the jump table."); }
   else if (!strcmp(name, "spill_nodes")) { GM_FAIL_IF(value < 0 || value > (1 << 22), "spill_nodes out of range."); ctx->spill_cap = value ? ((value + SPILL_CHUNK - 1) / SPILL_CHUNK) * SPILL_CHUNK : 16384; }   // (batches size their spill regions at their next run)
   else GM_FAIL("unknown option.");
   return 0;
failure_return:
   return -1;
}

extern "C" const char *gpumap_version(void) { return "gpumap 0.1 (gfx950)"; }

// ------------------------------------------------------------------------------------------------
// Batched operators
// ------------------------------------------------------------------------------------------------

extern "C" int gpumap_occ_all(gpumap_ctx *ctx, int64_t n, const int64_t *ptr, int64_t *occ)
{
   GM_FAIL_IF(ctx == nullptr, "argument 'ctx' is NULL.");
   GM_FAIL_IF(n < 0 || (n > 0 && (ptr == nullptr || occ == nullptr)), "invalid arguments.");
   if (n == 0) return 0;
   HIP_TRY(hipSetDevice(ctx->device));
   HIP_TRY(ctx->op_in.ensure(8 * (size_t)n));
   HIP_TRY(ctx->op_out.ensure(8 * SYMS * (size_t)n));
   HIP_TRY(hipMemcpyAsync(ctx->op_in.p, ptr, 8 * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
   hipLaunchKernelGGL(k_occ_all, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, ctx->stream, ctx->ix, n,
                      ctx->op_in.as<const int64_t>(), ctx->op_out.as<int64_t>());
   HIP_TRY(hipGetLastError());
   HIP_TRY(hipMemcpyAsync(occ, ctx->op_out.p, 8 * SYMS * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
   HIP_TRY(hipStreamSynchronize(ctx->stream));
   return 0;
failure_return:
   return -1;
}

extern "C" int gpumap_query_all(gpumap_ctx *ctx, int end, int64_t n, const int64_t *fp, const int64_t *rp,
                                const int64_t *sz, int64_t *out)
{
   GM_FAIL_IF(ctx == nullptr, "argument 'ctx' is NULL.");
   GM_FAIL_IF(end != GPUMAP_QUERY_SUFFIX && end != GPUMAP_QUERY_PREFIX, "invalid 'end' flag.");
   GM_FAIL_IF(n < 0 || (n > 0 && (!fp || !rp || !sz || !out)), "invalid arguments.");
   if (n == 0) return 0;
   HIP_TRY(hipSetDevice(ctx->device));
   HIP_TRY(ctx->op_in.ensure(24 * (size_t)n));
   HIP_TRY(ctx->op_out.ensure(120 * (size_t)n));
   {
      int64_t *d = ctx->op_in.as<int64_t>();
      HIP_TRY(hipMemcpyAsync(d, fp, 8 * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
      HIP_TRY(hipMemcpyAsync(d + n, rp, 8 * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
      HIP_TRY(hipMemcpyAsync(d + 2 * n, sz, 8 * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
      hipLaunchKernelGGL(k_query_all, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, ctx->stream, ctx->ix, end, n,
                         (const int64_t *)d, (const int64_t *)(d + n), (const int64_t *)(d + 2 * n), ctx->op_out.as<int64_t>());
   }
   HIP_TRY(hipGetLastError());
   HIP_TRY(hipMemcpyAsync(out, ctx->op_out.p, 120 * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
   HIP_TRY(hipStreamSynchronize(ctx->stream));
   return 0;
failure_return:
   return -1;
}

extern "C" int gpumap_prefix_all(gpumap_ctx *ctx, int64_t n, const int64_t *fp, const int64_t *sz, int64_t *out)
{
   GM_FAIL_IF(ctx == nullptr, "argument 'ctx' is NULL.");
   GM_FAIL_IF(n < 0 || (n > 0 && (!fp || !sz || !out)), "invalid arguments.");
   if (n == 0) return 0;
   HIP_TRY(hipSetDevice(ctx->device));
   HIP_TRY(ctx->op_in.ensure(16 * (size_t)n));
   HIP_TRY(ctx->op_out.ensure(16 * SYMS * (size_t)n));
   {
      int64_t *d = ctx->op_in.as<int64_t>();
      HIP_TRY(hipMemcpyAsync(d, fp, 8 * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
      HIP_TRY(hipMemcpyAsync(d + n, sz, 8 * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
      hipLaunchKernelGGL(k_prefix_all, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, ctx->stream, ctx->ix, n,
                         (const int64_t *)d, (const int64_t *)(d + n), ctx->op_out.as<int64_t>());
   }
   HIP_TRY(hipGetLastError());
   HIP_TRY(hipMemcpyAsync(out, ctx->op_out.p, 16 * SYMS * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
   HIP_TRY(hipStreamSynchronize(ctx->stream));
   return 0;
failure_return:
   return -1;
}

// gpumap_query_all for callers that cannot allocate: the staging of gpumap_query_all (op_in / op_out, grown with
// hipMalloc) is the static `scratch` array, through the device address looked up at context creation.  No
// hipMalloc / hipFree here.  Nothing in the scratch has to start at zero: k_query_all writes all 15 words of
// every one of the n outputs that are copied back.  Calls that overlap on one GPU share the array and race.
extern "C" int gpumap_query_all_noalloc(gpumap_ctx *ctx, int end, int64_t n, const int64_t *fp, const int64_t *rp,
                                        const int64_t *sz, int64_t *out)
{
   GM_FAIL_IF(ctx == nullptr, "argument 'ctx' is NULL.");
   GM_FAIL_IF(end != GPUMAP_QUERY_SUFFIX && end != GPUMAP_QUERY_PREFIX, "invalid 'end' flag.");
   GM_FAIL_IF(n < 0 || (n > 0 && (!fp || !rp || !sz || !out)), "invalid arguments.");
   GM_FAIL_IF(n > GPUMAP_NOALLOC_MAX_QUERIES, "more than GPUMAP_NOALLOC_MAX_QUERIES intervals in one allocation-free call.");
   GM_FAIL_IF(ctx->noalloc_scratch == nullptr, "the context has no static scratch.");
   if (n == 0) return 0;
   HIP_TRY(hipSetDevice(ctx->device));
   {
      int64_t *d = reinterpret_cast<int64_t *>(ctx->noalloc_scratch);
      int64_t *d_out = reinterpret_cast<int64_t *>(static_cast<unsigned char *>(ctx->noalloc_scratch) + NOALLOC_IN_BYTES);
      HIP_TRY(hipMemcpyAsync(d, fp, 8 * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
      HIP_TRY(hipMemcpyAsync(d + n, rp, 8 * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
      HIP_TRY(hipMemcpyAsync(d + 2 * n, sz, 8 * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
      hipLaunchKernelGGL(k_query_all, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, ctx->stream, ctx->ix, end, n,
                         (const int64_t *)d, (const int64_t *)(d + n), (const int64_t *)(d + 2 * n), d_out);
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipMemcpyAsync(out, d_out, 120 * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
      HIP_TRY(hipStreamSynchronize(ctx->stream));
   }
   return 0;
failure_return:
   return -1;
}

extern "C" int gpumap_sar_get(gpumap_ctx *ctx, int64_t n, const int64_t *pos, int64_t *out)
{
   GM_FAIL_IF(ctx == nullptr, "argument 'ctx' is NULL.");
   GM_FAIL_IF(n < 0 || (n > 0 && (!pos || !out)), "invalid arguments.");
   if (n == 0) return 0;
   HIP_TRY(hipSetDevice(ctx->device));
   HIP_TRY(ctx->op_in.ensure(8 * (size_t)n));
   HIP_TRY(ctx->op_out.ensure(8 * (size_t)n));
   HIP_TRY(hipMemcpyAsync(ctx->op_in.p, pos, 8 * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
   hipLaunchKernelGGL(k_sar_get, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, ctx->stream, ctx->ix, n,
                      ctx->op_in.as<const int64_t>(), ctx->op_out.as<int64_t>());
   HIP_TRY(hipGetLastError());
   HIP_TRY(hipMemcpyAsync(out, ctx->op_out.p, 8 * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
   HIP_TRY(hipStreamSynchronize(ctx->stream));
   return 0;
failure_return:
   return -1;
}

extern "C" int gpumap_sar_get_range(gpumap_ctx *ctx, int64_t beg, int64_t size, int64_t *vec)
{
   GM_FAIL_IF(ctx == nullptr, "argument 'ctx' is NULL.");
   GM_FAIL_IF(vec == nullptr, "argument 'vec' is NULL.");
   GM_FAIL_IF(beg < 0, "'beg' index must be positive.");
   GM_FAIL_IF(size < 1, "'size' must be greater than 0.");
   GM_FAIL_IF(beg + size > ctx->ix.n, "range out of bounds.");
   {
      std::vector<int64_t> pos((size_t)size);
      for (int64_t i = 0; i < size; i++) pos[i] = beg + i;
      return gpumap_sar_get(ctx, size, pos.data(), vec);
   }
failure_return:
   return -1;
}

// The sequences' names for the hit lines made on the device (k_format_write): back to back, with their offsets.
extern "C" int gpumap_set_seq_names(gpumap_ctx *ctx, const char *const *names, int64_t n_names)
{
   GM_FAIL_IF(ctx == nullptr, "argument 'ctx' is NULL.");
   GM_FAIL_IF(names == nullptr || n_names != ctx->ix.seq_cnt, "one name per sequence of the index.");
   {
      std::vector<uint32_t> off((size_t)n_names + 1, 0);
      std::string blob;
      for (int64_t i = 0; i < n_names; i++) {
         GM_FAIL_IF(names[i] == nullptr, "a name is NULL.");
         blob += names[i];
         off[(size_t)i + 1] = (uint32_t)blob.size();
      }
      HIP_TRY(hipSetDevice(ctx->device));
      HIP_TRY(ctx->d_names.ensure(blob.size() + 16));
      HIP_TRY(ctx->d_name_off.ensure(4 * off.size()));
      HIP_TRY(hipMemcpyAsync(ctx->d_names.p, blob.data(), blob.size(), hipMemcpyHostToDevice, ctx->stream));
      HIP_TRY(hipMemcpyAsync(ctx->d_name_off.p, off.data(), 4 * off.size(), hipMemcpyHostToDevice, ctx->stream));
      HIP_TRY(hipStreamSynchronize(ctx->stream));
   }
   return 0;
failure_return:
   return -1;
}

extern "C" int gpumap_encode(gpumap_ctx *ctx, const char *text, const int64_t *offsets, int64_t n_reads, uint8_t *codes)
{
   GM_FAIL_IF(ctx == nullptr, "argument 'ctx' is NULL.");
   GM_FAIL_IF(n_reads < 0 || offsets == nullptr, "invalid arguments.");
   {
      int64_t total = offsets[n_reads] - offsets[0];
      if (total <= 0) return 0;
      GM_FAIL_IF(text == nullptr || codes == nullptr, "argument 'str' is NULL.");
      HIP_TRY(hipSetDevice(ctx->device));
      HIP_TRY(ctx->op_in.ensure((size_t)total));
      HIP_TRY(ctx->op_out.ensure((size_t)total));
      HIP_TRY(hipMemcpyAsync(ctx->op_in.p, text + offsets[0], (size_t)total, hipMemcpyHostToDevice, ctx->stream));
      hipLaunchKernelGGL(k_encode_flat, dim3((unsigned)std::min<int64_t>(cdiv(total, 256), 1 << 22)), dim3(256), 0, ctx->stream,
                         ctx->op_in.as<const uint8_t>(), total, (const uint8_t *)ctx->d_sym_table, ctx->op_out.as<uint8_t>());
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipMemcpyAsync(codes + offsets[0], ctx->op_out.p, (size_t)total, hipMemcpyDeviceToHost, ctx->stream));
      HIP_TRY(hipStreamSynchronize(ctx->stream));
   }
   return 0;
failure_return:
   return -1;
}

// ------------------------------------------------------------------------------------------------
// Batches
// ------------------------------------------------------------------------------------------------

extern "C" void gpumap_batch_free(gpumap_batch *b)
{
   if (!b) return;
   if (b->ctx) (void)hipSetDevice(b->ctx->device);
   for (hipStream_t st : {b->st_up, b->st_run, b->st_copy, b->st_post}) if (st) (void)hipStreamSynchronize(st);
   b->release();
   delete b;
}

// Batch objects of finished pipelines stay with their context (device buffers, page-locked block, events and all) and
// serve the next pipeline: a mapper() call over a few GB of reads otherwise spends as long on allocating and freeing
// them as on mapping.  gpumap_destroy and option "trim" free them.
gpumap_batch *gm_ctx_take_batch(gpumap_ctx *ctx)
{
   {
      std::lock_guard<std::mutex> lock(ctx->spare_mutex);
      if (!ctx->spare_batches.empty()) {
         gpumap_batch *b = ctx->spare_batches.back();
         ctx->spare_batches.pop_back();
         b->st_run = b->st_copy = b->st_up = b->st_post = b->stream;
         b->on_searched = nullptr;
         b->sc_plan = nullptr; b->extra_factor = 1; b->ran = false; b->encoded = false; b->n_reads = 0;
         return b;
      }
   }
   return gpumap_batch_new(ctx);
}

void gm_ctx_give_batch(gpumap_ctx *ctx, gpumap_batch *b)
{
   if (!b) return;
   (void)hipSetDevice(ctx->device);
   for (hipStream_t st : {b->st_up, b->st_run, b->st_copy, b->st_post}) if (st) (void)hipStreamSynchronize(st);
   {
      std::lock_guard<std::mutex> lock(ctx->spare_mutex);
      static const bool keep = !(getenv("GPUMAP_KEEP_BATCHES") && atoi(getenv("GPUMAP_KEEP_BATCHES")) == 0);
      if (keep && ctx->spare_batches.size() < 8) { ctx->spare_batches.push_back(b); return; }
   }
   gpumap_batch_free(b);
}

static void ctx_free_spare_batches(gpumap_ctx *ctx)
{
   std::vector<gpumap_batch *> spare;
   { std::lock_guard<std::mutex> lock(ctx->spare_mutex); spare.swap(ctx->spare_batches); }
   for (gpumap_batch *b : spare) gpumap_batch_free(b);
}

extern "C" gpumap_batch *gpumap_batch_new(gpumap_ctx *ctx)
{
   gpumap_batch *b = nullptr;
   GM_FAIL_IF(ctx == nullptr, "argument 'ctx' is NULL.");
   HIP_TRY(hipSetDevice(ctx->device));
   b = new gpumap_batch();
   b->ctx = ctx;
   memset(b->length_seen, 0, sizeof b->length_seen);
   HIP_TRY(hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking));
   b->st_run = b->st_copy = b->st_up = b->st_post = b->stream;
   HIP_TRY(hipEventCreateWithFlags(&b->ev_sync, hipEventDisableTiming | hipEventBlockingSync));
   HIP_TRY(hipEventCreateWithFlags(&b->ev_drain, hipEventDisableTiming | hipEventBlockingSync));
   HIP_TRY(hipEventCreateWithFlags(&b->ev_up, hipEventDisableTiming));
   HIP_TRY(hipHostMalloc((void **)&b->h_small, H_SMALL_BYTES, hipHostMallocDefault));
   HIP_TRY(hipEventCreateWithFlags(&b->ev_seen, hipEventDisableTiming));
   for (auto &e : b->ev) HIP_TRY(hipEventCreate(&e));
   HIP_TRY(b->seen.ensure(SEEN_BYTES));
   HIP_TRY(b->counters.ensure(8 * (CNT_WORDS + CNT_SUMS + 2)));     // counters + 64 partial sums of k_verify + 64 of k_verify_group + the count of long intervals (k_place)
   return b;
failure_return:
   gpumap_batch_free(b);
   return nullptr;
}

// The three streams the batches of a pipeline on this context share, created at the first call.  Streams of
// different priority come from different pools of hardware queues in the HIP runtime: the copy stream (high
// priority: on this platform copies are kernels, which take few wavefronts and should not queue behind persistent
// search kernels) therefore never shares a hardware queue with the run stream, whatever other streams the
// process has made (streams of one priority beyond GPU_MAX_HW_QUEUES share queues, and work on one queue is serial).
int gm_ctx_pipe_streams(gpumap_ctx *ctx, void **run, void **copy, void **up, void **post)
{
   HIP_TRY(hipSetDevice(ctx->device));
   {
      std::lock_guard<std::mutex> lock(g_scratch_mutex);
      if (ctx->pipe_run == nullptr) {
         int least = 0, greatest = 0;
         HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));      // numerically: greatest <= 0 <= least
         // No stream priorities (round 3).  What they did depended on things outside the library: in a process of its own,
         // with the copy and post streams of high priority, a chunk's ordering ran beside the next chunk's search (steps of
         // 14.4 ms at GRCh38 size); with all streams of normal priority the kernels ran one after the other (15.3 ms).  In a
         // process that holds an RCCL communicator (bench.py --gpus N) it was the other way round, and mixed settings (copy
         // high, post normal) there gave copies that only ran while no kernel did: 24.5 ms per step (tools/debug/prio_ab4.sh,
         // profiles/r3_stream_priorities.txt).  What the priorities really changed was which streams shared a hardware
         // queue; that is now looked at directly (below).  Priorities also were what stalled the first form of the pipeline
         // (a low-priority upload stream that the run stream waited for) and what stalls it under rocprofv3 --pmc.
         // GPUMAP_PIPE_PRIO=1 / GPUMAP_PIPE_POST_PRIO=1 bring them back.
         const int mode = getenv("GPUMAP_PIPE_PRIO") ? atoi(getenv("GPUMAP_PIPE_PRIO")) : 0;      // tuning aid: 1 = result copies high, 2 = uploads high too
         const int post_mode = getenv("GPUMAP_PIPE_POST_PRIO") ? atoi(getenv("GPUMAP_PIPE_POST_PRIO")) : 0;      // tuning aid: 1 = high priority
         (void)least;
         HIP_TRY(hipStreamCreateWithFlags(&ctx->pipe_run, hipStreamNonBlocking));
         // Streams share hardware queues (GPU_MAX_HW_QUEUES, 4 by default), work on one queue is serial, and which queue a
         // stream gets depends on every stream the process has made before.  The post, copy and upload streams are
         // therefore CHOSEN: of a few candidates the first that is seen to run beside the streams made so far (kernels
         // that spin for a millisecond on those, an empty kernel on the candidate: does it end before they do?).
         static const bool probe = !(getenv("GPUMAP_PIPE_PROBE") && atoi(getenv("GPUMAP_PIPE_PROBE")) == 0);
         static const bool probe_dbg = getenv("GPUMAP_PIPE_DEBUG") != nullptr;
         hipEvent_t ea[3] = {nullptr, nullptr, nullptr}, eb = nullptr;
         for (auto &e : ea) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
         HIP_TRY(hipEventCreateWithFlags(&eb, hipEventDisableTiming));
         auto side_by_side = [&](const std::vector<hipStream_t> &beside, hipStream_t st) -> int {      // 1 yes, 0 no, -1 error
            for (int rep = 0; rep < 2; rep++) {                           // (both tries must say no for a 0)
               int n = 0;
               for (hipStream_t o : beside) {
                  hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, o, (long long)2000000);
                  if (hipEventRecord(ea[n++], o) != hipSuccess) return -1;
               }
               hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, st, (long long)0);
               if (hipEventRecord(eb, st) != hipSuccess) return -1;
               if (hipEventSynchronize(eb) != hipSuccess) return -1;
               bool overtook = true;
               for (int k = 0; k < n; k++) if (hipEventQuery(ea[k]) != hipErrorNotReady) overtook = false;
               for (int k = 0; k < n; k++) if (hipEventSynchronize(ea[k]) != hipSuccess) return -1;
               if (overtook) return 1;
            }
            return 0;
         };
         auto choose = [&](hipStream_t *out, int priority, const std::vector<hipStream_t> &beside, const char *what) -> hipError_t {
            std::vector<hipStream_t> cand;
            int pick = -1;
            for (int k = 0; k < (probe ? 8 : 1) && pick < 0; k++) {
               hipStream_t st = nullptr;
               hipError_t e = hipStreamCreateWithPriority(&st, hipStreamNonBlocking, priority);
               if (e != hipSuccess) return e;
               cand.push_back(st);
               if (!probe || side_by_side(beside, st) == 1) pick = k;
            }
            if (probe_dbg) fprintf(stderr, "[pipe] %s stream: candidate %d of %zu%s\n", what, pick < 0 ? 0 : pick, cand.size(), pick < 0 ? " (none ran beside the others)" : "");
            if (pick < 0) pick = 0;
            *out = cand[(size_t)pick];
            for (size_t k = 0; k < cand.size(); k++) if ((int)k != pick) (void)hipStreamDestroy(cand[k]);
            return hipSuccess;
         };
         HIP_TRY(choose(&ctx->pipe_post, post_mode >= 1 ? greatest : 0, {ctx->pipe_run}, "post"));
         HIP_TRY(choose(&ctx->pipe_copy, mode >= 1 ? greatest : 0, {ctx->pipe_run, ctx->pipe_post}, "copy"));
         HIP_TRY(choose(&ctx->pipe_up, mode >= 2 ? greatest : 0, {ctx->pipe_run, ctx->pipe_post, ctx->pipe_copy}, "upload"));
         for (auto &e : ea) (void)hipEventDestroy(e);
         (void)hipEventDestroy(eb);
      }
   }
   *run = ctx->pipe_run; *copy = ctx->pipe_copy; *up = ctx->pipe_up; *post = ctx->pipe_post;
   return 0;
failure_return:
   return -1;
}

void gm_batch_set_streams(gpumap_batch *b, void *run, void *copy, void *up, void *post)
{
   b->st_run = (hipStream_t)run; b->st_copy = (hipStream_t)copy; b->st_up = (hipStream_t)up; b->st_post = (hipStream_t)post;
}

void gm_batch_set_hook(gpumap_batch *b, std::function<void()> on_searched) { b->on_searched = std::move(on_searched); }
void gm_batch_set_gate(gpumap_batch *b, std::function<void()> before_search, std::function<void()> on_ordering)
{
   b->before_search = std::move(before_search); b->on_ordering = std::move(on_ordering);
}

// Waits until everything this thread has put on `st` so far is done (not for what other batches add meanwhile).
static const bool gm_sync_debug = getenv("GPUMAP_SYNC_DEBUG") != nullptr;      // (debug aid: which wait a thread sits in)
#define GM_SYNC(b, st) do { \
      if (gm_sync_debug) fprintf(stderr, "[sync] batch %p line %d stream %p: wait\n", (void *)(b), __LINE__, (void *)(st)); \
      HIP_TRY(hipEventRecord((b)->ev_sync, (st))); HIP_TRY(hipEventSynchronize((b)->ev_sync)); \
      if (gm_sync_debug) fprintf(stderr, "[sync] batch %p line %d: over\n", (void *)(b), __LINE__); \
   } while (0)

// Encoding of a filled batch: the lengths seen (k_read_lengths, read back by gm_batch_fill) size the read
// matrix; letters (or codes) -> codes in fixed-stride rows (k_encode) and the rows as bit planes (k_pack_planes).
// want_bytes = false: only the planes are written (the byte matrix is three quarters of what the kernel writes, and the
// mapping search reads none of it); a run that needs the bytes of a batch encoded without them encodes it again.
static int batch_encode(gpumap_ctx *ctx, gpumap_batch *b, bool want_bytes)
{
   hipStream_t st = b->st_run;
   const int64_t n_reads = b->n_reads;
   if (n_reads > 0) {
      HIP_TRY(hipEventSynchronize(b->ev_seen));
      if (b->st_up != st) HIP_TRY(hipStreamWaitEvent(st, b->ev_up, 0));      // the letters are on the device
      const uint8_t *seen = b->h_small;
      GM_FAIL_IF(seen[SEEN_ERROR] != 0, "offsets must not decrease.");
      GM_FAIL_IF(b->from_file && seen[SEEN_QUAL] != 0, "inconsistent length of sequence quality.");      // src/seqread.c:47-49
      int64_t maxlen = 1;
      for (int m = 0; m <= MAX_READ; m++) {
         b->length_seen[m] = seen[m] != 0;
         if (seen[m] && m > maxlen) maxlen = m;
      }
      b->stride = (int)((maxlen + 127) & ~(int64_t)127);    // the rows as bit planes come in units of 128 symbols
      b->maxlen = (int)maxlen;
      if (want_bytes) HIP_TRY(b->reads.ensure((size_t)n_reads * b->stride * (size_t)b->extra_factor));
      HIP_TRY(b->reads4.ensure((size_t)n_reads * (size_t)(12 * (b->stride / 32))));
      const int64_t groups = n_reads * (b->stride / 32);
      hipLaunchKernelGGL(k_encode, dim3((unsigned)std::min<int64_t>(cdiv(groups, 256), 1 << 22)), dim3(256), 0, st,
                         b->d_src, b->d_off, b->d_end, b->src_relative ? 1 : 0, n_reads, b->stride,
                         b->src_text ? (const uint8_t *)ctx->d_sym_table : (const uint8_t *)nullptr, want_bytes ? b->reads.as<uint8_t>() : (uint8_t *)nullptr,
                         b->reads4.as<uint32_t>());
      HIP_TRY(hipGetLastError());
   } else {
      memset(b->length_seen, 0, sizeof b->length_seen);
      b->stride = 128; b->maxlen = 1;
   }
   b->encoded = true; b->have_bytes = want_bytes;
   return 0;
failure_return:
   return -1;
}

// Fills `b` with a new set of reads; the device buffers `b` already holds are kept and grow only when
// the new set needs more (a caller that maps chunk after chunk pays for the allocations once).
// src_flags: GM_SRC_TEXT = letters, else codes; GM_SRC_DEVICE = src and offsets are device pointers, used in
// place (they must stay valid until the batch has been run).  encode_now = false leaves the batch staged:
// gm_batch_run encodes it first, inside its timed region.
int gm_batch_fill(gpumap_ctx *ctx, gpumap_batch *b, const void *src, const int64_t *offsets, int64_t n_reads, int src_flags, bool encode_now)
{
   GM_FAIL_IF(n_reads < 0 || offsets == nullptr, "invalid arguments.");
   GM_FAIL_IF(n_reads >= ((int64_t)1 << 31), "too many reads in one batch.");
   HIP_TRY(hipSetDevice(ctx->device));
   {
      hipStream_t st = b->st_up;
      b->n_reads = n_reads;
      b->ran = false; b->have_intervals = false; b->encoded = false; b->have_lines = false;
      b->src_text = (src_flags & GM_SRC_TEXT) != 0;
      b->from_file = false; b->d_end = nullptr;
      HIP_TRY(b->read_len.ensure(2 * (size_t)std::max<int64_t>(1, n_reads) * (size_t)b->extra_factor));
      HIP_TRY(b->status.ensure(8 * (size_t)std::max<int64_t>(1, n_reads)));
      int64_t total = 0;
      if (src_flags & GM_SRC_FILE) {
         // `src` = a record-aligned block of the read file (host memory), `offsets` = its description (gpumap.h): the
         // block is copied to the device and parsed there (gm_parse.hpp)
         GM_FAIL_IF(src_flags & GM_SRC_DEVICE, "a block of a file comes from host memory.");
         const int64_t bytes = offsets[0], per = offsets[1], n_lines = offsets[2];
         GM_FAIL_IF(bytes < 0 || bytes >= ((int64_t)1 << 31), "a block of a read file holds fewer than 2 GiB.");
         GM_FAIL_IF(per != 1 && per != 2 && per != 4, "lines per record: 4 (FASTQ), 2 (FASTA) or 1 (raw).");
         GM_FAIL_IF(n_lines < 0 || n_reads * per > n_lines, "the block holds fewer lines than its records need.");
         GM_FAIL_IF(bytes > 0 && src == nullptr, "argument 'src' is NULL.");
         b->from_file = true; b->src_text = true; b->lines_per_rec = (int)per; b->first_line = offsets[4];
         const size_t RX = (size_t)std::max<int64_t>(1, n_reads);
         const int64_t tiles = cdiv(std::max<int64_t>(bytes, 1), LINE_TILE);
         HIP_TRY(b->up_src.ensure((size_t)bytes + 64));
         HIP_TRY(b->nl_pos.ensure(4 * (size_t)(n_lines + 1)));
         HIP_TRY(b->tile_cnt.ensure(4 * (size_t)(tiles + 1)));
         HIP_TRY(b->tile_off.ensure(4 * (size_t)(tiles + 1)));
         HIP_TRY(b->raw_beg.ensure(8 * RX)); HIP_TRY(b->raw_end.ensure(8 * RX));
         HIP_TRY(b->tag_off.ensure(4 * RX)); HIP_TRY(b->tag_len.ensure(4 * RX));
         HIP_TRY(b->nul_flag.ensure(8));
         if (bytes > 0) HIP_TRY(hipMemcpyAsync(b->up_src.p, src, (size_t)bytes, hipMemcpyHostToDevice, st));
         HIP_TRY(hipMemsetAsync(b->seen.p, 0, SEEN_BYTES, st));
         HIP_TRY(hipMemsetAsync(b->nul_flag.p, 0, 8, st));
         HIP_TRY(hipMemsetAsync(b->tile_cnt.as<uint32_t>() + tiles, 0, 4, st));
         if (bytes > 0) {
            hipLaunchKernelGGL(k_line_count, dim3((unsigned)tiles), dim3(256), 0, st, b->up_src.as<const uint8_t>(), bytes,
                               b->tile_cnt.as<uint32_t>(), b->nul_flag.as<uint32_t>());
            size_t tmp = 0;
            HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp, b->tile_cnt.as<uint32_t>(), b->tile_off.as<uint32_t>(), (int)(tiles + 1), st));
            HIP_TRY(b->cub_tmp.ensure(tmp));
            HIP_TRY(hipcub::DeviceScan::ExclusiveSum(b->cub_tmp.p, tmp, b->tile_cnt.as<uint32_t>(), b->tile_off.as<uint32_t>(), (int)(tiles + 1), st));
            hipLaunchKernelGGL(k_line_write, dim3((unsigned)tiles), dim3(256), 0, st, b->up_src.as<const uint8_t>(), bytes,
                               b->tile_off.as<const uint32_t>(), b->nl_pos.as<uint32_t>(), (uint32_t)(n_lines + 1));
         } else {
            HIP_TRY(hipMemsetAsync(b->tile_off.p, 0, 4 * (size_t)(tiles + 1), st));
         }
         if (n_reads > 0) {
            ParseArgs P;
            P.raw = b->up_src.as<const uint8_t>(); P.n = bytes; P.nl_pos = b->nl_pos.as<const uint32_t>();
            P.n_nl_dev = b->tile_off.as<const uint32_t>() + tiles; P.nl_cap = n_lines + 1;
            P.lines_per_rec = (int)per; P.n_rec = n_reads; P.nul_flag = b->nul_flag.as<const uint32_t>();
            P.beg = b->raw_beg.as<int64_t>(); P.end = b->raw_end.as<int64_t>(); P.read_len = b->read_len.as<uint16_t>();
            P.seen = b->seen.as<uint8_t>(); P.tag_off = b->tag_off.as<uint32_t>(); P.tag_len = b->tag_len.as<uint32_t>();
            hipLaunchKernelGGL(k_parse_records, dim3((unsigned)std::min<int64_t>(cdiv(n_reads, 256), 4096)), dim3(256), 0, st, P);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpyAsync(b->h_small, b->seen.p, SEEN_BYTES, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipEventRecord(b->ev_seen, st));
         }
         b->d_src = b->up_src.as<const uint8_t>(); b->d_off = b->raw_beg.as<const int64_t>(); b->d_end = b->raw_end.as<const int64_t>();
         b->src_relative = false;
         HIP_TRY(hipEventRecord(b->ev_up, st));
         return 0;
      }
      if (src_flags & GM_SRC_DEVICE) {
         GM_FAIL_IF(n_reads > 0 && src == nullptr, "argument 'src' is NULL.");
         b->d_src = (const uint8_t *)src; b->d_off = offsets; b->src_relative = false;
      } else {
         total = offsets[n_reads] - offsets[0];
         GM_FAIL_IF(total < 0, "offsets must not decrease.");
         GM_FAIL_IF(total > 0 && src == nullptr, "argument 'codes' is NULL.");
         HIP_TRY(b->up_off.ensure(8 * (size_t)(n_reads + 1)));
         HIP_TRY(b->up_src.ensure((size_t)std::max<int64_t>(1, total)));
         HIP_TRY(hipMemcpyAsync(b->up_off.p, offsets, 8 * (size_t)(n_reads + 1), hipMemcpyHostToDevice, st));
         b->d_src = b->up_src.as<const uint8_t>(); b->d_off = b->up_off.as<const int64_t>(); b->src_relative = true;
      }
      if (n_reads > 0) {
         HIP_TRY(hipMemsetAsync(b->seen.p, 0, SEEN_BYTES, st));
         hipLaunchKernelGGL(k_read_lengths, dim3((unsigned)std::min<int64_t>(cdiv(n_reads, 256), 4096)), dim3(256), 0, st,
                            b->d_off, n_reads, b->read_len.as<uint16_t>(), b->seen.as<uint8_t>());
         HIP_TRY(hipGetLastError());
         HIP_TRY(hipMemcpyAsync(b->h_small, b->seen.p, SEEN_BYTES, hipMemcpyDeviceToHost, st));
         HIP_TRY(hipEventRecord(b->ev_seen, st));
      }
      if (total > 0) HIP_TRY(hipMemcpyAsync(b->up_src.p, (const uint8_t *)src + offsets[0], (size_t)total, hipMemcpyHostToDevice, st));
      HIP_TRY(hipEventRecord(b->ev_up, st));
      if (encode_now) {
         GM_FAIL_IF(batch_encode(ctx, b, true) != 0, "cannot encode the batch.");
         GM_SYNC(b, b->st_run);                         // the caller's buffers are free again
      }
   }
   return 0;
failure_return:
   return -1;
}

static gpumap_batch *batch_upload(gpumap_ctx *ctx, const uint8_t *src, const int64_t *offsets, int64_t n_reads, bool is_text)
{
   gpumap_batch *b = gpumap_batch_new(ctx);
   GM_FAIL_IF(b == nullptr, "cannot create the batch.");
   GM_FAIL_IF(gm_batch_fill(ctx, b, src, offsets, n_reads, is_text ? GM_SRC_TEXT : 0, true) != 0, "cannot upload the batch.");
   b->up_src.release(); b->up_off.release();             // a one-off batch does not keep its staging
   b->d_src = nullptr; b->d_off = nullptr;
   return b;
failure_return:
   gpumap_batch_free(b);
   return nullptr;
}

extern "C" int gpumap_batch_refill_text(gpumap_ctx *ctx, gpumap_batch *b, const char *text, const int64_t *offsets, int64_t n_reads)
{
   GM_FAIL_IF(ctx == nullptr, "argument 'ctx' is NULL.");
   GM_FAIL_IF(b == nullptr, "argument 'batch' is NULL.");
   GM_FAIL_IF(b->ctx != ctx, "the batch belongs to another context.");
   return gm_batch_fill(ctx, b, text, offsets, n_reads, GM_SRC_TEXT, true);
failure_return:
   return -1;
}

extern "C" int gpumap_batch_stage(gpumap_ctx *ctx, gpumap_batch *b, const void *src, const int64_t *offsets, int64_t n_reads, int flags)
{
   GM_FAIL_IF(ctx == nullptr, "argument 'ctx' is NULL.");
   GM_FAIL_IF(b == nullptr, "argument 'batch' is NULL.");
   GM_FAIL_IF(b->ctx != ctx, "the batch belongs to another context.");
   GM_FAIL_IF(flags & ~(GPUMAP_SRC_TEXT | GPUMAP_SRC_DEVICE), "unknown flag.");
   return gm_batch_fill(ctx, b, src, offsets, n_reads, flags, false);
failure_return:
   return -1;
}

extern "C" gpumap_batch *gpumap_batch_upload(gpumap_ctx *ctx, const uint8_t *codes, const int64_t *offsets, int64_t n_reads)
{
   GM_FAIL_IF(ctx == nullptr, "argument 'ctx' is NULL.");
   return batch_upload(ctx, codes, offsets, n_reads, false);
failure_return:
   return nullptr;
}

extern "C" gpumap_batch *gpumap_batch_upload_text(gpumap_ctx *ctx, const char *text, const int64_t *offsets, int64_t n_reads)
{
   GM_FAIL_IF(ctx == nullptr, "argument 'ctx' is NULL.");
   return batch_upload(ctx, (const uint8_t *)text, offsets, n_reads, true);
failure_return:
   return nullptr;
}

static int bits_for(uint64_t v) { int b = 0; while (b < 64 && (v >> b)) b++; return b; }

// Small uploads of one run (phase tables) go through the batch's page-locked block: the copies are asynchronous
// and the block is only rewritten by the next run of the same batch, after this run's last synchronisation.
struct SmallStage {
   uint8_t *base;
   size_t cap;
   size_t used = 0;
   template <class T> const T *put(const std::vector<T> &v)
   {
      const size_t bytes = sizeof(T) * v.size();
      used = (used + 15) & ~(size_t)15;
      if (used + bytes > cap) return nullptr;
      T *p = reinterpret_cast<T *>(base + used);
      memcpy(p, v.data(), bytes);
      used += bytes;
      return p;
   }
};

// The suffixes of the list's intervals added up (the locate stage's slots): sizes, their scan, the total to *h_total
// (page-locked) -- enqueued on `st`, read by the host after its next wait for that stream.
static int batch_count_intervals(gpumap_batch *b, hipStream_t st, int64_t n_look, const uint32_t *dense, int64_t *h_total)
{
   HIP_TRY(b->hit_sizes.ensure(8 * (size_t)(n_look + 1)));
   HIP_TRY(b->hit_off.ensure(8 * (size_t)(n_look + 1)));
   HIP_TRY(hipMemsetAsync(b->hit_sizes.as<int64_t>() + n_look, 0, 8, st));
   hipLaunchKernelGGL(k_hit_sizes, dim3((unsigned)cdiv(n_look, 256)), dim3(256), 0, st, b->hits.as<const Node>(), dense, n_look,
                      b->hit_sizes.as<int64_t>());
   {
      size_t tmp = 0;
      HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp, b->hit_sizes.as<int64_t>(), b->hit_off.as<int64_t>(), (int)(n_look + 1), st));
      HIP_TRY(b->cub_tmp.ensure(tmp));
      HIP_TRY(hipcub::DeviceScan::ExclusiveSum(b->cub_tmp.p, tmp, b->hit_sizes.as<int64_t>(), b->hit_off.as<int64_t>(), (int)(n_look + 1), st));
   }
   HIP_TRY(hipMemcpyAsync(h_total, b->hit_off.as<int64_t>() + n_look, 8, hipMemcpyDeviceToHost, st));
   return 0;
failure_return:
   return -1;
}

int gm_batch_run(gpumap_ctx *ctx, gpumap_batch *b, int tau, int want_intervals, int result_flags)
{
   GM_FAIL_IF(ctx == nullptr || b == nullptr, "NULL argument.");
   GM_FAIL_IF(b->ctx != ctx, "batch belongs to another context.");
   GM_FAIL_IF(tau < 0, "argument 'tau' must be positive.");
   GM_FAIL_IF(tau > MAX_TAU, "argument 'tau' exceeds GPUMAP_MAX_TAU.");
   HIP_TRY(hipSetDevice(ctx->device));
   {
      hipStream_t st = b->st_run;
      const int64_t R = b->n_reads;
      const bool compact = (result_flags & GM_RESULT_COMPACT) != 0 && !want_intervals;
      b->tau = tau; b->ran = false; b->n_hits = b->n_locs = b->n_ext = 0; b->have_intervals = false;
      b->sorted_keys = nullptr; b->compact = compact; b->n_big = 0;
      // ---- K4 (a staged batch): letters -> codes -> bit planes, inside the timed region
      HIP_TRY(hipEventRecord(b->ev[5], st));
      {
         // the byte matrix of the reads: for blocksc_trail (k_sc_prepare, the plan's queries) and the hits' alignments (k_unpack_hits)
         const bool want_bytes = (result_flags & GM_RUN_SC) != 0 || want_intervals != 0;
         if (!b->encoded || (want_bytes && !b->have_bytes)) GM_FAIL_IF(batch_encode(ctx, b, want_bytes) != 0, "cannot encode the batch.");
      }
      // ---- schedule: one phase-table row per read length present (src/blocksearch.c:138-170); for blocksc_trail
      // one per (length, tau - N count, strand class) (src/blocksearch.c:55-70)
      const bool sc = (result_flags & GM_RUN_SC) != 0;
      int64_t RS = R;                 // queries the search runs (gpumap_ann_build's plan may add some behind the batch's)
      std::vector<uint16_t> len_row(MAX_READ + 1, NO_ROW);
      std::vector<uint32_t> phases, phase_rec;
      std::vector<uint8_t> phase_lo, row_leaves, row_state;
      std::vector<uint16_t> row_len;
      int rows = 0;
      AnnPlanHost *plan = sc ? b->sc_plan : nullptr;
      int n_nat = 0;                  // rows of the queries' own schedules (the plan's rows come behind them)
      if (sc) {
         int n_len = 0;
         for (int m = 1; m <= MAX_READ; m++) if (b->length_seen[m]) n_len++;
         GM_FAIL_IF(n_len * (tau + 1) * 2 * (plan ? 3 : 1) > 256, "too many distinct query lengths for one batch of blocksc_trail searches.");
         auto add_row = [&](int m, int te, int last, int tl_forced) {
            uint32_t tab[MAX_PHASES];
            uint8_t lo_tab[MAX_PHASES];
            memset(tab, 0, sizeof tab);
            memset(lo_tab, 0, sizeof lo_tab);
            int scp = 0, leaves = 0;
            const int st = build_sc_phase_table(m, tau, te, last, tab, lo_tab, &scp, &leaves, tl_forced);
            rows++;
            row_len.push_back((uint16_t)m);
            row_leaves.push_back((uint8_t)leaves);
            row_state.push_back((uint8_t)(st == SC_SEARCH ? 0 : (st == SC_NO_HITS ? 1 : 2)));
            phases.insert(phases.end(), tab, tab + MAX_PHASES);
            phase_lo.insert(phase_lo.end(), lo_tab, lo_tab + MAX_PHASES);
            for (int p = 0; p < MAX_PHASES; p++) {
               uint32_t x = 0, y = 0;
               if (st == SC_SEARCH && p <= scp) {
                  phase_record(tab, p, x, y);
                  if (phase_next(tab[p]) == scp) y |= 1u << 9;        // a hit of the left block enters the bounded extension
               }
               phase_rec.push_back(x); phase_rec.push_back(y);
            }
         };
         for (int m = 1; m <= MAX_READ; m++) {
            if (!b->length_seen[m]) continue;
            len_row[m] = (uint16_t)(rows / ((tau + 1) * 2));
            for (int te = 0; te <= tau; te++)
               for (int last = 0; last < 2; last++) add_row(m, te, last, -1);
         }
         n_nat = rows;
         // the plan's rows: row n_nat + 2 * r + tl = the schedule of row r with the left block searched with tl mismatches
         if (plan)
            for (int r = 0; r < n_nat; r++)
               for (int tl = 0; tl < 2; tl++) add_row(row_len[(size_t)r], (r >> 1) % (tau + 1), r & 1, tl);
      }
      for (int m = 1; m <= MAX_READ && !sc; m++) {
         if (!b->length_seen[m]) continue;
         uint32_t tab[MAX_PHASES];
         uint8_t lo_tab[MAX_PHASES];
         memset(tab, 0, sizeof tab);
         memset(lo_tab, 0, sizeof lo_tab);
         if (!build_phase_table(m, tau, tab, lo_tab)) continue;        // the reference returns -1 for this length
         len_row[m] = (uint16_t)rows++;                                 // at most 256 rows: the node keeps its row in one byte
         row_len.push_back((uint16_t)m);
         phases.insert(phases.end(), tab, tab + MAX_PHASES);
         phase_lo.insert(phase_lo.end(), lo_tab, lo_tab + MAX_PHASES);
         for (int p = 0; p < MAX_PHASES; p++) {
            uint32_t x = 0, y = 0;
            if (p < 3 * tau + 1) phase_record(tab, p, x, y);
            phase_rec.push_back(x); phase_rec.push_back(y);
         }
      }
      if (phases.empty()) { phases.assign(MAX_PHASES, 0); phase_rec.assign(2 * MAX_PHASES, 0); phase_lo.assign(MAX_PHASES, 0); }
      if (row_len.empty()) row_len.push_back(0);
      {
         SmallStage ss{b->h_small + 512, H_SMALL_BYTES - 512 - HOST_CNT_BYTES - 1024};                               // (the first bytes hold the lengths seen)
         const uint16_t *h_len_row = ss.put(len_row);
         const uint32_t *h_phases = ss.put(phases), *h_rec = ss.put(phase_rec);
         const uint8_t *h_lo = ss.put(phase_lo);
         const uint16_t *h_row_len = ss.put(row_len);
         GM_FAIL_IF(!h_len_row || !h_phases || !h_rec || !h_lo || !h_row_len, "phase tables do not fit the staging block.");
         HIP_TRY(b->len_row.ensure(2 * len_row.size()));
         HIP_TRY(b->phases.ensure(4 * phases.size()));
         HIP_TRY(b->phase_rec.ensure(4 * phase_rec.size()));
         HIP_TRY(b->phase_lo.ensure(phase_lo.size()));
         HIP_TRY(b->row_len.ensure(2 * row_len.size()));
         HIP_TRY(hipMemcpyAsync(b->phase_rec.p, h_rec, 4 * phase_rec.size(), hipMemcpyHostToDevice, st));
         HIP_TRY(hipMemcpyAsync(b->phase_lo.p, h_lo, phase_lo.size(), hipMemcpyHostToDevice, st));
         HIP_TRY(hipMemcpyAsync(b->len_row.p, h_len_row, 2 * len_row.size(), hipMemcpyHostToDevice, st));
         HIP_TRY(hipMemcpyAsync(b->row_len.p, h_row_len, 2 * row_len.size(), hipMemcpyHostToDevice, st));
         HIP_TRY(hipMemcpyAsync(b->phases.p, h_phases, 4 * phases.size(), hipMemcpyHostToDevice, st));
      }
      if (sc) {
         if (row_leaves.empty()) { row_leaves.push_back(0); row_state.push_back(1); }
         SmallStage s2{b->h_small + H_SMALL_BYTES - HOST_CNT_BYTES - 1024, 1024};      // (the KiB before the counters: 2 x 256 bytes)
         const uint8_t *h_leaves = s2.put(row_leaves), *h_state = s2.put(row_state);
         GM_FAIL_IF(!h_leaves || !h_state, "row tables do not fit the staging block.");
         HIP_TRY(b->row_leaves.ensure(row_leaves.size()));
         HIP_TRY(b->row_state.ensure(row_state.size()));
         const size_t RX = (size_t)std::max<int64_t>(R, 1) * (size_t)b->extra_factor;
         HIP_TRY(b->read_row.ensure(2 * RX));
         HIP_TRY(b->sc_bound.ensure(8 * RX));
         HIP_TRY(b->score_add.ensure(RX));
         if (plan) { HIP_TRY(b->sc_flags.ensure(RX)); HIP_TRY(b->sc_alias.ensure(4 * RX)); }
         HIP_TRY(b->sc_qfp.ensure(8 * (size_t)std::max<int64_t>(R, 1)));
         HIP_TRY(b->sc_qrp.ensure(8 * (size_t)std::max<int64_t>(R, 1)));
         HIP_TRY(hipMemcpyAsync(b->row_leaves.p, h_leaves, row_leaves.size(), hipMemcpyHostToDevice, st));
         HIP_TRY(hipMemcpyAsync(b->row_state.p, h_state, row_state.size(), hipMemcpyHostToDevice, st));
      }
      if (R > 0 && sc) {
         hipLaunchKernelGGL(k_sc_prepare, dim3((unsigned)cdiv(R, 256)), dim3(256), 0, st, ctx->ix, b->reads.as<const uint8_t>(),
                            b->read_len.as<const uint16_t>(), b->len_row.as<const uint16_t>(), b->row_state.as<const uint8_t>(), R,
                            b->stride, tau, b->read_row.as<uint16_t>(), b->sc_bound.as<int64_t>(), b->score_add.as<uint8_t>(),
                            b->status.as<int64_t>(), b->sc_qfp.as<int64_t>(), b->sc_qrp.as<int64_t>(),
                            (const uint4 *)ctx->d_jump, ctx->jump_k, (const uint4 *)ctx->d_jump2, ctx->jump2_k);
         HIP_TRY(hipGetLastError());
         if (plan) {
            // the reference's reuse of the last k-mer's search (gm_ann.hpp): may patch queries and add some behind the R of the chunk
            GM_FAIL_IF(b->extra_factor < 2 || tau > 5, "the batch is not set up for the plan.");
            AnnPlanArgs P;
            P.ix = ctx->ix; P.text = (const uint8_t *)ctx->d_text; P.q_sa = plan->q_sa; P.c0 = plan->c0; P.nq = R;
            P.kmer = plan->kmer; P.tau = tau; P.depth = plan->depth; P.stride = b->stride; P.n_nat = n_nat;
            P.q_fp = b->sc_qfp.as<const int64_t>(); P.q_rp = b->sc_qrp.as<const int64_t>();
            P.reads = b->reads.as<uint8_t>(); P.read_len = b->read_len.as<uint16_t>(); P.read_row = b->read_row.as<uint16_t>();
            P.sc_bound = b->sc_bound.as<int64_t>(); P.score_add = b->score_add.as<uint8_t>(); P.sc_flags = b->sc_flags.as<uint8_t>();
            P.alias = b->sc_alias.as<uint32_t>();
            for (DevBuf *x : {&plan->s1, &plan->s2a, &plan->s2b}) HIP_TRY(x->ensure(8 * (size_t)R));
            HIP_TRY(plan->n_extra_dev.ensure(8));
            P.s1 = plan->s1.as<unsigned long long>(); P.s2a = plan->s2a.as<unsigned long long>(); P.s2b = plan->s2b.as<unsigned long long>();
            P.carry = plan->carry.as<AnnCarry>(); P.n_extra = plan->n_extra_dev.as<unsigned int>();
            HIP_TRY(hipMemsetAsync(P.n_extra, 0, 8, st));
            const dim3 pg((unsigned)cdiv(R, 256)), pb(256);
            auto scan = [&](unsigned long long *v) -> hipError_t {
               size_t bytes = 0;
               hipError_t e = hipcub::DeviceScan::InclusiveScan(nullptr, bytes, v, v, MaxHalves(), (int)R, st);
               if (e != hipSuccess) return e;
               e = plan->tmp.ensure(bytes);
               if (e != hipSuccess) return e;
               return hipcub::DeviceScan::InclusiveScan(plan->tmp.p, bytes, v, v, MaxHalves(), (int)R, st);
            };
            hipLaunchKernelGGL(k_ann_plan_class, pg, pb, 0, st, P);
            HIP_TRY(scan(P.s1));
            hipLaunchKernelGGL(k_ann_plan_trail, pg, pb, 0, st, P);
            HIP_TRY(scan(P.s2a));
            HIP_TRY(scan(P.s2b));
            hipLaunchKernelGGL(k_ann_plan_apply, pg, pb, 0, st, P);
            hipLaunchKernelGGL(k_ann_plan_carry, dim3(1), dim3(1), 0, st, P);
            HIP_TRY(hipGetLastError());
            unsigned int *h_extra = reinterpret_cast<unsigned int *>(b->h_small + H_SMALL_BYTES - HOST_CNT_BYTES - 1024 + 768);
            HIP_TRY(hipMemcpyAsync(h_extra, P.n_extra, 4, hipMemcpyDeviceToHost, st));
            GM_SYNC(b, st);
            plan->n_extra = (int64_t)*h_extra;
            GM_FAIL_IF(plan->n_extra > R, "the plan added more queries than the batch has room for.");
            RS = R + plan->n_extra;
         }
      } else if (R > 0) {
         hipLaunchKernelGGL(k_status, dim3((unsigned)cdiv(R, 256)), dim3(256), 0, st, b->read_len.as<const uint16_t>(),
                            b->len_row.as<const uint16_t>(), R, b->status.as<int64_t>());
         HIP_TRY(hipGetLastError());
      }

      // ---- K2 search (rerun once with an exact buffer if the hit list overflowed)
      // persistent waves: as many workgroups as are resident at once (the staged superblock table grows with
      // the text and can cost a workgroup per CU), unless the option asks for a number
      const size_t search_lds = (size_t)ctx->ix.n_super * SYMS * 8;
      int resident_wg = 0;
      if (sc) HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&resident_wg, k_search<true>, WG_THREADS, search_lds));
      else HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&resident_wg, k_search<false>, WG_THREADS, search_lds));
      if (getenv("GPUMAP_VERBOSE")) fprintf(stderr, "k_search: %d workgroups of %d waves resident per CU (dynamic LDS %zu B)\n", resident_wg, WG_WAVES, search_lds);
      // 16 waves per CU where four workgroups fit (texts below 2.2 G symbols: the table is small) and the batch
      // is large enough (small batches start and drain faster with 12).  At GRCh38 size 12 and 16 waves run
      // alike: the kernel is bound by the rate of random line requests there, not by latency.
      const int64_t refills = R * (tau + 1) / 64;
      const int want_wg = refills >= (int64_t)16 * ctx->cu_count * 16 ? 4 : 3;
      const int wpc = ctx->waves_per_cu ? ctx->waves_per_cu : WG_WAVES * std::max(1, std::min(want_wg, resident_wg));
      const int n_wg = std::max(1, ctx->cu_count * wpc / WG_WAVES);
      const int64_t n_waves = (int64_t)n_wg * WG_WAVES;
      HIP_TRY(b->spill.ensure((size_t)n_waves * (size_t)ctx->spill_cap * sizeof(Node)));
      unsigned long long *cnt = b->counters.as<unsigned long long>();   // [0] cursor [1] hits [2] n_ext [3] error
      unsigned long long *host_cnt = reinterpret_cast<unsigned long long *>(b->h_small + H_SMALL_BYTES - HOST_CNT_BYTES);   // page-locked
      constexpr size_t CNT_BYTES = 8 * (CNT_WORDS + CNT_SUMS);
      memset(host_cnt, 0, CNT_BYTES);
      // interval output needs the final interval of every hit: no short cut through the text then
      // 0: every step on the Occ table; 1: nodes of one suffix are finished on the text; 2: also nodes of two; 4: also
      // nodes of three and four whose interval has survived a step unchanged (k_verify_group).
      // -1 (default): 4 for texts of 2^31 symbols and more, else 1 (measured: pairs and groups pay where most k-mers
      // of the jump table's length occur more than once; at GRCh38 size 22.6 -> 21.0 ms per batch with pairs,
      // 19.1 -> 18.0 with groups; pairs cost 13 % at chr21 size)
      const int verify_mode = ctx->verify >= 0 ? ctx->verify : (ctx->ix.n >= ((int64_t)1 << 26) ? 4 : 1);
      const int use_verify = (verify_mode && ctx->d_text != nullptr && ctx->d_text4 != nullptr && !want_intervals && !sc) ? verify_mode : 0;
      // first guess of the list's length (a batch that is run or refilled again keeps the longer list): with
      // k_verify every seed can leave a candidate, plus what repeats add (5.5 entries per read at GRCh38
      // size, tau 2); too short costs a second run of the search.  Option "hit_list_guess" overrides (tests).
      size_t hit_cap = ctx->hit_guess > 0 ? (size_t)ctx->hit_guess
                                          : std::max<size_t>((size_t)1 << 16, (size_t)R * (use_verify ? (size_t)tau + 4 : 2));
      HIP_TRY(hipEventRecord(b->ev[0], st));
      b->attempts = 0;
      for (int attempt = 0; attempt < 2; attempt++) {
         b->attempts = attempt + 1;
         HIP_TRY(b->hits.ensure(hit_cap * sizeof(Node)));
         if (!(ctx->hit_guess > 0 && attempt == 0)) hit_cap = b->hits.cap / sizeof(Node);      // (a guess given by option is taken as it is: tests)
         HIP_TRY(hipMemsetAsync(cnt, 0, CNT_BYTES, st));
         SearchArgs A;
         A.ix = ctx->ix;
         A.reads = b->reads.as<const uint8_t>(); A.planes = b->reads4.as<const uint32_t>(); A.read_len = b->read_len.as<const uint16_t>();
         A.len_row = b->len_row.as<const uint16_t>(); A.phases = b->phases.as<const uint32_t>();
         A.phase_rec = b->phase_rec.as<const uint2>(); A.n_rows = std::max(rows, 1);
         A.n_reads = RS; A.stride = b->stride; A.tau = tau;
         A.cursor = cnt + 0; A.hits = b->hits.as<Node>(); A.hit_cap = hit_cap; A.hit_count = cnt + 1; A.n_ext = cnt + 2;
         A.spill = b->spill.as<Node>(); A.spill_cap = (uint32_t)ctx->spill_cap; A.error = (int *)(cnt + 3); A.stats = cnt + 4;
         A.jump = (const uint4 *)ctx->d_jump; A.jump_k = ctx->jump_k;
         A.jump2 = (const uint4 *)ctx->d_jump2; A.jump2_k = ctx->jump2_k;
         A.verify = use_verify;
         A.verify_stable = 1;
         A.seed_cand = (use_verify >= 1 && ctx->seed_candidates) ? 1 : 0;
         A.read_row = sc ? b->read_row.as<const uint16_t>() : nullptr;
         A.sc_bound = sc ? b->sc_bound.as<const int64_t>() : nullptr;
         A.row_leaves = sc ? b->row_leaves.as<const uint8_t>() : nullptr;
         A.sc_flags = plan ? b->sc_flags.as<const uint8_t>() : nullptr;
         {
            const int rpr = 64 / (tau + 1);
            int64_t k = R / std::max<int64_t>(1, n_waves * 8 * rpr);           // at least 8 reservations per wave: balance at the end
            A.cursor_chunk = (uint32_t)(rpr * std::min<int64_t>(16, std::max<int64_t>(1, k)));
         }
         if (b->before_search) { b->before_search(); b->before_search = nullptr; }
         if (R > 0) {
            if (sc) hipLaunchKernelGGL(k_search<true>, dim3((unsigned)n_wg), dim3(WG_THREADS), search_lds, st, A);
            else hipLaunchKernelGGL(k_search<false>, dim3((unsigned)n_wg), dim3(WG_THREADS), search_lds, st, A);
            HIP_TRY(hipGetLastError());
         }
         HIP_TRY(hipMemcpyAsync(host_cnt, cnt, 8 * CNT_WORDS, hipMemcpyDeviceToHost, st));
         GM_SYNC(b, st);
         GM_FAIL_IF(host_cnt[3] != 0, "search spill region overflowed (raise 'spill_nodes').");
         if (host_cnt[1] <= hit_cap) break;
         GM_FAIL_IF(attempt == 1, "hit list overflowed twice.");
         hit_cap = (size_t)host_cnt[1] + (size_t)n_waves * HIT_CHUNK;           // pieces are reserved whole, in another order next time
      }
      unsigned long long verify_steps = 0, verify_windows = 0;
      const int pos_bits = ctx->ix.sar_bits;
      const int read_bits = bits_for((uint64_t)std::max<int64_t>(R, 1));
      GM_FAIL_IF(pos_bits + 4 + read_bits > 64, "batch too large for the 64-bit sort key (split the batch).");
      b->key_pos_bits = pos_bits;
      HIP_TRY(hipEventRecord(b->ev[4], st));
      const bool verified = use_verify != 0 && host_cnt[1] > 0;
      int64_t *h_L = reinterpret_cast<int64_t *>(host_cnt + CNT_WORDS + CNT_SUMS);      // the suffixes of the list's intervals
      bool iv_counted = false, iv_count_back = false;              // ... are being added up on the run stream; the host has waited for it
      int64_t n_stage1 = 0, n_stage2[2] = {0, 0}, n_iv = 0;        // keys k_verify / the two launches of k_verify_group left, entries that are intervals
      unsigned int vblocks = 0, gblocks[2] = {0, 0};
      int vstage_per = 0;
      if (verified) {
         const int64_t H = (int64_t)host_cnt[1];
         GM_FAIL_IF(H >= ((int64_t)1 << 31), "too many entries in the hit list (split the batch).");
         VerifyArgs V;
         V.ix = ctx->ix; V.text = (const uint8_t *)ctx->d_text; V.text4 = (const uint8_t *)ctx->d_text4;
         V.reads4 = b->reads4.as<const uint8_t>();
         V.reads = b->reads.as<const uint8_t>(); V.row_len = b->row_len.as<const uint16_t>();
         V.phase_rec = b->phase_rec.as<const uint2>(); V.phase_lo = b->phase_lo.as<const uint8_t>();
         V.list = b->hits.as<const Node>(); V.n_list = H; V.stride = b->stride;
         V.key_shift = pos_bits + 4;
         const int vbd = 256;
         const bool pairs = use_verify >= 2;                             // two lanes per entry
         const bool groups = use_verify >= 4;                            // entries of three and four suffixes go to k_verify_group<4>
         const bool wides = use_verify >= 4;                             // those of five to eight (sixteen) to k_verify_group<8> (<16>); verify = 4: those of
                                                                         // three to k_verify_group<3> (21 entries per wave instead of 16 with a lane idle each)
         V.split3 = use_verify == 4;
         // entries a block takes: with pairs about 1.4 suffixes to replay per entry: rounds of 256 lanes, only the last one of a block partly filled
         const int vtile = ctx->verify_tile > 0 ? ctx->verify_tile : 512;
         V.tile = vtile; V.stage_per = pairs ? 2 * vtile : vtile;
         const dim3 vgrid((unsigned)cdiv(V.n_list, vtile));
         vblocks = vgrid.x; vstage_per = V.stage_per;
         HIP_TRY(b->vsteps.ensure(8 * (size_t)vgrid.x));
         V.n_ext = b->vsteps.as<unsigned long long>();
         // per block: a slot of keys, a slot of interval entries, slots of big candidates; their counts (+ one word behind each for the scan's totals)
         const size_t nb1 = (size_t)vgrid.x + 1;
         HIP_TRY(b->stage.ensure(8 * (size_t)V.stage_per * vgrid.x));
         HIP_TRY(b->iv_idx.ensure(4 * (size_t)vtile * vgrid.x));
         HIP_TRY(b->blk_cnt.ensure(4 * 4 * nb1));
         HIP_TRY(b->blk_off.ensure(4 * 4 * nb1));
         HIP_TRY(hipMemsetAsync(b->blk_cnt.p, 0, 4 * 4 * nb1, st));
         uint32_t *c_stage = b->blk_cnt.as<uint32_t>(), *c_iv = c_stage + nb1, *c_big = c_iv + nb1, *c_wide = c_big + nb1;
         uint32_t *o_stage = b->blk_off.as<uint32_t>(), *o_iv = o_stage + nb1, *o_big = o_iv + nb1, *o_wide = o_big + nb1;
         V.stage = b->stage.as<unsigned long long>(); V.stage_cnt = c_stage;
         V.iv_idx = b->iv_idx.as<uint32_t>(); V.iv_cnt = c_iv;
         V.big_idx = nullptr; V.big_cnt = nullptr; V.wide_idx = nullptr; V.wide_cnt = nullptr;
         if (groups) {
            HIP_TRY(b->big_idx.ensure(4 * (size_t)vtile * vgrid.x));
            V.big_idx = b->big_idx.as<uint32_t>(); V.big_cnt = c_big;
         }
         if (wides) {
            HIP_TRY(b->wide_idx.ensure(4 * (size_t)vtile * vgrid.x));
            V.wide_idx = b->wide_idx.as<uint32_t>(); V.wide_cnt = c_wide;
         }
#define GM_VERIFY(N) do { if (pairs) hipLaunchKernelGGL((k_verify<N, true>), vgrid, dim3(vbd), 0, st, V); \
                          else hipLaunchKernelGGL((k_verify<N, false>), vgrid, dim3(vbd), 0, st, V); } while (0)
         switch (cdiv(b->maxlen, 64)) {                                 // 64-letter pieces of the longest read
            case 1:  GM_VERIFY(1); break;
            case 2:  GM_VERIFY(2); break;
            case 3:  GM_VERIFY(3); break;
            default: GM_VERIFY(4); break;
         }
#undef GM_VERIFY
         hipLaunchKernelGGL(k_sum_steps, dim3(64), dim3(256), 0, st, b->vsteps.as<const unsigned long long>(), (int64_t)vgrid.x, cnt + CNT_WORDS);
         HIP_TRY(hipGetLastError());
         uint32_t *h_tot = reinterpret_cast<uint32_t *>(host_cnt + CNT_WORDS + CNT_SUMS + 1);   // running totals of the scan: keys, + intervals, + big, + wide; [4], [5]: keys of the groups
         for (int k = 0; k < 6; k++) h_tot[k] = 0;
         {
            size_t tmp = 0;
            const int items = 4 * (int)nb1;                     // the four count arrays lie one behind the other, each with a zero at its end:
            HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp, b->blk_cnt.as<uint32_t>(), b->blk_off.as<uint32_t>(), items, st));
            HIP_TRY(b->cub_tmp.ensure(tmp));                    // one scan; the offsets of the later ones are taken relative to their first
            HIP_TRY(hipcub::DeviceScan::ExclusiveSum(b->cub_tmp.p, tmp, b->blk_cnt.as<uint32_t>(), b->blk_off.as<uint32_t>(), items, st));
            HIP_TRY(hipMemcpyAsync(&h_tot[0], o_stage + vgrid.x, 4, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(&h_tot[1], o_iv + vgrid.x, 4, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(&h_tot[2], o_big + vgrid.x, 4, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(&h_tot[3], o_wide + vgrid.x, 4, hipMemcpyDeviceToHost, st));
         }
         HIP_TRY(hipMemcpyAsync(host_cnt, cnt, 8 * (CNT_WORDS + 64), hipMemcpyDeviceToHost, st));
         GM_SYNC(b, st);
         n_stage1 = h_tot[0];
         n_iv = (int64_t)h_tot[1] - (int64_t)h_tot[0];
         const int64_t n_class[2] = {(int64_t)h_tot[2] - (int64_t)h_tot[1], (int64_t)h_tot[3] - (int64_t)h_tot[2]};
         if (n_iv > 0) {
            HIP_TRY(b->iv_dense.ensure(4 * (size_t)n_iv));
            hipLaunchKernelGGL(k_list_compact, vgrid, dim3(64), 0, st, b->iv_idx.as<const uint32_t>(), (const uint32_t *)c_iv,
                               (const uint32_t *)o_iv, vtile, b->iv_dense.as<uint32_t>() - h_tot[0]);
         }
         // the two classes of big candidates: a dense list each, then k_verify_group with 4 / 16 lanes per entry
         bool any_group = false;
         for (int c = 0; c < 2; c++) {
            const int64_t n_big = n_class[c];
            n_stage2[c] = 0; gblocks[c] = 0;
            if (n_big <= 0) continue;
            any_group = true;
            const int G = c == 0 ? 4 : (use_verify == 4 ? 3 : use_verify);
            DevBuf &dense_buf = c == 0 ? b->big_dense : b->wide_dense;
            HIP_TRY(dense_buf.ensure(4 * (size_t)n_big));
            hipLaunchKernelGGL(k_list_compact, vgrid, dim3(64), 0, st, (c == 0 ? b->big_idx : b->wide_idx).as<const uint32_t>(),
                               (const uint32_t *)(c == 0 ? c_big : c_wide), (const uint32_t *)(c == 0 ? o_big : o_wide), vtile,
                               dense_buf.as<uint32_t>() - h_tot[1 + c]);
            const dim3 ggrid((unsigned)cdiv(n_big, (int64_t)(vbd / 64) * (64 / G)));
            gblocks[c] = ggrid.x;
            HIP_TRY(b->vsteps2[c].ensure(8 * (size_t)ggrid.x));
            HIP_TRY(b->stage2[c].ensure(8 * (size_t)STAGE_PER_BLOCK * ggrid.x));
            HIP_TRY(b->blk_cnt2[c].ensure(4 * ((size_t)ggrid.x + 1)));
            HIP_TRY(b->blk_off2[c].ensure(4 * ((size_t)ggrid.x + 1)));
            HIP_TRY(hipMemsetAsync(b->blk_cnt2[c].p, 0, 4 * ((size_t)ggrid.x + 1), st));
            VerifyArgs V2 = V;
            V2.stage = b->stage2[c].as<unsigned long long>(); V2.stage_cnt = b->blk_cnt2[c].as<uint32_t>();
#define GM_GROUP(N) do { if (c == 0) hipLaunchKernelGGL((k_verify_group<N, 4>), ggrid, dim3(vbd), 0, st, V2, dense_buf.as<const uint32_t>(), n_big, \
                                                          b->vsteps2[c].as<unsigned long long>()); \
                         else if (G == 3) hipLaunchKernelGGL((k_verify_group<N, 3>), ggrid, dim3(vbd), 0, st, V2, dense_buf.as<const uint32_t>(), n_big, \
                                                 b->vsteps2[c].as<unsigned long long>()); \
                         else if (G == 8) hipLaunchKernelGGL((k_verify_group<N, 8>), ggrid, dim3(vbd), 0, st, V2, dense_buf.as<const uint32_t>(), n_big, \
                                                 b->vsteps2[c].as<unsigned long long>()); \
                         else hipLaunchKernelGGL((k_verify_group<N, 16>), ggrid, dim3(vbd), 0, st, V2, dense_buf.as<const uint32_t>(), n_big, \
                                                 b->vsteps2[c].as<unsigned long long>()); } while (0)
            switch (cdiv(b->maxlen, 64)) {
               case 1:  GM_GROUP(1); break;
               case 2:  GM_GROUP(2); break;
               case 3:  GM_GROUP(3); break;
               default: GM_GROUP(4); break;
            }
#undef GM_GROUP
            hipLaunchKernelGGL(k_sum_steps, dim3(64), dim3(256), 0, st, b->vsteps2[c].as<const unsigned long long>(), (int64_t)ggrid.x,
                               cnt + CNT_WORDS + 64 * (1 + c));
            HIP_TRY(hipGetLastError());
            size_t tmp = 0;
            HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp, b->blk_cnt2[c].as<uint32_t>(), b->blk_off2[c].as<uint32_t>(), (int)ggrid.x + 1, st));
            HIP_TRY(b->cub_tmp.ensure(tmp));
            HIP_TRY(hipcub::DeviceScan::ExclusiveSum(b->cub_tmp.p, tmp, b->blk_cnt2[c].as<uint32_t>(), b->blk_off2[c].as<uint32_t>(), (int)ggrid.x + 1, st));
            HIP_TRY(hipMemcpyAsync(&h_tot[4 + c], b->blk_off2[c].as<uint32_t>() + ggrid.x, 4, hipMemcpyDeviceToHost, st));
         }
         // the intervals' suffixes are counted here already where the post stream still holds the chunk before (the locate
         // stage needs the number before it can launch anything): the count comes back with the group kernels' wait below
         // instead of after a wait of its own behind that stream's queue (see "What follows runs on the post stream" below)
         if (n_iv > 0 && !(result_flags & GM_RUN_NO_LOCATE) && b->st_post != nullptr && b->st_post != st && hipStreamQuery(b->st_post) == hipErrorNotReady) {
            GM_FAIL_IF(batch_count_intervals(b, st, n_iv, b->iv_dense.as<const uint32_t>(), h_L) != 0, "cannot count the intervals' suffixes.");
            iv_counted = true;
         }
         if (any_group) {
            HIP_TRY(hipMemcpyAsync(host_cnt + CNT_WORDS + 64, cnt + CNT_WORDS + 64, 8 * 128, hipMemcpyDeviceToHost, st));
            GM_SYNC(b, st);
            n_stage2[0] = h_tot[4]; n_stage2[1] = h_tot[5];
            iv_count_back = iv_counted;
         }
         for (int k = 0; k < (any_group ? CNT_SUMS : 64); k++) {
            const unsigned long long st_k = host_cnt[CNT_WORDS + k] & 0xffffffffull;
            host_cnt[2] += st_k; verify_steps += st_k; verify_windows += host_cnt[CNT_WORDS + k] >> 32;
         }
         b->n_big = n_class[0] + n_class[1];
      }
      HIP_TRY(hipEventRecord(b->ev[1], st));
      static const bool gate_timing = getenv("GPUMAP_GATE_TIMING") != nullptr;      // (debug aid: host time from the end of the search to the sort's launch)
      const auto t_searched = std::chrono::steady_clock::now();
      b->n_hits = (int64_t)host_cnt[1];            // entries of the list, holes included
      b->n_real = (int64_t)host_cnt[7];            // hits (and candidates handed to k_verify) among them
      b->n_ext = (int64_t)host_cnt[2];
      for (int i = 0; i < 4; i++) b->stats[i] = (int64_t)host_cnt[4 + i];
      b->stats[4] = (int64_t)verify_steps;
      b->stats[5] = (int64_t)host_cnt[1];
      b->spill_chunks = (int64_t)host_cnt[8];
#ifdef GM_STEP_HIST
      fprintf(stderr, "k_search steps by interval size: 2: %llu  3-4: %llu  5-16: %llu  17-256: %llu  >256: %llu  1: %llu\n",
              host_cnt[9], host_cnt[10], host_cnt[11], host_cnt[12], host_cnt[13], host_cnt[14]);
#endif
      b->verify_windows = (int64_t)verify_windows;

      // ---- K3 locate: the keys k_verify / k_verify_group left are made dense; the intervals (after a verified search
      // a short list of entries, else every entry of the list) get their slots by a scan of their sizes and read
      // their suffix-array entries
      const int64_t H = b->n_hits;
      const int64_t n_look = (result_flags & GM_RUN_NO_LOCATE) ? 0 : (verified ? n_iv : H);      // entries the interval path looks at
      const uint32_t *dense = verified ? b->iv_dense.as<const uint32_t>() : (const uint32_t *)nullptr;
      int64_t L_iv = 0;
      HIP_TRY(b->read_first.ensure(8 * (size_t)(R + 1)));
      // What follows runs on the post stream, which is one per context and may still hold the ordering of the chunk before.
      // Behind that the count of the intervals' suffixes came back 1.1 ms after the search's end at GRCh38 size, and the next
      // chunk's k_search, which waits for this chunk's sort to be enqueued (gm_batch_set_gate), that much later: where the
      // post stream was busy when k_verify had ended, the count has run on the run stream (above) and is back already.
      // Where it was idle the count runs there, beside the next chunk's encoding (with millions of intervals, 150 nt and
      // tau 4, it takes a quarter of a millisecond).
      // (The post stream never waits ON THE DEVICE for an event of the run stream: with priorities and chunks of a few reads
      // that stalled both for good, as the low-priority upload stream of round 2 did.  The host waits for the run stream.)
      bool drain_marked = false;
      if (n_look > 0) GM_FAIL_IF(H >= ((int64_t)1 << 31), "too many entries in the hit list (split the batch).");
      if (iv_counted) {
         if (!iv_count_back) GM_SYNC(b, st);
         L_iv = *h_L;
      }
      if (b->st_post != nullptr && b->st_post != st) {
         if (!iv_counted) GM_SYNC(b, st);
         st = b->st_post;
         HIP_TRY(hipEventRecord(b->ev_drain, st));
         drain_marked = true;
      }
      if (b->on_searched) { b->on_searched(); b->on_searched = nullptr; }      // (the host has waited for the run stream's last kernel: it is free)
      if (n_look > 0 && !iv_counted) {
         GM_FAIL_IF(batch_count_intervals(b, st, n_look, dense, h_L) != 0, "cannot count the intervals' suffixes.");
         GM_SYNC(b, st);
         L_iv = *h_L;
      }
      const auto t_counted = std::chrono::steady_clock::now();
      const int64_t L = n_stage1 + n_stage2[0] + n_stage2[1] + L_iv;
      b->n_locs = L;
      if (L > 0) {
         GM_FAIL_IF(L >= ((int64_t)1 << 31), "too many located hits in one batch (split the batch).");
         HIP_TRY(b->keys.ensure(8 * (size_t)L));
         HIP_TRY(b->keys_alt.ensure(8 * (size_t)L));
         if (n_stage1 > 0)
            hipLaunchKernelGGL(k_stage_compact, dim3(vblocks), dim3(256), 0, st, b->stage.as<const unsigned long long>(),
                               b->blk_cnt.as<const uint32_t>(), b->blk_off.as<const uint32_t>(), vstage_per, b->keys.as<unsigned long long>());
         for (int c = 0; c < 2; c++)
            if (n_stage2[c] > 0)
               hipLaunchKernelGGL(k_stage_compact, dim3(gblocks[c]), dim3(256), 0, st, b->stage2[c].as<const unsigned long long>(),
                                  b->blk_cnt2[c].as<const uint32_t>(), b->blk_off2[c].as<const uint32_t>(), STAGE_PER_BLOCK,
                                  b->keys.as<unsigned long long>() + n_stage1 + (c == 1 ? n_stage2[0] : 0));
         if (L_iv > 0) {
            unsigned long long *keys_iv = b->keys.as<unsigned long long>() + n_stage1 + n_stage2[0] + n_stage2[1];
            const unsigned int longer_cap = (unsigned int)(L_iv / PLACE_INLINE + 1);
            HIP_TRY(b->longer.ensure(4 * (size_t)longer_cap));
            unsigned int *n_longer = (unsigned int *)(cnt + CNT_WORDS + CNT_SUMS);
            HIP_TRY(hipMemsetAsync(n_longer, 0, 8, st));
            hipLaunchKernelGGL(k_place, dim3((unsigned)cdiv(n_look, 256)), dim3(256), 0, st, ctx->ix, b->hits.as<const Node>(), dense, n_look,
                               b->hit_sizes.as<const int64_t>(), b->hit_off.as<const int64_t>(), pos_bits, keys_iv,
                               b->longer.as<unsigned int>(), n_longer, longer_cap, sc ? b->score_add.as<const uint8_t>() : (const uint8_t *)nullptr);
            hipLaunchKernelGGL(k_place_long, dim3((unsigned)(4 * ctx->cu_count)), dim3(256), 0, st, ctx->ix, b->hits.as<const Node>(), dense,
                               b->hit_sizes.as<const int64_t>(), b->hit_off.as<const int64_t>(), pos_bits, keys_iv,
                               b->longer.as<const unsigned int>(), (const unsigned int *)n_longer,
                               sc ? b->score_add.as<const uint8_t>() : (const uint8_t *)nullptr);
         }
         HIP_TRY(hipGetLastError());
      }
      HIP_TRY(hipEventRecord(b->ev[2], st));
      // ---- order: reads in input order, text positions ascending inside a read
      if (L == 0) HIP_TRY(hipMemsetAsync(b->read_first.p, 0, 8 * (size_t)(R + 1), st));
      b->have_lines = false; b->line_bytes = 0;
      bool packed_by_order = false;      // the ordering's last kernel has written the packed hits
      if (L == 0 && (result_flags & GM_RESULT_LINES)) b->have_lines = true;
      if (L > 0) {
         hipcub::DoubleBuffer<unsigned long long> db(b->keys.as<unsigned long long>(), b->keys_alt.as<unsigned long long>());
         size_t tmp = 0;
         // A pass of the sort takes 8 bits.  Where leaving out the score and up to three bits of the position saves a
         // pass (GRCh38 size with 12.5 M reads: 61 bits, 56 without the score and one bit), the sort starts above
         // them and k_order_runs puts the short runs of keys that agree above in order.
         const int end_bit = pos_bits + 4 + read_bits;
         // Round 3: where nearly every read has one hit (texts without repeats: L <= 1.25 R) the radix sort runs over the
         // read's bits only (3 passes for 12.5 M reads instead of 7 over the key) and k_order_first / k_order_big put the
         // hits of the reads that have several in order where they lie.  Measured where reads have many hits that loses:
         // at GRCh38 size 98.7 % of the reads have one hit but 70 % of the hits belong to the 0.8 % of the reads with 65 ..
         // 1024 hits each (profiles/r3_hits_hist.jsonl): ordering those 100 000 runs of hundreds of keys costs more than
         // the four passes saved (order 1.97 -> 4.0 ms; chr21 size 4.7 -> 11.6 ms).  There the sort runs over the key, leaving
         // out the score (the text position determines it) and up to three low bits of the position where that saves a
         // pass; k_order_runs / k_read_first order the few keys that agree above.  Option "sort_all_bits": every bit.
         // (end_bit = 64 with begin_bit > 0: hipcub returns unsorted keys for 10^4 .. 10^6 items, tools/sortpairs_check.hip)
         int begin_bit = 0;
         const bool by_read = !ctx->sort_all_bits && end_bit < 64 && (ctx->sort_by_read == 1 || (ctx->sort_by_read < 0 && L <= R + R / 4));
         if (by_read) begin_bit = pos_bits + 4;
         // Second form of round 3 (sort_by_read = 2; the default where reads have up to six hits on average): two
         // passes over the key's top 16 bits bring the hits of a segment of reads together (256 reads where a chunk has
         // 2^24), and k_order_segments orders every segment in LDS by the rest of the key: the keys cross the memory 3 times
         // instead of 7.  Measured at GRCh38 size (850 keys per segment): order 1.97 -> 1.66 ms, 774-782 -> 798-801 M reads/s
         // in the pipeline.  (A first comparison in the pipeline said the opposite; it was made before the pipeline held the
         // next k_search back until the sort is enqueued, when a run got one of two regimes by chance.)
         // chr21 size (12 hits per read on average, thousands for some: the hits of one read cannot be split, and segments beyond
         // the 8 192 keys a workgroup takes go through the other buffer): 4.67 -> 5.2 .. 6.9 ms with 16 .. 20 bits: not there.
         // 4 M reads of 150 nt with 14 bits: 0.65 -> 0.60 ms.
         // (as many of the key's top bits as make segments of about 768 keys: 16 at GRCh38 size, 18 at chr21 size, where a read
         // has 12 hits on average)
         int seg_bits = std::min(read_bits, 8);
         while (seg_bits < read_bits && (L >> seg_bits) > 768) seg_bits++;
         if (getenv("GPUMAP_SEG_BITS")) seg_bits = std::max(1, std::min(read_bits, atoi(getenv("GPUMAP_SEG_BITS"))));      // (tuning aid)
         const int seg_shift = end_bit - seg_bits;
         const int64_t n_seg = ((std::max<int64_t>(R, 1) - 1) >> (read_bits - seg_bits)) + 1;
         const bool by_seg = !by_read && !ctx->sort_all_bits && end_bit < 64 &&
                             (ctx->sort_by_read == 2 || (ctx->sort_by_read < 0 && n_seg >= 4096 && L <= 6 * R));
         if (by_seg) {
            int lo = 4;
            for (int drop = 1; drop <= 3 && lo == 4; drop++)
               if (cdiv(seg_shift - (4 + drop), 8) < cdiv(seg_shift - 4, 8)) lo = 4 + drop;
            HIP_TRY(hipcub::DeviceRadixSort::SortKeys(nullptr, tmp, db, (int)L, seg_shift, end_bit, st));
            HIP_TRY(b->cub_tmp.ensure(tmp));
            HIP_TRY(hipcub::DeviceRadixSort::SortKeys(b->cub_tmp.p, tmp, db, (int)L, seg_shift, end_bit, st));
            HIP_TRY(b->seg_first.ensure(4 * (size_t)(n_seg + 1)));
            hipLaunchKernelGGL(k_seg_first, dim3((unsigned)cdiv(L, 256)), dim3(256), 0, st, (const unsigned long long *)db.Current(), L, seg_shift, n_seg,
                               b->seg_first.as<uint32_t>());
            // (the workgroup of a segment also orders the keys that agree from `lo` up and writes the CSR offsets of its reads)
            SegOut so;
            so.read_first = b->read_first.as<int64_t>(); so.n_reads = R; so.n_keys = L; so.read_shift = pos_bits + 4; so.seg_reads_log = read_bits - seg_bits;
            so.packed = nullptr;
            if ((result_flags & GM_RESULT_PACKED) && compact && !(result_flags & GM_RESULT_LINES)) {
               // packed results: the segments' workgroups write the hits as they leave (5 bytes each) instead of the sorted keys
               HIP_TRY(b->packed.ensure(20 * (size_t)cdiv(std::max<int64_t>(L, 1), 4) + 16));
               so.packed = b->packed.as<uint8_t>();
               packed_by_order = true;
            }
            hipLaunchKernelGGL(k_order_segments<4>, dim3((unsigned)n_seg), dim3(256), 0, st, db.Current(), db.Alternate(), b->seg_first.as<const uint32_t>(),
                               lo, seg_shift, -1, false, so);
            hipLaunchKernelGGL(k_order_segments<8>, dim3((unsigned)n_seg), dim3(256), 0, st, db.Current(), db.Alternate(), b->seg_first.as<const uint32_t>(),
                               lo, seg_shift, 1024, false, so);
            hipLaunchKernelGGL(k_order_segments<16>, dim3((unsigned)n_seg), dim3(256), 0, st, db.Current(), db.Alternate(), b->seg_first.as<const uint32_t>(),
                               lo, seg_shift, 2048, false, so);
            hipLaunchKernelGGL(k_order_segments<32>, dim3((unsigned)n_seg), dim3(256), 0, st, db.Current(), db.Alternate(), b->seg_first.as<const uint32_t>(),
                               lo, seg_shift, 4096, true, so);
            b->sorted_keys = db.Current();
            begin_bit = 0;
         } else {
            for (int drop = 0; drop <= 3 && begin_bit == 0; drop++)
               if (cdiv(end_bit - (4 + drop), 8) < cdiv(end_bit, 8) && !ctx->sort_all_bits && end_bit < 64) begin_bit = 4 + drop;
            HIP_TRY(hipcub::DeviceRadixSort::SortKeys(nullptr, tmp, db, (int)L, begin_bit, end_bit, st));
            HIP_TRY(b->cub_tmp.ensure(tmp));
            HIP_TRY(hipcub::DeviceRadixSort::SortKeys(b->cub_tmp.p, tmp, db, (int)L, begin_bit, end_bit, st));
            b->sorted_keys = db.Current();
         }
         if (gate_timing)
            fprintf(stderr, "[gate] batch %p: %.0f us from the search's end to the sort's launch (%.0f us until the number of located hits was known)\n", (void *)b,
                    std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_searched).count(),
                    std::chrono::duration<double, std::micro>(t_counted - t_searched).count());
         // Where a chunk has many hits per read (chr21 size: locate and ordering take 6 ms, more than the next chunk's search
         // leaves room for) the next k_search also waits until the post stream has finished the chunk before this one: without
         // that the post stream falls behind by more every step until the pipe's slots run out, and results come in bursts.
         static const int drain_rule = getenv("GPUMAP_PIPE_DRAIN") ? atoi(getenv("GPUMAP_PIPE_DRAIN")) : -1;      // (tuning aid: 0 never, 1 always)
         if (b->on_ordering && drain_marked && (drain_rule < 0 ? L > 6 * R : drain_rule != 0)) HIP_TRY(hipEventSynchronize(b->ev_drain));
         if (b->on_ordering) { b->on_ordering(); b->on_ordering = nullptr; }
         if (by_read) {
            const int64_t ob = cdiv(L, 256);
            HIP_TRY(b->big_start.ensure(4 * (size_t)ORDER_SLOT * (size_t)ob));
            HIP_TRY(b->big_cnt.ensure(4 * (size_t)ob));
            hipLaunchKernelGGL(k_order_first, dim3((unsigned)ob), dim3(256), 0, st, (const unsigned long long *)db.Current(), db.Alternate(), L, begin_bit,
                               b->big_start.as<uint32_t>(), b->big_cnt.as<uint32_t>());
            hipLaunchKernelGGL(k_order_big, dim3((unsigned)std::min<int64_t>(ob, 8 * (int64_t)ctx->cu_count)), dim3(256), 0, st, db.Alternate(),
                               db.Current(), L, begin_bit, 4, b->big_start.as<const uint32_t>(), b->big_cnt.as<const uint32_t>(), ob);
            b->sorted_keys = db.Alternate();                 // (the keys are in order there from here on)
            begin_bit = 0;
         }
         unsigned long long *sk = const_cast<unsigned long long *>(b->sorted_keys);
         if (result_flags & GM_RESULT_LINES) {
            // the lines mapper() prints, made where the keys lie (gm_parse.hpp)
            GM_FAIL_IF(!b->from_file, "hit lines need a batch that was filled from a block of a read file (its tags).");
            GM_FAIL_IF(ctx->d_names.p == nullptr, "hit lines need the sequences' names (gpumap_set_seq_names).");
            if (begin_bit > 0) hipLaunchKernelGGL(k_order_runs, dim3((unsigned)cdiv(L, 256)), dim3(256), 0, st, sk, L, begin_bit);
            HIP_TRY(b->line_len.ensure(8 * (size_t)(L + 1)));
            FormatArgs F;
            F.keys = b->sorted_keys; F.n_keys = L; F.pos_bits = pos_bits; F.n = ctx->ix.n; F.seq_beg = ctx->ix.seq_beg; F.seq_cnt = ctx->ix.seq_cnt;
            F.names = ctx->d_names.as<const uint8_t>(); F.name_off = ctx->d_name_off.as<const uint32_t>();
            F.raw = b->up_src.as<const uint8_t>();
            F.tag_off = b->lines_per_rec > 1 ? b->tag_off.as<const uint32_t>() : nullptr;
            F.tag_len = b->lines_per_rec > 1 ? b->tag_len.as<const uint32_t>() : nullptr;
            F.first_line = b->first_line; F.line_len = b->line_len.as<int64_t>(); F.out = nullptr;
            hipLaunchKernelGGL(k_format_len, dim3((unsigned)cdiv(L + 1, 256)), dim3(256), 0, st, F);
            size_t t2 = 0;
            HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, t2, F.line_len, F.line_len, (int)(L + 1), st));
            HIP_TRY(b->cub_tmp.ensure(t2));
            HIP_TRY(hipcub::DeviceScan::ExclusiveSum(b->cub_tmp.p, t2, F.line_len, F.line_len, (int)(L + 1), st));
            int64_t *h_bytes = reinterpret_cast<int64_t *>(host_cnt + CNT_WORDS + CNT_SUMS);
            HIP_TRY(hipMemcpyAsync(h_bytes, F.line_len + L, 8, hipMemcpyDeviceToHost, st));
            GM_SYNC(b, st);
            b->line_bytes = *h_bytes;
            HIP_TRY(b->lines.ensure((size_t)b->line_bytes + 64));
            F.out = b->lines.as<uint8_t>();
            hipLaunchKernelGGL(k_format_write, dim3((unsigned)cdiv(L, 256)), dim3(256), 0, st, F);
            b->have_lines = true;
         } else if (compact) {
            if (!by_seg)
               hipLaunchKernelGGL(k_read_first, dim3((unsigned)cdiv(L, 256)), dim3(256), 0, st, sk, L, pos_bits, R,
                                  b->read_first.as<int64_t>(), begin_bit);
         } else {
            if (begin_bit > 0) hipLaunchKernelGGL(k_order_runs, dim3((unsigned)cdiv(L, 256)), dim3(256), 0, st, sk, L, begin_bit);
            for (DevBuf *x : {&b->loc_read, &b->loc_pos, &b->loc_fwd, &b->loc_seq, &b->loc_seqpos}) HIP_TRY(x->ensure(8 * (size_t)L));
            HIP_TRY(b->loc_strand.ensure((size_t)L));
            HIP_TRY(b->loc_score.ensure((size_t)L));
            hipLaunchKernelGGL(k_unpack_locs, dim3((unsigned)cdiv(L, 256)), dim3(256), 0, st, ctx->ix,
                               b->sorted_keys, L, pos_bits, R, b->read_first.as<int64_t>(), b->loc_read.as<int64_t>(),
                               b->loc_pos.as<int64_t>(), b->loc_fwd.as<int64_t>(), b->loc_seq.as<int64_t>(),
                               b->loc_seqpos.as<int64_t>(), b->loc_strand.as<uint8_t>(), b->loc_score.as<uint8_t>());
         }
         HIP_TRY(hipGetLastError());
      }
      b->have_packed = false;
      if ((result_flags & GM_RESULT_PACKED) && compact && !(result_flags & GM_RESULT_LINES)) {
         // the packed result: what the link carries, made from the sorted keys, the offsets and the status
         HIP_TRY(b->packed.ensure(20 * (size_t)cdiv(std::max<int64_t>(L, 1), 4) + 16));
         HIP_TRY(b->first32.ensure(4 * (size_t)(R + 1)));
         HIP_TRY(b->status8.ensure((size_t)std::max<int64_t>(R, 1)));
         const int64_t threads = packed_by_order ? R + 1 : std::max<int64_t>(cdiv(L, 4), R + 1);
         hipLaunchKernelGGL(k_pack_results, dim3((unsigned)cdiv(threads, 256)), dim3(256), 0, st,
                            packed_by_order ? (const unsigned long long *)nullptr : b->sorted_keys, L, pos_bits + 4,
                            b->read_first.as<const int64_t>(), b->status.as<const int64_t>(), R, b->packed.as<uint32_t>(),
                            b->first32.as<uint32_t>(), b->status8.as<uint8_t>());
         HIP_TRY(hipGetLastError());
         b->have_packed = true;
      }
      if (want_intervals && H > 0) {
         GM_FAIL_IF(RS >= ((int64_t)1 << 24) - 1, "interval output supports fewer than 2^24 - 1 reads per batch.");
         GM_FAIL_IF(H >= ((int64_t)1 << 31), "too many interval hits in one batch.");
         HIP_TRY(b->hkeys.ensure(8 * (size_t)H)); HIP_TRY(b->hkeys_alt.ensure(8 * (size_t)H));
         HIP_TRY(b->hidx.ensure(4 * (size_t)H)); HIP_TRY(b->hidx_alt.ensure(4 * (size_t)H));
         hipLaunchKernelGGL(k_hit_keys, dim3((unsigned)cdiv(H, 256)), dim3(256), 0, st, b->hits.as<const Node>(), H,
                            b->hkeys.as<unsigned long long>(), b->hidx.as<unsigned int>(), plan ? b->sc_alias.as<const uint32_t>() : (const uint32_t *)nullptr);
         hipcub::DoubleBuffer<unsigned long long> dk(b->hkeys.as<unsigned long long>(), b->hkeys_alt.as<unsigned long long>());
         hipcub::DoubleBuffer<unsigned int> dv(b->hidx.as<unsigned int>(), b->hidx_alt.as<unsigned int>());
         size_t tmp = 0;
         // gpumap_ann_build (GM_RUN_NO_LOCATE) only needs the hits of a query together, in any order: the query's bits alone,
         // 3 passes instead of 8 (end_bit 63, not 64: see k_hit_keys)
         const bool by_query = (result_flags & GM_RUN_NO_LOCATE) != 0;
         const int sort_lo = by_query ? 39 : 0, sort_hi = by_query ? 63 : 64;
         HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp, dk, dv, (int)H, sort_lo, sort_hi, st));
         HIP_TRY(b->cub_tmp.ensure(tmp));
         HIP_TRY(hipcub::DeviceRadixSort::SortPairs(b->cub_tmp.p, tmp, dk, dv, (int)H, sort_lo, sort_hi, st));
         for (DevBuf *x : {&b->hit_read, &b->hit_fp, &b->hit_rp, &b->hit_sz, &b->hit_dp, &b->hit_score}) HIP_TRY(x->ensure(8 * (size_t)H));
         HIP_TRY(b->hit_align.ensure(32 * (size_t)H));
         // the holes of the list sorted last: the first n_real entries are the hits
         hipLaunchKernelGGL(k_unpack_hits, dim3((unsigned)cdiv(std::max<int64_t>(b->n_real, 1), 256)), dim3(256), 0, st, b->hits.as<const Node>(),
                            (const unsigned int *)dv.Current(), b->n_real, b->reads.as<const uint8_t>(), b->read_len.as<const uint16_t>(),
                            b->stride, b->hit_read.as<int64_t>(), b->hit_fp.as<int64_t>(), b->hit_rp.as<int64_t>(),
                            b->hit_sz.as<int64_t>(), b->hit_dp.as<int64_t>(), b->hit_score.as<int64_t>(), b->hit_align.as<uint64_t>(),
                            sc ? b->score_add.as<const uint8_t>() : (const uint8_t *)nullptr, plan ? b->sc_alias.as<const uint32_t>() : (const uint32_t *)nullptr);
         HIP_TRY(hipGetLastError());
      }
      b->have_intervals = want_intervals != 0;
      HIP_TRY(hipEventRecord(b->ev[3], st));
      GM_SYNC(b, st);
      HIP_TRY(hipEventElapsedTime(&b->ms[0], b->ev[0], b->ev[1]));
      HIP_TRY(hipEventElapsedTime(&b->ms[1], b->ev[1], b->ev[2]));
      HIP_TRY(hipEventElapsedTime(&b->ms[2], b->ev[2], b->ev[3]));
      HIP_TRY(hipEventElapsedTime(&b->ms[3], b->ev[0], b->ev[3]));
      HIP_TRY(hipEventElapsedTime(&b->ms_encode, b->ev[5], b->ev[0]));
      {
         float ms_k = 0, ms_v = 0;
         HIP_TRY(hipEventElapsedTime(&ms_k, b->ev[0], b->ev[4]));
         HIP_TRY(hipEventElapsedTime(&ms_v, b->ev[4], b->ev[1]));
         b->stats[6] = (int64_t)(1000.0f * ms_k);
         b->stats[7] = (int64_t)(1000.0f * ms_v);
      }
      b->ran = true;
   }
   return 0;
failure_return:
   return -1;
}

static int result_flags_of(const gpumap_ctx *ctx)
{
   return (ctx->pinned_results ? GM_RESULT_PINNED : 0) | (ctx->compact_results ? GM_RESULT_COMPACT : 0);
}

extern "C" int gpumap_batch_run(gpumap_ctx *ctx, gpumap_batch *b, int tau, int want_intervals)
{
   GM_FAIL_IF(ctx == nullptr || b == nullptr, "NULL argument.");
   return gm_batch_run(ctx, b, tau, want_intervals, result_flags_of(ctx));
failure_return:
   return -1;
}

extern "C" int gpumap_batch_counters(const gpumap_batch *b, int64_t *n_hits, int64_t *n_locs, int64_t *n_ext, float ms[4])
{
   GM_FAIL_IF(b == nullptr, "argument 'batch' is NULL.");
   GM_FAIL_IF(!b->ran, "batch has not been run.");
   if (n_hits) *n_hits = b->n_real;
   if (n_locs) *n_locs = b->n_locs;
   if (n_ext) *n_ext = b->n_ext;
   if (ms) for (int i = 0; i < 4; i++) ms[i] = b->ms[i];
   return 0;
failure_return:
   return -1;
}

extern "C" int gpumap_batch_stats(const gpumap_batch *b, int64_t stats[8])
{
   GM_FAIL_IF(b == nullptr || stats == nullptr, "NULL argument.");
   GM_FAIL_IF(!b->ran, "batch has not been run.");
   for (int i = 0; i < 8; i++) stats[i] = b->stats[i];
   return 0;
failure_return:
   return -1;
}

extern "C" int gpumap_batch_stats_ext(const gpumap_batch *b, int64_t stats[8])
{
   GM_FAIL_IF(b == nullptr || stats == nullptr, "NULL argument.");
   GM_FAIL_IF(!b->ran, "batch has not been run.");
   stats[0] = b->attempts;
   stats[1] = (int64_t)(1000.0f * b->ms_encode);
   stats[2] = b->compact ? 1 : 0;
   stats[3] = (int64_t)(b->hits.cap / sizeof(Node));
   stats[4] = b->spill_chunks;
   stats[5] = b->verify_windows;
   stats[6] = b->n_big;
   stats[7] = 0;
   return 0;
failure_return:
   return -1;
}

extern "C" void *gpumap_host_alloc(size_t bytes)
{
   void *p = nullptr;
   if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
      GM_WARN("cannot allocate page-locked host memory.");
      errno = 1;
      return nullptr;
   }
   return p;
}

extern "C" void gpumap_host_free(void *p) { if (p) (void)hipHostFree(p); }

extern "C" void gpumap_hits_free(gpumap_hits *h)
{
   if (!h) return;
   if (!h->arena) {
      free(h->loc_packed); free(h->read_first32); free(h->status8);
      free(h->status); free(h->read_first_loc); free(h->loc_read); free(h->loc_pos); free(h->loc_fwd); free(h->loc_seq);
      free(h->loc_seqpos); free(h->loc_strand); free(h->loc_score); free(h->loc_key);
      free(h->hit_read); free(h->hit_fp); free(h->hit_rp); free(h->hit_sz); free(h->hit_dp); free(h->hit_score); free(h->hit_align);
   } else if (h->expanded) {          // gpumap_hits_expand malloc'ed these beside an arena-backed result
      if (h->expanded & 1) { free(h->loc_read); free(h->loc_pos); free(h->loc_fwd); free(h->loc_seq); free(h->loc_seqpos); free(h->loc_strand); free(h->loc_score); }
      if (h->expanded & 2) { free(h->status); free(h->read_first_loc); free(h->loc_key); }
   }
   memset(h, 0, sizeof *h);
}

namespace {
// Destination of one fetch: malloc'ed arrays, or slices of a page-locked arena of the batch.
struct FetchDst {
   char *arena = nullptr;
   size_t used = 0;
   template <class T> T *take(int64_t count)
   {
      size_t bytes = sizeof(T) * (size_t)std::max<int64_t>(count, 1);
      if (!arena) return (T *)malloc(bytes);
      T *p = (T *)(arena + used);
      used += (bytes + 255) & ~(size_t)255;
      return p;
   }
};
template <class T>
int fetch_array(FetchDst &d, T **dst, const void *src, int64_t count, hipStream_t st)
{
   *dst = d.take<T>(count);
   if (!*dst) return -1;
   // (the runtime's copy, a kernel over the whole device on this platform.  A copy kernel of our own on 16..128
   // workgroups holds the link's 56 GB/s alone, tools/copy_bench.hip, but beside a running k_search it gets too
   // small a share of the CUs' memory queues: chunk time 21.5 -> 25.4 .. 27.6 ms at GRCh38 size.)
   if (count > 0 && hipMemcpyAsync(*dst, src, sizeof(T) * (size_t)count, hipMemcpyDeviceToHost, st) != hipSuccess) return -1;
   return 0;
}
}  // namespace

int gm_batch_fetch(gpumap_ctx *ctx, gpumap_batch *b, gpumap_hits *out, int result_flags)
{
   GM_FAIL_IF(ctx == nullptr || b == nullptr || out == nullptr, "NULL argument.");
   GM_FAIL_IF(!b->ran, "batch has not been run.");
   memset(out, 0, sizeof *out);
   HIP_TRY(hipSetDevice(ctx->device));
   {
      hipStream_t st = b->st_copy;                   // (the run has ended with a wait for its last kernel)
      const int64_t R = b->n_reads, L = b->n_locs, H = b->have_intervals ? b->n_real : 0;
      const bool compact = b->compact;               // what the run left on the device decides
      FetchDst d;
      if (result_flags & GM_RESULT_PINNED) {
         size_t need = 16 * 256 + 8 * (size_t)(2 * R + 1) + (compact ? 8 : 42) * (size_t)L + 80 * (size_t)H;
         if (result_flags & GM_RESULT_LINES) need = 1024 + (size_t)b->line_bytes;
         else if (result_flags & GM_RESULT_PACKED) need = 16 * 256 + 5 * (size_t)R + 5 * (size_t)L + 64;
         // two arenas in turn: what a fetch hands out stays valid until the fetch after the next one on this batch
         // (the pipeline releases a slot's result before the slot is used again: one arena there)
         const int k = (result_flags & GM_RESULT_ONE_ARENA) ? 0 : b->pin_turn;
         b->pin_turn ^= 1;
         if (need > b->pin_cap[k]) {
            if (b->pin_arena[k]) { gm_pinned_give(b->pin_arena[k], b->pin_cap[k]); b->pin_arena[k] = nullptr; b->pin_cap[k] = 0; }
            need += need / 4;
            size_t got = 0;
            b->pin_arena[k] = gm_pinned_take(need, &got);        // (page-locked blocks go round: gm_reads.cpp)
            GM_FAIL_IF(b->pin_arena[k] == nullptr, "cannot allocate the result arena.");
            b->pin_cap[k] = got;
         }
         d.arena = (char *)b->pin_arena[k];
         out->arena = b->pin_arena[k];
      }
      out->n_reads = R; out->n_locs = L; out->n_hits = H; out->n_ext = b->n_ext;
      out->ms_search = b->ms[0]; out->ms_locate = b->ms[1]; out->ms_sort = b->ms[2]; out->ms_total = b->ms[3];
      out->key_pos_bits = b->key_pos_bits;
      int rc = 0;
      if (result_flags & GM_RESULT_LINES) {
         // the hit lines and nothing else cross the link
         GM_FAIL_IF(!b->have_lines, "the batch was not run for hit lines.");
         GM_FAIL_IF(!(result_flags & GM_RESULT_PINNED), "hit lines come in a page-locked arena.");
         out->line_bytes = b->line_bytes;
         rc |= fetch_array(d, &out->line_text, b->lines.p, b->line_bytes, st);
         GM_SYNC(b, st);
         GM_FAIL_IF(rc != 0, "memory error.");
         return 0;
      }
      if (result_flags & GM_RESULT_PACKED) {
         GM_FAIL_IF(!b->have_packed, "the batch was not run for packed results.");
         rc |= fetch_array(d, &out->status8, b->status8.p, R, st);
         rc |= fetch_array(d, &out->read_first32, b->first32.p, R + 1, st);
         rc |= fetch_array(d, &out->loc_packed, b->packed.p, 5 * L, st);
         GM_SYNC(b, st);
         GM_FAIL_IF(rc != 0, "memory error.");
         return 0;
      }
      rc |= fetch_array(d, &out->status, b->status.p, R, st);
      rc |= fetch_array(d, &out->read_first_loc, b->read_first.p, R + 1, st);
      if (compact) {
         rc |= fetch_array(d, &out->loc_key, (const void *)b->sorted_keys, L, st);
      } else {
         rc |= fetch_array(d, &out->loc_read, b->loc_read.p, L, st);
         rc |= fetch_array(d, &out->loc_pos, b->loc_pos.p, L, st);
         rc |= fetch_array(d, &out->loc_fwd, b->loc_fwd.p, L, st);
         rc |= fetch_array(d, &out->loc_seq, b->loc_seq.p, L, st);
         rc |= fetch_array(d, &out->loc_seqpos, b->loc_seqpos.p, L, st);
         rc |= fetch_array(d, &out->loc_strand, b->loc_strand.p, L, st);
         rc |= fetch_array(d, &out->loc_score, b->loc_score.p, L, st);
      }
      if (b->have_intervals) {
         rc |= fetch_array(d, &out->hit_read, b->hit_read.p, H, st);
         rc |= fetch_array(d, &out->hit_fp, b->hit_fp.p, H, st);
         rc |= fetch_array(d, &out->hit_rp, b->hit_rp.p, H, st);
         rc |= fetch_array(d, &out->hit_sz, b->hit_sz.p, H, st);
         rc |= fetch_array(d, &out->hit_dp, b->hit_dp.p, H, st);
         rc |= fetch_array(d, &out->hit_score, b->hit_score.p, H, st);
         rc |= fetch_array(d, &out->hit_align, b->hit_align.p, 4 * H, st);
      }
      GM_SYNC(b, st);
      GM_FAIL_IF(rc != 0, "memory error.");
   }
   return 0;
failure_return:
   gpumap_hits_free(out);
   return -1;
}

extern "C" int gpumap_batch_fetch(gpumap_ctx *ctx, gpumap_batch *b, gpumap_hits *out)
{
   GM_FAIL_IF(ctx == nullptr || b == nullptr || out == nullptr, "NULL argument.");
   return gm_batch_fetch(ctx, b, out, result_flags_of(ctx));
failure_return:
   return -1;
}

// The structure of arrays of a compact result, derived on the host from the sorted keys by the rule of
// txt_pos_to_str (src/index_txt.c:492-505) and seq_bisect (:770-787): what k_unpack_locs computes on the device.
extern "C" int gpumap_hits_expand(const gpumap_ctx *ctx, gpumap_hits *h)
{
   GM_FAIL_IF(ctx == nullptr || h == nullptr, "NULL argument.");
   if (h->loc_packed != nullptr && h->loc_key == nullptr) {
      // a packed result: status, offsets and keys first (the read of a hit is the one whose offsets hold it)
      const int64_t L = h->n_locs, R = h->n_reads;
      const int bits = h->key_pos_bits;
      h->status = (int64_t *)malloc(8 * (size_t)std::max<int64_t>(R, 1));
      h->read_first_loc = (int64_t *)malloc(8 * (size_t)(R + 1));
      h->loc_key = (uint64_t *)malloc(8 * (size_t)std::max<int64_t>(L, 1));
      if (h->arena) h->expanded |= 2;
      GM_FAIL_IF(!h->status || !h->read_first_loc || !h->loc_key, "memory error.");
      const int T = gm_host_threads();
      gm_parallel_for(T, [&](int t) {
         for (int64_t r = R * t / T; r < R * (t + 1) / T; r++) {
            h->status[r] = h->status8[r] ? -1 : 0;
            h->read_first_loc[r] = (int64_t)h->read_first32[r];
            for (int64_t i = h->read_first32[r]; i < (int64_t)h->read_first32[r + 1]; i++)
               h->loc_key[i] = ((uint64_t)r << (bits + 4)) | GPUMAP_PACKED_AT(h->loc_packed, i);
         }
      });
      h->read_first_loc[R] = (int64_t)h->read_first32[R];
   }
   if (h->loc_key == nullptr || h->loc_read != nullptr) return 0;       // nothing to do
   {
      const int64_t L = h->n_locs, n = ctx->ix.n;
      const int bits = h->key_pos_bits;
      const size_t cnt = (size_t)std::max<int64_t>(L, 1);
      h->loc_read = (int64_t *)malloc(8 * cnt); h->loc_pos = (int64_t *)malloc(8 * cnt); h->loc_fwd = (int64_t *)malloc(8 * cnt);
      h->loc_seq = (int64_t *)malloc(8 * cnt); h->loc_seqpos = (int64_t *)malloc(8 * cnt);
      h->loc_strand = (uint8_t *)malloc(cnt); h->loc_score = (uint8_t *)malloc(cnt);
      h->expanded |= 1;
      GM_FAIL_IF(!h->loc_read || !h->loc_pos || !h->loc_fwd || !h->loc_seq || !h->loc_seqpos || !h->loc_strand || !h->loc_score, "memory error.");
      const std::vector<int64_t> &beg = ctx->seq_beg_host;
      const int T = gm_host_threads();
      gm_parallel_for(T, [&](int t) {
         for (int64_t i = L * t / T; i < L * (t + 1) / T; i++) {
            const uint64_t k = h->loc_key[i];
            const int64_t pos = GPUMAP_KEY_POS(k, bits);
            const int strand = pos >= n / 2;
            const int64_t f = strand ? n - 2 - pos : pos;
            int64_t lo = 0, hi = (int64_t)beg.size();
            while (hi - lo >= 2) { const int64_t mid = (lo + hi) / 2; if (f < beg[(size_t)mid]) hi = mid; else lo = mid; }
            h->loc_read[i] = GPUMAP_KEY_READ(k, bits); h->loc_pos[i] = pos; h->loc_score[i] = (uint8_t)GPUMAP_KEY_SCORE(k);
            h->loc_strand[i] = (uint8_t)strand; h->loc_fwd[i] = f; h->loc_seq[i] = lo; h->loc_seqpos[i] = f - beg[(size_t)lo] + 1;
         }
      });
   }
   return 0;
failure_return:
   return -1;
}

// blocksc_trail for a batch of queries (k-mers): src/blocksearch.c:7-103 with trail = 0 for every query, the path array
// its callers hand over computed on the device (k_sc_prepare).  Interval hits and located hits, as gpumap_map_batch
// with want_intervals gives them for blocksearch_trail_rec.
extern "C" int gpumap_blocksc_batch(gpumap_ctx *ctx, const uint8_t *codes, const int64_t *offsets, int64_t n_queries,
                                    int tau, gpumap_hits *out)
{
   gpumap_batch *b = nullptr;
   GM_FAIL_IF(out == nullptr, "argument 'out' is NULL.");
   memset(out, 0, sizeof *out);
   GM_FAIL_IF(ctx == nullptr, "argument 'ctx' is NULL.");
   b = gpumap_batch_upload(ctx, codes, offsets, n_queries);
   GM_FAIL_IF(b == nullptr, "cannot upload the batch.");
   GM_FAIL_IF(gm_batch_run(ctx, b, tau, 1, GM_RUN_SC) != 0, "batch failed.");
   GM_FAIL_IF(gm_batch_fetch(ctx, b, out, 0) != 0, "cannot fetch results.");
   gpumap_batch_free(b);
   return 0;
failure_return:
   gpumap_batch_free(b);
   return -1;
}

extern "C" int gpumap_map_batch(gpumap_ctx *ctx, const uint8_t *codes, const int64_t *offsets, int64_t n_reads,
                                int tau, int want_intervals, gpumap_hits *out)
{
   gpumap_batch *b = nullptr;
   GM_FAIL_IF(out == nullptr, "argument 'out' is NULL.");
   memset(out, 0, sizeof *out);
   GM_FAIL_IF(ctx == nullptr, "argument 'ctx' is NULL.");
   if (ctx->pinned_results) {
      // the arenas belong to a batch: one batch that lives as long as the context serves these calls
      if (ctx->map_batch_obj == nullptr) ctx->map_batch_obj = gpumap_batch_new(ctx);
      b = ctx->map_batch_obj;
      GM_FAIL_IF(b == nullptr, "cannot create the batch.");
      GM_FAIL_IF(gm_batch_fill(ctx, b, codes, offsets, n_reads, 0, true) != 0, "cannot upload the batch.");
      GM_FAIL_IF(gpumap_batch_run(ctx, b, tau, want_intervals) != 0, "batch failed.");
      GM_FAIL_IF(gpumap_batch_fetch(ctx, b, out) != 0, "cannot fetch results.");
      return 0;
   }
   b = gpumap_batch_upload(ctx, codes, offsets, n_reads);
   GM_FAIL_IF(b == nullptr, "cannot upload the batch.");
   GM_FAIL_IF(gpumap_batch_run(ctx, b, tau, want_intervals) != 0, "batch failed.");
   GM_FAIL_IF(gpumap_batch_fetch(ctx, b, out) != 0, "cannot fetch results.");
   gpumap_batch_free(b);
   return 0;
failure_return:
   if (ctx == nullptr || b != ctx->map_batch_obj) gpumap_batch_free(b);
   return -1;
}

// ------------------------------------------------------------------------------------------------
// ann_build (src/index_ann.c:56-257) on the device: see gm_ann.hpp.  The k-mers are searched in chunks through the
// batch machinery above (gm_batch_run with GM_RUN_SC and interval output: blocksc_trail per k-mer, trail = 0).
// ------------------------------------------------------------------------------------------------
namespace {
struct IsQuery {
   __host__ __device__ uint8_t operator()(uint8_t c) const { return (uint8_t)((c & 3) == ANN_CLS_QUERY); }
};
struct MaxOwner {
   __host__ __device__ int64_t operator()(int64_t a, int64_t b) const { return a > b ? a : b; }
};
}  // namespace

extern "C" int gpumap_ann_build(gpumap_ctx *ctx, int kmer, int tau, uint8_t *table, int64_t stats[8])
{
   gpumap_batch *b = nullptr;
   DevBuf cls, rec, d_table, q_sa, n_sel, codes, offs, ukey, ukey2, uval, uval2, tmp;
   AnnPlanHost plan;
   hipEvent_t e0 = nullptr, e1 = nullptr;
   int rc = -1;
   GM_FAIL_IF(ctx == nullptr, "argument 'ctx' is NULL.");
   GM_FAIL_IF(table == nullptr, "argument 'table' is NULL.");
   GM_FAIL_IF(kmer < 2, "argument 'kmer' must be greater than 1.");                // src/index_ann.c:76-79
   GM_FAIL_IF(tau < 1, "argument 'tau' must be greater than 0.");
   GM_FAIL_IF(tau >= 4, "argument 'tau' must be less than 4.");
   GM_FAIL_IF(tau >= kmer, "incompatible arguments (tau >= kmer).");
   GM_FAIL_IF(kmer > 250, "argument 'kmer' must be at most 250 (positions are kept in bytes).");
   GM_FAIL_IF(ctx->d_text == nullptr, "the context holds no text.");
   HIP_TRY(hipSetDevice(ctx->device));
   {
      // The left block of a k-mer (src/blocksearch.c:69-70) is searched exactly, in (tau / 2 + 1) leaf blocks at most:
      // where the context's jump table is longer than those, a second one of their length stands in for their steps
      // (k = 25, tau 1: 11 of about 40 steps per k-mer).
      const int leaf = (kmer / 2) / (tau / 2 + 1);
      const int k2 = std::min(leaf - 1, 12);
      const bool fits = k2 >= 8 && ((int64_t)1 << (2 * k2)) <= ctx->ix.n && !(getenv("GPUMAP_ANN_JUMP2") && atoi(getenv("GPUMAP_ANN_JUMP2")) == 0);
      // (jump_k = 0: the tables are switched off, or the text is too short for any)
      if (fits && ctx->jump_k > k2) GM_FAIL_IF(gm_ctx_build_jump2(ctx, k2) != 0, "cannot build the second jump table.");
   }
   {
      const int64_t n = ctx->ix.n, half = n / 2;
      const uint8_t *text = (const uint8_t *)ctx->d_text;
      hipStream_t st = ctx->stream;
      // entries per launch of the kernels that run over the suffix array (a launch of 2^32 threads is cut short) and
      // per call of the device-wide primitives (they count their items in 32 bits)
      const int64_t piece = getenv("GPUMAP_ANN_PIECE") ? std::max<int64_t>(256, atoll(getenv("GPUMAP_ANN_PIECE"))) : ((int64_t)1 << 30);
      const int64_t launch_piece = piece;
      int64_t n_ext = 0, n_hits = 0, n_updates = 0, n_extra = 0;
      // the reference cuts its jobs at prefixes of `depth` letters: the smallest depth with 5^depth >= 5 * threads (:116-121)
      int depth = 1;
      for (int64_t jobs = 5; jobs < 5 * (int64_t)ctx->ann_threads; jobs *= 5) depth++;
      const bool reuse = ctx->ann_trail_reuse;
      HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
      HIP_TRY(hipEventRecord(e0, st));
      HIP_TRY(cls.ensure((size_t)n + 8));
      HIP_TRY(rec.ensure(8 * (size_t)n));
      HIP_TRY(d_table.ensure((size_t)half + 8));
      HIP_TRY(hipMemsetAsync(cls.p, 0, (size_t)n + 8, st));
      HIP_TRY(hipMemsetAsync(rec.p, 0, 8 * (size_t)n, st));
      HIP_TRY(hipMemsetAsync(d_table.p, 0, (size_t)half + 8, st));
      for (int64_t f = 0; f < n; f += launch_piece) {
         const int64_t l = std::min(n, f + launch_piece);
         hipLaunchKernelGGL(k_ann_classify, dim3((unsigned)cdiv(l - f, 256)), dim3(256), 0, st, ctx->ix, text, kmer, tau, depth, cls.as<uint8_t>(), f, l);
      }
      HIP_TRY(hipGetLastError());
      // the k-mers to search, in suffix-array order
      HIP_TRY(q_sa.ensure(8 * (size_t)n));
      HIP_TRY(n_sel.ensure(8));
      int64_t n_q = 0;
      for (int64_t p0 = 0; p0 < n; p0 += piece) {
         const int items = (int)std::min(piece, n - p0);
         hipcub::CountingInputIterator<int64_t> idx(p0);
         hipcub::TransformInputIterator<uint8_t, IsQuery, const uint8_t *> flags(cls.as<const uint8_t>() + p0, IsQuery());
         size_t bytes = 0;
         int64_t got = 0;
         HIP_TRY(hipcub::DeviceSelect::Flagged(nullptr, bytes, idx, flags, q_sa.as<int64_t>() + n_q, n_sel.as<int64_t>(), items, st));
         HIP_TRY(tmp.ensure(bytes));
         HIP_TRY(hipcub::DeviceSelect::Flagged(tmp.p, bytes, idx, flags, q_sa.as<int64_t>() + n_q, n_sel.as<int64_t>(), items, st));
         HIP_TRY(hipMemcpyAsync(&got, n_sel.p, 8, hipMemcpyDeviceToHost, st));
         HIP_TRY(hipStreamSynchronize(st));
         n_q += got;
      }
      b = gpumap_batch_new(ctx);
      GM_FAIL_IF(b == nullptr, "cannot create the batch.");
      // (interval output: < 2^24 queries per batch; 4 M: 5.6 s, 8 M: 5.1 s, 16 M: 4.9 s at 2.4 G symbols.  With the reference's
      // reuse of searches a k-mer can become two queries: half the chunk, so that any number of them fits)
      const int64_t chunk = getenv("GPUMAP_ANN_CHUNK") ? std::max<int64_t>(1, atoll(getenv("GPUMAP_ANN_CHUNK"))) : (int64_t)(reuse ? 8000000 : 16000000);
      if (reuse) {
         AnnCarry fresh;
         for (int c = 0; c < 2; c++) { fresh.prev[c] = fresh.ol[c] = fresh.oll[c] = -1; fresh.ol_bound[c] = 0; fresh.ol_tl[c] = 0; }
         HIP_TRY(plan.carry.ensure(sizeof fresh));
         HIP_TRY(hipMemcpyAsync(plan.carry.p, &fresh, sizeof fresh, hipMemcpyHostToDevice, st));
         HIP_TRY(hipStreamSynchronize(st));
         plan.q_sa = q_sa.as<const int64_t>(); plan.kmer = kmer; plan.depth = depth;
         b->sc_plan = &plan; b->extra_factor = 2;
      }
      for (int64_t c0 = 0; c0 < n_q; c0 += chunk) {
         const int64_t nq = std::min(chunk, n_q - c0);
         plan.c0 = c0;
         HIP_TRY(codes.ensure((size_t)nq * (size_t)kmer + 4));
         HIP_TRY(offs.ensure(8 * (size_t)(nq + 1)));
         hipLaunchKernelGGL(k_ann_gather, dim3((unsigned)cdiv(std::max<int64_t>(nq + 1, cdiv(nq * kmer, 4)), 256)), dim3(256), 0, st, ctx->ix, text, q_sa.as<const int64_t>() + c0, nq,
                            kmer, codes.as<uint8_t>(), offs.as<int64_t>());
         HIP_TRY(hipGetLastError());
         HIP_TRY(hipStreamSynchronize(st));               // (the batch works on a stream of its own)
         GM_FAIL_IF(gm_batch_fill(ctx, b, codes.p, offs.as<const int64_t>(), nq, GM_SRC_DEVICE, false) != 0, "cannot stage the k-mers.");
         GM_FAIL_IF(gm_batch_run(ctx, b, tau, 1, GM_RUN_SC | GM_RUN_NO_LOCATE) != 0, "the search of the k-mers failed.");
         const int64_t H = b->n_real, U = H + nq;
         GM_FAIL_IF(U >= ((int64_t)1 << 31), "too many hits in one chunk of k-mers (lower GPUMAP_ANN_CHUNK).");
         n_ext += b->n_ext; n_hits += H; n_updates += U; n_extra += plan.n_extra;
         for (DevBuf *x : {&ukey, &ukey2, &uval, &uval2}) HIP_TRY(x->ensure(8 * (size_t)U));
         hipLaunchKernelGGL(k_ann_updates, dim3((unsigned)cdiv(nq, 256)), dim3(256), 0, st, nq, kmer, tau, b->sc_qfp.as<const int64_t>(),
                            b->sc_qrp.as<const int64_t>(), b->hit_read.as<const int64_t>(), b->hit_fp.as<const int64_t>(),
                            b->hit_rp.as<const int64_t>(), b->hit_score.as<const int64_t>(), b->hit_align.as<const uint64_t>(), H,
                            ukey.as<unsigned long long>(), uval.as<unsigned long long>(), cls.as<uint32_t>());
         HIP_TRY(hipGetLastError());
         hipcub::DoubleBuffer<unsigned long long> dk(ukey.as<unsigned long long>(), ukey2.as<unsigned long long>());
         hipcub::DoubleBuffer<unsigned long long> dv(uval.as<unsigned long long>(), uval2.as<unsigned long long>());
         size_t bytes = 0;
         const int end_bit = std::min(64, 8 + ctx->ix.sar_bits + 1);
         HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, dk, dv, (int)U, 0, end_bit, st));
         HIP_TRY(tmp.ensure(bytes));
         HIP_TRY(hipcub::DeviceRadixSort::SortPairs(tmp.p, bytes, dk, dv, (int)U, 0, end_bit, st));
         hipLaunchKernelGGL(k_ann_fold, dim3((unsigned)cdiv(U, 256)), dim3(256), 0, st, (const unsigned long long *)dk.Current(),
                            (const unsigned long long *)dv.Current(), U, rec.as<uint64_t>());
         HIP_TRY(hipGetLastError());
         HIP_TRY(hipStreamSynchronize(st));
      }
      // the compression (the list of k-mers is done with: its memory holds the owners now)
      int64_t *own = q_sa.as<int64_t>();
      for (int64_t f = 0; f < n; f += launch_piece) {
         const int64_t l = std::min(n, f + launch_piece);
         hipLaunchKernelGGL(k_ann_state, dim3((unsigned)cdiv(l - f, 256)), dim3(256), 0, st, rec.as<const uint64_t>(), cls.as<const uint8_t>(), f, l, own);
      }
      for (int64_t p0 = 0; p0 < n; p0 += piece) {
         const int items = (int)std::min(piece, n - p0);
         if (p0 > 0) hipLaunchKernelGGL(k_ann_carry, dim3(1), dim3(1), 0, st, own + p0);
         size_t bytes = 0;
         HIP_TRY(hipcub::DeviceScan::InclusiveScan(nullptr, bytes, own + p0, own + p0, MaxOwner(), items, st));
         HIP_TRY(tmp.ensure(bytes));
         HIP_TRY(hipcub::DeviceScan::InclusiveScan(tmp.p, bytes, own + p0, own + p0, MaxOwner(), items, st));
      }
      for (int64_t f = 0; f < n; f += launch_piece) {
         const int64_t l = std::min(n, f + launch_piece);
         hipLaunchKernelGGL(k_ann_push, dim3((unsigned)cdiv(l - f, 256)), dim3(256), 0, st, ctx->ix, rec.as<const uint64_t>(), (const int64_t *)own,
                            kmer, d_table.as<uint32_t>(), f, l);
      }
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipMemcpyAsync(table, d_table.p, (size_t)half, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipEventRecord(e1, st));
      HIP_TRY(hipStreamSynchronize(st));
      float ms = 0;
      HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
      if (stats) {
         stats[0] = n_q; stats[1] = n_ext; stats[2] = n_hits; stats[3] = n_updates; stats[4] = (int64_t)(1000.0f * ms);
         stats[5] = ctx->jump2_k;                      // letters of the second jump table in place (0: none)
         stats[6] = n_extra;                           // k-mers that the reuse of searches made two queries of
         stats[7] = 0;
      }
   }
   rc = 0;
failure_return:
   if (e0) (void)hipEventDestroy(e0);
   if (e1) (void)hipEventDestroy(e1);
   if (b) b->sc_plan = nullptr;
   plan.release();
   gpumap_batch_free(b);
   for (DevBuf *x : {&cls, &rec, &d_table, &q_sa, &n_sel, &codes, &offs, &ukey, &ukey2, &uval, &uval2, &tmp}) x->release();
   return rc;
}
