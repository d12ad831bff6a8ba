// gm_kernels.hpp -- HIP kernels of the query hot path (gfx950 / CDNA4, wave64).
//
//   k_super_counts, k_pack_blocks   K0  reference Occ layout -> 64-byte rank blocks (one-off)
//   k_search                        K2  block search for a batch of reads (the hot kernel)
//   k_place, k_place_long           K3  every entry of the hit list fills its slots of the located output
//   k_encode                        K4  letters -> codes, fixed-stride read matrix
//   k_occ_all, k_query_all, k_sar_get   batched forms of the reference's single-query operators
//
// Design notes (DESIGN.md has the long form): the path is integer and gather-bound, no MFMA.  What is
// tuned is the number of cache lines touched per extension step, the number of independent chains
// in flight and the number of instructions per step.  The search kernel is in gm_search.hpp.
#pragma once
#include <hip/hip_runtime.h>
#include "gm_core.hpp"
#include "gm_devctx.hpp"

namespace gm {

constexpr int WG_WAVES = 4;
constexpr int WG_THREADS = 64 * WG_WAVES;
constexpr int QCAP = 128;            // deque capacity per wave (nodes), power of two
constexpr int SPILL_CHUNK = 64;      // nodes moved per spill / reload
constexpr uint16_t NO_ROW = 0xffff;  // len_row[m]: the reference returns -1 for reads of m letters (src/blocksearch.c:126)

// ------------------------------------------------------------------------------------------------
// K0: repack
// ------------------------------------------------------------------------------------------------

// Counts of the five symbols in BWT[0, e) from the reference's array (the arithmetic of
// get_occ_all, src/index_bwt.c:863-883, always from the mark before the word).
// __noinline__: inlined into a kernel, this function makes hipcc 7.2's correlated-propagation pass
// crash; it only runs in the one-off repack, so the call costs nothing that matters.
__device__ __noinline__ void ref_rank_excl(const uint64_t *occ, int64_t intv, int64_t e, uint64_t cnt[SYMS])
{
   int64_t w = e >> 6;
   int bit = (int)(e & 63);
   int64_t mk = w / intv;
   const uint64_t *m = occ + mk * (intv + 1) * SYMS;
   for (int j = 0; j < SYMS; j++) cnt[j] = m[j];
   for (int64_t ww = mk * intv; ww < w; ww++) {
      const uint64_t *g = occ + (ww + ww / intv + 1) * SYMS;
      for (int j = 0; j < SYMS; j++) cnt[j] += __popcll(g[j]);
   }
   if (bit) {
      const uint64_t *g = occ + (w + w / intv + 1) * SYMS;
      for (int j = 0; j < SYMS; j++) cnt[j] += __popcll(g[j] >> (64 - bit));
   }
}

__global__ void k_super_counts(const uint64_t *ref_occ, int64_t intv, int64_t n, int64_t n_super,
                               const uint64_t *c, uint64_t *tables)
{
   int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
   if (s < SYMS) tables[TAB_C + s] = c[s];
   if (s >= n_super) return;
   uint64_t cnt[SYMS];
   int64_t e = s << SUPER_LOG;
   if (e > n) e = n;
   ref_rank_excl(ref_occ, intv, e, cnt);
   for (int j = 0; j < SYMS; j++) tables[TAB_SBASE + SYMS * s + j] = cnt[j];
}

struct PackArgs {
   const uint64_t *ref_occ;
   const uint64_t *tables;
   uint32_t *blocks;
   int64_t intv, n_ref_words, n_blocks;
};

__global__ void k_pack_blocks(PackArgs P)
{
   int64_t b = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
   if (b >= P.n_blocks) return;
   int64_t e = b << BLOCK_POS_LOG;
   uint64_t cnt[SYMS];
   ref_rank_excl(P.ref_occ, P.intv, e, cnt);
   uint32_t rel[SYMS];
   const uint64_t *base = P.tables + TAB_SBASE + SYMS * (e >> SUPER_LOG);
   for (int j = 0; j < SYMS; j++) rel[j] = (uint32_t)(cnt[j] - base[j]);
   uint32_t out[BLOCK_DWORDS];
   pack_block(P.ref_occ, P.intv, P.n_ref_words, b, rel, out);
   uint4 *dst = reinterpret_cast<uint4 *>(P.blocks) + 4 * b;
   for (int k = 0; k < 4; k++) dst[k] = make_uint4(out[4 * k], out[4 * k + 1], out[4 * k + 2], out[4 * k + 3]);
}

// ------------------------------------------------------------------------------------------------
// Single-thread rank / extension (operator kernels and locate; not the hot path)
// ------------------------------------------------------------------------------------------------

__device__ inline void rank_abs(const IndexDev &ix, int64_t e, int64_t occ[SYMS])
{
   uint32_t rel[SYMS];
   block_rank(ix.occ + (e >> BLOCK_POS_LOG) * BLOCK_DWORDS, (int)(e & (BLOCK_POS - 1)), rel);
   const uint64_t *base = ix.tables + TAB_SBASE + SYMS * (e >> SUPER_LOG);
   for (int j = 0; j < SYMS; j++) occ[j] = (int64_t)(base[j] + rel[j]);
}

// get_occ_all for n positions (src/index_bwt.c:840-905): ptr = -1 gives zeros.
__global__ void k_occ_all(IndexDev ix, int64_t n, const int64_t *ptr, int64_t *out)
{
   int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
   if (i >= n) return;
   int64_t p = ptr[i];
   int64_t occ[SYMS] = {-1, -1, -1, -1, -1};
   if (p >= -1 && p < ix.n) rank_abs(ix, p + 1, occ);
   for (int j = 0; j < SYMS; j++) out[SYMS * i + j] = occ[j];
}

// bwt_query_all for n intervals (src/index_bwt.c:198-304).
__global__ void k_query_all(IndexDev ix, int end, int64_t n, const int64_t *fp, const int64_t *rp,
                            const int64_t *sz, int64_t *out)
{
   int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
   if (i >= n) return;
   int64_t a = end ? rp[i] : fp[i], other = end ? fp[i] : rp[i], s = sz[i];
   int64_t *o = out + 15 * i;
   if (a < 0 || s < 0 || a + s > ix.n) { for (int k = 0; k < 15; k++) o[k] = -1; return; }
   int64_t occ0[SYMS], occ1[SYMS], fpn[SYMS], rpn[SYMS], szn[SYMS];
   uint64_t c[SYMS];
   for (int j = 0; j < SYMS; j++) c[j] = ix.tables[TAB_C + j];
   rank_abs(ix, a, occ0);
   rank_abs(ix, a + s, occ1);
   extend_all(other, s, occ0, occ1, c, fpn, rpn, szn);
   for (int j = 0; j < SYMS; j++) {
      int js = end ? complement(j) : j;                      // :280-286
      o[3 * j + 0] = end ? rpn[js] : fpn[js];
      o[3 * j + 1] = end ? fpn[js] : rpn[js];
      o[3 * j + 2] = szn[js];
   }
}

// bwt_prefix_all for n intervals (src/index_bwt.c:390-446): the forward pointer and the size of the five one-letter
// prefix extensions, out[i][j] = {c[j] + occ(fp - 1)[j], max(0, occ(fp + sz - 1)[j] - occ(fp - 1)[j])}; no reverse
// pointer (the reference sets rp = -1).  A rank the reference's get_occ_all refuses (pointer below -1 or beyond the
// last position) makes the ten words of that interval -1.
__global__ void k_prefix_all(IndexDev ix, int64_t n, const int64_t *fp, const int64_t *sz, int64_t *out)
{
   int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
   if (i >= n) return;
   int64_t a = fp[i], b = a + sz[i];
   int64_t *o = out + 2 * SYMS * i;
   if (a < 0 || a > ix.n || b < 0 || b > ix.n) { for (int k = 0; k < 2 * SYMS; k++) o[k] = -1; return; }
   int64_t occ0[SYMS], occ1[SYMS];
   rank_abs(ix, a, occ0);
   rank_abs(ix, b, occ1);
   for (int j = 0; j < SYMS; j++) {
      int64_t s = occ1[j] - occ0[j];
      o[2 * j + 0] = (int64_t)ix.tables[TAB_C + j] + occ0[j];
      o[2 * j + 1] = s < 0 ? 0 : s;
   }
}

// Jump table: entry idx = the state of the exact right-to-left search of one K-mer from the root
// interval, i.e. what the first K iterations of a leaf phase (seqsearch_bw with tau 0,
// src/blocksearch.c:138-144, :440-496) compute for the query symbols s_0 (first, rightmost) .. s_{K-1},
// idx = sum s_t << 2t.  16 bytes in the layout of a node's first four words: fp, rp, sz (40 bits each)
// and d in the top byte: d = K, or, when sz = 0, the step at which the interval became empty (the
// reference stops there, after d calls of bwt_query_all).  One thread per K-mer.
__global__ void k_build_jump(IndexDev ix, int K, int64_t first, uint4 *table)
{
   const int64_t idx = first + blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
   if (idx >= ((int64_t)1 << (2 * K))) return;
   uint64_t c[SYMS];
   for (int j = 0; j < SYMS; j++) c[j] = ix.tables[TAB_C + j];
   int64_t fp = 0, rp = 0, sz = ix.n;
   int d = K;
   for (int t = 0; t < K; t++) {
      const int s = (int)((idx >> (2 * t)) & 3);
      int64_t occ0[SYMS], occ1[SYMS], fpn[SYMS], rpn[SYMS], szn[SYMS];
      rank_abs(ix, fp, occ0);
      rank_abs(ix, fp + sz, occ1);
      extend_all(rp, sz, occ0, occ1, c, fpn, rpn, szn);
      if (szn[s] < 1) { sz = 0; d = t + 1; break; }
      fp = fpn[s]; rp = rpn[s]; sz = szn[s];
   }
   table[idx] = make_uint4((uint32_t)fp, (uint32_t)rp, (uint32_t)sz,
                           (uint32_t)((fp >> 32) & 0xff) | ((uint32_t)((rp >> 32) & 0xff) << 8) |
                           ((uint32_t)((sz >> 32) & 0xff) << 16) | ((uint32_t)d << 24));
}

// sar_get (src/index_sar.c:92-118): entry i = bits [i*b, (i+1)*b) of the little-endian stream.
__device__ inline int64_t sa_entry(const uint64_t *sa, int bits, int64_t i)
{
   uint64_t bit = (uint64_t)i * (uint64_t)bits;
   uint64_t w = bit >> 6;
   int sh = (int)(bit & 63);
   uint64_t v = sa[w] >> sh;
   if (sh + bits > 64) v |= sa[w + 1] << (64 - sh);
   return (int64_t)(v & (~(uint64_t)0 >> (64 - bits)));
}

__global__ void k_sar_get(IndexDev ix, int64_t n, const int64_t *pos, int64_t *out)
{
   int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
   if (i >= n) return;
   int64_t p = pos[i];
   out[i] = (p < 0 || p >= ix.n) ? -1 : sa_entry(ix.sa, ix.sar_bits, p);
}

// ------------------------------------------------------------------------------------------------
// K4: reads.  Letters (or codes) -> codes in a fixed-stride matrix [n_reads][stride].
// ------------------------------------------------------------------------------------------------
// Lengths of the reads of a batch from its offsets (device memory): read_len[r], and which lengths occur
// (seen[len] = 1 for len <= MAX_READ, seen[MAX_READ + 1] = 1 for longer reads, seen[MAX_READ + 2] = 1 when an
// offset decreases).  The host builds one phase-table row per length from `seen` and sizes the read matrix by
// the longest; no loop over the reads on the host.  One store per workgroup and length (LDS flags first).
constexpr int SEEN_LONGER = MAX_READ + 1, SEEN_ERROR = MAX_READ + 2, SEEN_BYTES = 264;
__global__ void __launch_bounds__(256) k_read_lengths(const int64_t *offsets, int64_t n_reads, uint16_t *read_len, uint8_t *seen)
{
   __shared__ uint32_t s_seen[SEEN_BYTES];
   for (int i = threadIdx.x; i < SEEN_BYTES; i += blockDim.x) s_seen[i] = 0;
   __syncthreads();
   for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < n_reads; r += (int64_t)gridDim.x * blockDim.x) {
      const int64_t len = offsets[r + 1] - offsets[r];
      read_len[r] = (uint16_t)(len < 0 ? 0 : (len > 0xffff ? 0xffff : len));
      s_seen[len < 0 ? SEEN_ERROR : (len > MAX_READ ? SEEN_LONGER : (int)len)] = 1;
   }
   __syncthreads();
   for (int i = threadIdx.x; i < SEEN_BYTES; i += blockDim.x)
      if (s_seen[i]) seen[i] = 1;
}

// K4: letters (or codes) of a batch -> the fixed-stride code matrix [n_reads][stride] (what k_search indexes by
// position; 0xff behind the end of a read) AND the same rows as bit planes (what k_verify compares; see
// k_pack_planes), in one pass: one thread per 32 symbols of a row.  src holds the letters of the batch: at
// offsets[r] (relative = 0: the caller's device buffer), or at offsets[r] - offsets[0] (relative = 1: the
// library's copy of the part of a host buffer the batch uses).  The letters are fetched as the aligned 32-bit
// words that hold them (a word that holds a letter of the buffer lies inside the buffer's pages) and brought
// into place with v_alignbit; `table` (sym_table of the index, 256 entries) sits in LDS.  table = nullptr: src
// holds codes already (values above 4 become N, as sym_index does for characters it does not know).
// ends (or nullptr): the letters of read r are src[offsets[r], ends[r]) (reads parsed out of a block of a file, gm_parse.hpp).
__global__ void __launch_bounds__(256) k_encode(const uint8_t *src, const int64_t *offsets, const int64_t *ends, int relative, int64_t n_reads, int stride,
                                                const uint8_t *table, uint8_t *dst, uint32_t *planes)
{
   __shared__ uint8_t s_tab[256];
   if (table) s_tab[threadIdx.x] = table[threadIdx.x];
   __syncthreads();
   const int64_t sub = relative ? offsets[0] : 0;
   const int gpr = stride >> 5;                                   // groups of 32 symbols per row
   // grid-stride: a launch holds fewer than 2^32 threads
   for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n_reads * gpr; t += (int64_t)gridDim.x * blockDim.x) {
      // (stride is a multiple of 128: gpr is 4, 8, 12, ...; a shift where it is a power of two, a 32-bit division
      // where the batch allows it: the 64-bit division was a third of the kernel's instructions)
      const int64_t r = (gpr & (gpr - 1)) == 0 ? t >> (31 - __clz(gpr)) : (t < ((int64_t)1 << 32) ? (int64_t)((uint32_t)t / (uint32_t)gpr) : t / gpr);
      const int g = (int)(t - r * gpr);
      const int64_t beg = offsets[r], len = (ends ? ends[r] : offsets[r + 1]) - beg;
      int64_t k64 = len - 32 * (int64_t)g;
      const int k = k64 < 0 ? 0 : (k64 > 32 ? 32 : (int)k64);    // letters of the read in this group
      uint32_t c[8];
#pragma unroll
      for (int j = 0; j < 8; j++) c[j] = 0xffffffffu;
      if (k > 0) {
         const uint8_t *addr = src + (beg - sub) + 32 * g;
         const uint32_t *base = reinterpret_cast<const uint32_t *>(reinterpret_cast<uintptr_t>(addr) & ~(uintptr_t)3);
         const uint32_t sh = (uint32_t)(reinterpret_cast<uintptr_t>(addr) & 3) * 8;
         const int nd = ((int)(sh >> 3) + k + 3) >> 2;           // aligned words that hold the k letters (at most 9)
         uint32_t w[9];
#pragma unroll
         for (int j = 0; j < 9; j++) w[j] = j < nd ? base[j] : 0u;
#pragma unroll
         for (int j = 0; j < 8; j++) {
            const uint32_t x = __builtin_amdgcn_alignbit(w[j + 1], w[j], sh);
            uint32_t y = 0;
#pragma unroll
            for (int q = 0; q < 4; q++) {
               const uint32_t ch = (x >> (8 * q)) & 0xff;
               const uint32_t code = table ? (uint32_t)s_tab[ch] : (ch < (uint32_t)SYMS ? ch : (uint32_t)UNKNOWN);
               y |= code << (8 * q);
            }
            // 0xff behind the read's last letter: per word (bytes k - 4j .. 3 of it), not per letter
            const int left = k - 4 * j;
            c[j] = left >= 4 ? y : (left <= 0 ? 0xffffffffu : (y | (0xffffffffu << (8 * left))));
         }
      }
      if (dst != nullptr) {                 // (nullptr: nobody reads the byte matrix in this run: the mapping search and k_verify read the planes)
         uint4 *d = reinterpret_cast<uint4 *>(dst + r * stride + 32 * g);
         d[0] = make_uint4(c[0], c[1], c[2], c[3]);
         d[1] = make_uint4(c[4], c[5], c[6], c[7]);
      }
      uint32_t pl[3] = {0, 0, 0};
#pragma unroll
      for (int j = 0; j < 8; j++)
#pragma unroll
         for (int p = 0; p < 3; p++)       // bits p of four bytes -> four adjacent bits (the products' terms do not meet)
            pl[p] |= (((((c[j] >> p) & 0x01010101u) * 0x01020408u) >> 24) & 0xfu) << (4 * j);
      uint32_t *o = planes + 3 * t;
      o[0] = pl[0]; o[1] = pl[1]; o[2] = pl[2];
   }
}

// Bit planes for the windows k_verify compares: 32 symbols are 12 bytes, word p = bit p of the symbols'
// codes (symbol k in bit k), so that the places where a window and a query differ come out of XORs of
// whole words, in the order of the letters, and a window is one run of 12 bytes per 32 letters.  src
// holds n_src bytes (codes 0..5, or 0xff behind the end of a read); what lies behind is written as
// `fill`.  One thread per 32 symbols.
__global__ void k_pack_planes(const uint8_t *src, int64_t n_src, uint8_t fill, uint32_t *dst, int64_t n_groups)
{
   for (int64_t g = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; g < n_groups; g += (int64_t)gridDim.x * blockDim.x) {
      uint32_t x[8];
      if (32 * g + 32 <= n_src) {
         const uint4 v0 = *reinterpret_cast<const uint4 *>(src + 32 * g), v1 = *reinterpret_cast<const uint4 *>(src + 32 * g + 16);
         x[0] = v0.x; x[1] = v0.y; x[2] = v0.z; x[3] = v0.w; x[4] = v1.x; x[5] = v1.y; x[6] = v1.z; x[7] = v1.w;
      } else {
#pragma unroll
         for (int j = 0; j < 8; j++) {
            x[j] = 0;
            for (int k = 0; k < 4; k++) {
               const int64_t at = 32 * g + 4 * j + k;
               x[j] |= (uint32_t)(at < n_src ? src[at] : fill) << (8 * k);
            }
         }
      }
      uint32_t pl[3] = {0, 0, 0};
#pragma unroll
      for (int j = 0; j < 8; j++)
#pragma unroll
         for (int p = 0; p < 3; p++)       // bits p of four bytes -> four adjacent bits (the products' terms do not meet)
            pl[p] |= (((((x[j] >> p) & 0x01010101u) * 0x01020408u) >> 24) & 0xfu) << (4 * j);
      uint32_t *o = dst + 3 * g;
      o[0] = pl[0]; o[1] = pl[1]; o[2] = pl[2];
   }
}

// Flat variant for gpumap_encode: sym_str_index (src/index_sym.c:293-319) over a whole buffer.
__global__ void k_encode_flat(const uint8_t *src, int64_t n, const uint8_t *table, uint8_t *dst)
{
   for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
      dst[i] = table[src[i]];
}

// ------------------------------------------------------------------------------------------------
// K2: the search
// ------------------------------------------------------------------------------------------------
struct SearchArgs {
   IndexDev ix;
   const uint8_t  *reads;       // [n_reads][stride] codes (the blocksc_trail search; the mapping search reads `planes`)
   const uint32_t *planes;      // [n_reads][3 * stride / 32] the rows as bit planes (k_encode)
   const uint16_t *read_len;    // [n_reads]
   const uint16_t *len_row;     // [MAX_READ + 1] -> row of `phases`, NO_ROW = the reference returns -1
   const uint32_t *phases;      // [rows][MAX_PHASES]
   const uint2    *phase_rec;   // [rows][MAX_PHASES] what a node becomes at the end of a phase (phase_record)
   int n_rows;
   int64_t n_reads;
   int stride, tau;
   unsigned long long *cursor;  // next read to admit
   Node *hits;                  // finished nodes
   unsigned long long hit_cap;
   unsigned long long *hit_count;
   unsigned long long *n_ext;   // extension steps executed
   Node *spill;                 // [waves][spill_cap]
   uint32_t spill_cap;
   int *error;                  // set to 1 when a spill region overflows
   unsigned long long *stats;   // [0] passes of the main loop over all waves, [1] continuations pushed,
                                // [2] extension steps the jump table stood in for, [3] entries of the list that are not holes,
                                // [4] chunks of SPILL_CHUNK nodes that deques moved to their spill regions
   const uint4 *jump;           // jump table (k_build_jump) or nullptr
   int jump_k;
   const uint4 *jump2;          // a shorter table for the leaves `jump` is too long for, or nullptr
   int jump2_k;
   int verify;                  // nodes with sz <= verify (0, 1, 2 or 4) leave as candidates for k_verify
   int verify_stable;           // 1: those of more than two suffixes only when the last step left the interval as it was
   int seed_cand;               // 1: a seed whose entry of the jump table holds one suffix leaves as a candidate at once
   uint32_t cursor_chunk;       // reads a wave reserves per atomic on the cursor (a multiple of its refill)
   // k_search<true> (blocksc_trail): per query its row of the phase tables (NO_ROW: not searched) and the bound of the
   // left block's hits (src/blocksearch.c:76), per row its number of leaf phases
   const uint16_t *read_row;
   const int64_t  *sc_bound;
   const uint8_t  *row_leaves;
   const uint8_t  *sc_flags;    // nullptr, or per query: bits 0..1 the leaves it starts from, bits 2.. the first phase whose steps count (k_ann_plan_apply)
};

// blocksc_trail's set-up for every query of a batch (src/blocksearch.c:46-70 and what its callers compute before it,
// src/index_ann.c:783-813, test/test_blocksearch.c:49-53): the path array by one bwt_query SUFFIX per letter from the
// root interval (N as the literal symbol it is there), of which the search needs the interval of the whole query (its
// strand class `last_fragment`, :55) and the end of the left block's interval (the bound of :76); the N count that
// lowers tau (:59-63).  Row of the phase tables = ((row of the length) * (tau + 1) + tau_eff) * 2 + last; the host has
// built every row and knows which are not searched (row_state: 0 search, 1 no hits, 2 error).
__global__ void __launch_bounds__(256) k_sc_prepare(IndexDev ix, const uint8_t *reads, const uint16_t *read_len, const uint16_t *len_row,
                                                    const uint8_t *row_state, int64_t n_reads, int stride, int tau,
                                                    uint16_t *read_row, int64_t *sc_bound, uint8_t *score_add, int64_t *status,
                                                    int64_t *q_fp, int64_t *q_rp /* the interval of the whole query (path[kmer] of ann_build_mt) */,
                                                    const uint4 *jump, int jump_k, const uint4 *jump2, int jump2_k)
{
   const int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
   if (r >= n_reads) return;
   const int m = read_len[r];
   read_row[r] = NO_ROW; sc_bound[r] = 0; score_add[r] = 0; q_fp[r] = 0; q_rp[r] = 0;
   if (m < 1 || m > MAX_READ) { status[r] = -1; return; }             // :37; longer than the mismatch bitmap (src/blocksearch.h:12-13)
   uint64_t c[SYMS];
   for (int j = 0; j < SYMS; j++) c[j] = ix.tables[TAB_C + j];
   const uint8_t *qrow = reads + r * stride;
   // the letters of a query of up to 32 come as two 16-byte loads, kept as nibbles (a byte load per letter and lane, each
   // lane on a row of its own, is a trip through the address unit per letter)
   uint64_t nib0 = 0, nib1 = 0;
   if (m <= 32) {
      const uint4 *q16 = reinterpret_cast<const uint4 *>(qrow);
      const uint4 a4 = q16[0], b4 = q16[1];
      auto squeeze = [](uint32_t w) { uint32_t x = w & 0x0f0f0f0fu; x = (x | (x >> 4)) & 0x00ff00ffu; return (uint64_t)((x | (x >> 8)) & 0xffffu); };
      nib0 = squeeze(a4.x) | (squeeze(a4.y) << 16) | (squeeze(a4.z) << 32) | (squeeze(a4.w) << 48);
      nib1 = squeeze(b4.x) | (squeeze(b4.y) << 16) | (squeeze(b4.z) << 32) | (squeeze(b4.w) << 48);
   }
   auto q = [&](int i) -> int { return m <= 32 ? (int)(((i < 16 ? nib0 : nib1) >> (4 * (i & 15))) & 15u) : (int)qrow[i]; };
   int64_t fp = 0, rp = 0, sz = ix.n;                                  // bwt_new_query
   int64_t end_lo = 0, end_hi = 0;                                     // fp + sz after m/2 and after m/2 + m%2 letters
   int n_cnt = 0;
   const int cut_lo = m / 2, cut_hi = m / 2 + m % 2;
   // The bi-interval of a prefix q[0, K) without N that occurs in the text is what the jump tables hold for that string
   // (k_build_jump reaches it letter by letter from the other end): one look-up stands in for K steps of the path,
   // once below the first cut and once behind the second.
   auto look_up = [&](int at_most, int more_than) {
      const uint4 *jt = nullptr;
      int K = 0;
      if (jump != nullptr && jump_k <= at_most && jump_k > more_than) { jt = jump; K = jump_k; }
      if (jump2 != nullptr && jump2_k <= at_most && jump2_k > more_than && jump2_k > K) { jt = jump2; K = jump2_k; }
      if (jt == nullptr) return 0;
      uint32_t idx = 0, bad = 0;
      for (int t = 0; t < K; t++) {
         const uint32_t sym = (uint32_t)q(K - 1 - t);
         bad |= sym & ~3u;
         idx |= (sym & 3u) << (2 * t);
      }
      if (bad) return 0;
      const uint4 e = jt[idx];
      const int64_t esz = (int64_t)e.z | ((int64_t)((e.w >> 16) & 0xff) << 32);
      if (esz < 1) return 0;                         // (an absent prefix: the path's own pointers are wanted)
      fp = (int64_t)e.x | ((int64_t)(e.w & 0xff) << 32);
      rp = (int64_t)e.y | ((int64_t)((e.w >> 8) & 0xff) << 32);
      sz = esz;
      return K;
   };
   int i = look_up(cut_lo, 0);
   if (i > 0 && i == cut_lo) end_lo = fp + sz;
   if (i > 0 && i == cut_hi) end_hi = fp + sz;
   bool second = false;
   while (i < m) {
      if (!second && i >= cut_hi) {
         second = true;
         const int K = look_up(m, i);
         if (K > 0) { i = K; continue; }
      }
      const int s = q(i);
      n_cnt += s == UNKNOWN;
      int64_t occ0[SYMS], occ1[SYMS], fpn[SYMS], rpn[SYMS], szn[SYMS];
      // bwt_query(s, BWT_QUERY_SUFFIX): the backward step on the reverse complement's side (src/index_bwt.c:234-238, :280-286)
      rank_abs(ix, rp, occ0);
      rank_abs(ix, rp + sz, occ1);
      extend_all(fp, sz, occ0, occ1, c, fpn, rpn, szn);
      const int js = complement(s);
      fp = rpn[js]; rp = fpn[js]; sz = szn[js];
      i++;
      if (i == cut_lo) end_lo = fp + sz;
      if (i == cut_hi) end_hi = fp + sz;
   }
   const int last = fp >= rp;                                           // :55
   q_fp[r] = fp; q_rp[r] = rp;
   if (n_cnt > tau) { status[r] = 0; return; }                          // :63-66: tau < 0
   const int tau_eff = tau - n_cnt;
   const int row = (len_row[m] * (tau + 1) + tau_eff) * 2 + last;
   const int state = row_state[row];
   status[r] = state == 2 ? -1 : 0;
   if (state != 0) return;
   read_row[r] = (uint16_t)row;
   sc_bound[r] = last ? end_hi : end_lo;                                // qarray[pos_r], pos_r = m/2 + (last ? m%2 : 0) (:69, :76)
   score_add[r] = (uint8_t)n_cnt;                                       // :94-97
}

// ------------------------------------------------------------------------------------------------
// k_verify: the rest of the search of a node whose interval holds ONE suffix.  From there on the
// reference's recursion (seqsearch_* / seqdash_*, src/blocksearch.c:184-496) is a walk along one
// place of the text: at every position exactly one child interval is non-empty (the text symbol; none
// at a separator), so the walk is replayed on the text itself -- same rules, same hits, and the same
// count of bwt_query_all calls (one per position visited, the one at which the path dies included).
// One suffix-array entry and a window of the text replace one Occ line per letter.
// One thread per entry of the output list of k_search.  A candidate that becomes a hit is finished here:
// its sort key (one suffix: its text position is known) goes to the block's slot of the staging area, and the
// locate stage only has to make the slots dense; the entries that are hits already (intervals) are listed for it.
// ------------------------------------------------------------------------------------------------
struct VerifyArgs {
   IndexDev ix;
   const uint8_t  *text;        // [n] codes 0..5
   const uint8_t  *text4;       // the same as bit planes (k_pack_planes), 512 symbols of separators behind
   const uint8_t  *reads;
   const uint8_t  *reads4;      // [n_reads][3 * stride / 32] words: the rows of `reads` as bit planes (stride % 128 == 0)
   const uint16_t *row_len;     // [rows] the read length of each row of the phase tables (one row per length)
   const uint2    *phase_rec;   // [rows][MAX_PHASES]
   const uint8_t  *phase_lo;    // [rows][MAX_PHASES]
   const Node *list;
   int64_t n_list;
   int stride;
   unsigned long long *n_ext;      // [blocks of the launch] steps per block (atomics on a few hot words cost a fifth of the kernel)
   int key_shift;                  // key = read << key_shift | text position << 4 | score (as k_place)
   int tile;                       // k_verify: entries of the list a block takes (<= VERIFY_TILE_MAX)
   int stage_per;                  // k_verify: keys of a block's slot of the staging area (2 * tile with pairs, else tile)
   // Per block of the launch, no global atomics: the sort keys of the hits its candidates became (a slot of stage_per
   // keys per block (k_verify_group: STAGE_PER_BLOCK), filled from its start: k_stage_compact makes them dense), the
   // entries that are hits of k_search itself (intervals: the locate stage reads their suffix-array entries) and the
   // entries that hold more than two suffixes (k_verify_group), as lists of entry numbers in slots of `tile`.
   unsigned long long *stage;
   uint32_t *stage_cnt;
   uint32_t *iv_idx, *iv_cnt;
   uint32_t *big_idx, *big_cnt;    // candidates of 3 .. 4 suffixes (nullptr: there are none)
   uint32_t *wide_idx, *wide_cnt;  // candidates of 5 .. 16 suffixes (nullptr: there are none); with split3 those of three
   int split3;                     // 1 (option verify = 4): candidates of three suffixes go to the second list (three lanes each), those of four to the first
};
constexpr int STAGE_PER_BLOCK = 256;     // keys a block of k_verify_group can produce (one per lane)
constexpr int VERIFY_TILE_MAX = 1024;    // entries a block of k_verify takes at most

// The query row and the text window of a candidate are read as bit planes (k_pack_planes: 16 bytes per
// plane and 128 symbols; every lane of a load touches a line of its own, so the number of load
// instructions and of lines is what the memory path pays for), the window's words are shifted to the
// query's with v_alignbit and the planes XORed; what is kept is one bit per letter: E = the letter costs a
// mismatch (text and query differ and the query letter is not N), S = the text holds a separator
// there.  The walk then goes from event to event with ffs / clz on those bits: plain matches between
// events are counted, not visited.
// NCH = 64-letter pieces of the longest read of the batch.
struct __attribute__((packed, aligned(4))) u4x { uint32_t x, y, z, w; };      // 16 bytes at any word

__device__ __forceinline__ uint32_t sel4w(uint32_t a, uint32_t b, uint32_t c, uint32_t d, uint32_t j)
{
   uint32_t r = a;
   r = j == 1 ? b : r;
   r = j == 2 ? c : r;
   r = j == 3 ? d : r;
   return r;
}

// First event of mask E (NCH 64-letter pieces) in the letters a..b, seen from a (up) or from b (down); -1: none.
template <int NCH>
__device__ __forceinline__ int first_event(const uint64_t (&E)[NCH], int a, int b, bool up)
{
   int first = -1, last = -1;
#pragma unroll
   for (int c = 0; c < NCH; c++) {
      const int l = max(a - 64 * c, 0), h = min(b - 64 * c, 63);
      if (l <= h) {
         const uint64_t mk = E[c] & (~0ull << l) & (~0ull >> (63 - h));
         if (mk) {
            if (first < 0) first = 64 * c + __ffsll((unsigned long long)mk) - 1;
            last = 64 * c + 63 - __clzll((long long)mk);
         }
      }
   }
   return up ? first : last;
}

template <int NCH>
__device__ __forceinline__ bool test_bit(const uint64_t (&E)[NCH], int pos)
{
   bool r = false;
#pragma unroll
   for (int c = 0; c < NCH; c++) if ((pos >> 6) == c) r = (E[c] >> (pos & 63)) & 1;
   return r;
}

// The walk of ONE suffix of a candidate `nd` (suffix q of its interval) along its place of the text: the masks of
// the window (E, S, QN; with KEEP also the window's planes, for the comparison with the other suffixes of the entry),
// the steps of the reference schedule the walk stands for, and the sort key of the hit it ends in (~0: none).
template <int NCH, bool KEEP>
__device__ __forceinline__ void verify_replay(const VerifyArgs &A, const Node &nd, int q, int64_t sa_val /* entry fp + q of the suffix array */,
                                              uint64_t (&E)[NCH], uint64_t (&S)[NCH],
                                              uint64_t (&QN)[NCH], uint32_t (&tw)[3][2 * NCH], bool &planes, uint32_t &steps,
                                              unsigned long long &key, int64_t &p0, uint32_t &read, int &row, int &phase0,
                                              int &pos0, int &end0, bool &fw0)
{
   int pos = (int)(nd.w[3] >> 24);
   uint32_t meta = nd.w[5] & ~(1u << META_CAND);
   read = nd.w[4];
   row = (int)(nd.w[7] >> 24);
   const int m = A.row_len[row];
   int phase = (int)(meta & 63);
   bool fw = (meta >> META_FW) & 1, dash = (meta >> META_DASH) & 1;
   int end = (meta >> META_END) & 255, tau = (meta >> META_TAU) & 7, diff = (meta >> META_DIFF) & 7;
   int score = (meta >> META_SCORE) & 7, sref = (meta >> META_SREF) & 7;
   phase0 = phase; pos0 = pos; end0 = end; fw0 = fw;
   const int lo = fw ? A.phase_lo[row * MAX_PHASES + phase] : pos + 1;      // lowest query position matched so far
   p0 = sa_val - lo;                                                        // text position of query position 0
   const int nq = (m + 31) >> 5;                                            // 32-letter words of the query row
   // the window inside the text (the planes have 512 symbols of padding: whole loads can be made)
   bool fast = p0 >= 0 && p0 + m <= A.ix.n;
   if (fast) {
      constexpr int NW = 2 * NCH;                 // 32-letter words of the masks
      constexpr int NT = (3 * (NW + 1) + 3) / 4;  // 16-byte loads of the window: NW + 1 runs of 12 bytes
      constexpr int NQ = (3 * NW + 3) / 4;        // 16-byte loads of the query row
      const uint4 *q16 = reinterpret_cast<const uint4 *>(A.reads4 + (size_t)read * (size_t)(3 * (A.stride >> 3)));
      const u4x *t16 = reinterpret_cast<const u4x *>(A.text4 + 12 * (p0 >> 5));      // (aligned to 4 bytes)
      const uint32_t bs = (uint32_t)p0 & 31;      // bit offset of the window in its first words
      uint32_t T[4 * NT], Q[4 * NQ];
#pragma unroll
      for (int k = 0; k < NT; k++) {
         const u4x v = t16[k];
         T[4 * k] = v.x; T[4 * k + 1] = v.y; T[4 * k + 2] = v.z; T[4 * k + 3] = v.w;
      }
#pragma unroll
      for (int k = 0; k < NQ; k++) {
         const uint4 v = q16[k];
         Q[4 * k] = v.x; Q[4 * k + 1] = v.y; Q[4 * k + 2] = v.z; Q[4 * k + 3] = v.w;
      }
      uint32_t qw[3][NW];
#pragma unroll
      for (int p = 0; p < 3; p++)
#pragma unroll
         for (int w = 0; w < NW; w++) {
            tw[p][w] = __builtin_amdgcn_alignbit(T[3 * (w + 1) + p], T[3 * w + p], bs);
            qw[p][w] = Q[3 * w + p];
         }
      uint32_t sepacc = 0, Ew[NW];
#pragma unroll
      for (int w = 0; w < NW; w++) {
         // plane 2 is "not a base": a text N against a base costs one, a query N (and what lies behind the read) is free
         Ew[w] = ((tw[0][w] ^ qw[0][w]) | (tw[1][w] ^ qw[1][w]) | (tw[2][w] ^ qw[2][w])) & ~qw[2][w];
         if (w < nq) sepacc |= tw[2][w] & tw[0][w];                                 // code 5: a separator
      }
#pragma unroll
      for (int c = 0; c < NCH; c++) {
         E[c] = (uint64_t)Ew[2 * c] | ((uint64_t)Ew[2 * c + 1] << 32);
         QN[c] = (uint64_t)qw[2][2 * c] | ((uint64_t)qw[2][2 * c + 1] << 32);
      }
      if (sepacc) fast = false;                          // a separator in (or just behind) the window
      planes = KEEP && fast;
   }
   if (!fast) {
      // a window over an end of the text or over a separator: letter by letter, separators outside the text
#pragma unroll
      for (int c = 0; c < NCH; c++) { E[c] = 0; S[c] = 0; QN[c] = 0; }
      if (KEEP) {
#pragma unroll
         for (int p = 0; p < 3; p++)
#pragma unroll
            for (int w = 0; w < 2 * NCH; w++) tw[p][w] = 0;
      }
      for (int k = 0; k < m; k++) {
         const int64_t tp = p0 + k;
         const int t = (tp < 0 || tp >= A.ix.n) ? SYMS : A.text[tp];
         if (KEEP) {                        // the window's planes, as the fast path keeps them
#pragma unroll
            for (int w = 0; w < 2 * NCH; w++)
               if ((k >> 5) == w) {
#pragma unroll
                  for (int p = 0; p < 3; p++) tw[p][w] |= (uint32_t)((t >> p) & 1) << (k & 31);
               }
         }
         const uint32_t *qpl = reinterpret_cast<const uint32_t *>(A.reads4) + (size_t)read * (size_t)(3 * (A.stride >> 5)) + 3 * (k >> 5);
         const int qc = (int)(((qpl[0] >> (k & 31)) & 1u) | (((qpl[1] >> (k & 31)) & 1u) << 1) | (((qpl[2] >> (k & 31)) & 1u) << 2));
         const uint64_t bit = 1ull << (k & 63);
#pragma unroll
         for (int c = 0; c < NCH; c++) {
            if ((k >> 6) == c) {
               if (t == SYMS) S[c] |= bit;
               if (t == SYMS || (qc != UNKNOWN && t != qc)) E[c] |= bit;     // text N against a base costs one; query N is free
               if (qc == UNKNOWN) QN[c] |= bit;
            }
         }
      }
      planes = KEEP;
   }
   bool alive = true, final = false;
   while (alive && !final) {
      // one phase: positions pos .. end in the phase's direction
      for (;;) {
         const int a = fw ? pos : end, b = fw ? end : pos;
         const int e = first_event<NCH>(E, a, b, fw);
         if (e < 0) {                                 // plain matches up to the end of the phase
            steps += (uint32_t)(b - a + 1);
            pos = end;
            if (!dash && score - sref < diff) alive = false;          // the rule of the phase's last position
            break;
         }
         // plain matches up to the first letter where something happens, then that letter by the rules
         steps += (uint32_t)(fw ? e - pos : pos - e) + 1;
         pos = e;
         if (test_bit<NCH>(S, e)) { alive = false; break; }           // separator: no child at all
         if (dash) { alive = false; break; }
         const int s = score + 1;
         if (s > tau) { alive = false; break; }
         if (s == tau) { if (s - sref < diff) { alive = false; break; } dash = true; }
         else if (pos == end && s - sref < diff) { alive = false; break; }
         score = s;
         if (pos == end) break;
         pos += fw ? 1 : -1;
      }
      if (!alive) break;
      const uint2 rec = A.phase_rec[row * MAX_PHASES + phase];
      if (rec.y >> 8) { final = true; break; }
      phase = (int)(rec.x & 63);
      fw = (rec.x >> META_FW) & 1; dash = (rec.x >> META_DASH) & 1;
      end = (rec.x >> META_END) & 255; tau = (rec.x >> META_TAU) & 7; diff = (rec.x >> META_DIFF) & 7;
      sref = score;
      pos = (int)(rec.y & 255);
   }
   // a hit: the text position of the read's first letter, one suffix; or nothing
   if (final) key = ((unsigned long long)read << A.key_shift) | ((unsigned long long)p0 << 4) | (unsigned long long)score;
}

// Steps two walks of one entry have in common: the positions of the walk (in the order of the phases, from the
// node's state) up to and including the first one at which the two places of the text differ (one bwt_query_all
// there serves both).  E / Eo: the two walks' mismatch masks, TD: where their windows differ (planes_both), else
// the letters are compared on the byte text.  The caller caps the result by the lengths of the two walks.
template <int NCH>
__device__ __forceinline__ uint32_t verify_walked(const VerifyArgs &A, const uint64_t (&E)[NCH], const uint64_t (&Eo)[NCH],
                                                  const uint64_t (&QN)[NCH], const uint64_t (&TD)[NCH], bool planes_both,
                                                  int64_t p0, int64_t p0_o, int row, int phase0, int pos0, int end0, bool fw0)
{
   uint64_t M[NCH], Dx[NCH];
#pragma unroll
   for (int c = 0; c < NCH; c++) { M[c] = E[c] | Eo[c] | QN[c]; Dx[c] = E[c] ^ Eo[c]; }
   int phase = phase0, pos = pos0, end = end0;
   bool fw = fw0, parted = false;
   uint32_t walked = 0;
   for (;;) {
      for (;;) {
         const int a = fw ? pos : end, b = fw ? end : pos;
         const int e = first_event<NCH>(M, a, b, fw);
         if (e < 0) { walked += (uint32_t)(b - a + 1); break; }
         walked += (uint32_t)(fw ? e - pos : pos - e) + 1;
         bool differ = test_bit<NCH>(Dx, e);
         if (planes_both) differ = test_bit<NCH>(TD, e);
         else if (!differ) {
            const int64_t ta = p0 + e, tb = p0_o + e;
            const int la = (ta < 0 || ta >= A.ix.n) ? SYMS : A.text[ta], lb = (tb < 0 || tb >= A.ix.n) ? SYMS : A.text[tb];
            differ = la != lb;
         }
         if (differ) { parted = true; break; }
         if (e == end) break;
         pos = e + (fw ? 1 : -1);
      }
      if (parted) break;
      const uint2 rec = A.phase_rec[row * MAX_PHASES + phase];
      if (rec.y >> 8) break;
      phase = (int)(rec.x & 63);
      fw = (rec.x >> META_FW) & 1;
      end = (rec.x >> META_END) & 255;
      pos = (int)(rec.y & 255);
   }
   return walked;
}

// The same from the mask of the positions where the two windows differ (both lanes hold their window's planes):
// the first such position in the order of the walk, no stop at the positions where only the query differs.
template <int NCH>
__device__ __forceinline__ uint32_t verify_walked_td(const VerifyArgs &A, const uint64_t (&TD)[NCH], int row, int phase0, int pos0,
                                                     int end0, bool fw0)
{
   int phase = phase0, pos = pos0, end = end0;
   bool fw = fw0;
   uint32_t walked = 0;
   for (;;) {
      const int a = fw ? pos : end, b = fw ? end : pos;
      const int e = first_event<NCH>(TD, a, b, fw);
      if (e >= 0) return walked + (uint32_t)(fw ? e - pos : pos - e) + 1;
      walked += (uint32_t)(b - a + 1);
      const uint2 rec = A.phase_rec[row * MAX_PHASES + phase];
      if (rec.y >> 8) return walked;
      phase = (int)(rec.x & 63);
      fw = (rec.x >> META_FW) & 1;
      end = (rec.x >> META_END) & 255;
      pos = (int)(rec.y & 255);
   }
}

// The value of the other lane of a pair (lanes 2k and 2k + 1): a DPP move inside the quad, no trip through the LDS
// crossbar (`__shfl_xor(v, 1)` becomes ds_bpermute_b32).
__device__ __forceinline__ uint32_t pair_swap(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true); }
__device__ __forceinline__ uint64_t pair_swap(uint64_t v) { return (uint64_t)pair_swap((uint32_t)v) | ((uint64_t)pair_swap((uint32_t)(v >> 32)) << 32); }
__device__ __forceinline__ int64_t pair_swap(int64_t v) { return (int64_t)pair_swap((uint64_t)v); }

// Per block: steps in the low half, windows in the high half (k_sum_steps adds the words up: the halves of its 64
// sums stay below 2^32 for launches of up to 2^38 steps).  Ends with a workgroup barrier.
__device__ __forceinline__ void verify_block_sums(uint32_t steps, uint32_t windows /* suffixes this lane replayed */, unsigned long long *out)
{
   for (int o = 32; o > 0; o >>= 1) { steps += (uint32_t)__shfl_xor((int)steps, o); windows += (uint32_t)__shfl_xor((int)windows, o); }
   __shared__ uint32_t s_steps[4], s_windows[4];
   if ((threadIdx.x & 63) == 0) { s_steps[threadIdx.x >> 6] = steps; s_windows[threadIdx.x >> 6] = windows; }
   __syncthreads();
   if (threadIdx.x == 0) out[blockIdx.x] = ((unsigned long long)s_steps[0] + s_steps[1] + s_steps[2] + s_steps[3]) |
                                           (((unsigned long long)s_windows[0] + s_windows[1] + s_windows[2] + s_windows[3]) << 32);
}

// A slot of a list the lanes of a block fill together: the lanes of the wave that hold `flag` get consecutive slots from
// `counter` (in LDS; one atomic per wave).  Converged call: every lane of the wave executes it.
__device__ __forceinline__ uint32_t wave_slot(bool flag, uint32_t *counter, int lane)
{
   const unsigned long long m = __ballot(flag);
   uint32_t base = 0;
   if (lane == 0 && m != 0) base = atomicAdd(counter, (uint32_t)__popcll(m));
   base = (uint32_t)__shfl((int)base, 0);
   return base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
}

// A block takes a tile of A.tile entries of the list.  It first sorts them by what they are: candidates of two
// suffixes (PAIRS only), candidates of one suffix, hits of k_search itself (intervals: listed for the locate stage),
// candidates of more suffixes (option verify >= 4: listed for k_verify_group).  The suffixes to replay then fill the
// lanes densely, 256 per round: two adjacent lanes per pair (lane q replays suffix q), one lane per single; with
// one lane (or two) per ENTRY a third of the lanes had nothing to replay.  The two walks of a pair share the nodes of
// the reference's search tree as long as the two places of the text spell the same letters along the walk: those
// steps are counted once.  Without PAIRS every candidate holds one suffix.
// The order of the keys and of the lists inside a block's slots depends on the order the waves arrive in: the sort
// that follows makes the result the same.
template <int NCH, bool PAIRS>
__global__ void __launch_bounds__(256) k_verify(VerifyArgs A)
{
   const int lane = threadIdx.x & 63;
   __shared__ uint16_t s_pairs[VERIFY_TILE_MAX], s_singles[VERIFY_TILE_MAX];
   __shared__ uint32_t s_n[6];         // pairs, singles, intervals, big, wide, keys
   if (threadIdx.x < 6) s_n[threadIdx.x] = 0;
   __syncthreads();
   const int64_t tile0 = blockIdx.x * (int64_t)A.tile;
   for (int e0 = 0; e0 < A.tile; e0 += 256) {
      const int e = e0 + (int)threadIdx.x;
      const int64_t i = tile0 + e;
      int kind = 0;                    // 1: pair, 2: single, 3: interval, 4: three or four suffixes, 5: more
      if (e < A.tile && i < A.n_list) {
         const Node &nd = A.list[i];
         const uint32_t w2 = nd.w[2], w3 = nd.w[3], w5 = nd.w[5];
         if (w5 >> META_CAND) kind = !PAIRS ? 2 : w2 == 2 ? 1 : (w2 > 4 || (A.split3 && w2 == 3)) ? 5 : w2 > 2 ? 4 : 2;
         else if ((w2 | ((w3 >> 16) & 0xff)) != 0) kind = 3;      // (else a hole of the list)
      }
      uint32_t s;
      if (PAIRS) { s = wave_slot(kind == 1, &s_n[0], lane); if (kind == 1) s_pairs[s] = (uint16_t)e; }
      s = wave_slot(kind == 2, &s_n[1], lane); if (kind == 2) s_singles[s] = (uint16_t)e;
      s = wave_slot(kind == 3, &s_n[2], lane); if (kind == 3) A.iv_idx[(size_t)blockIdx.x * A.tile + s] = (uint32_t)i;
      if (PAIRS && A.big_idx != nullptr) { s = wave_slot(kind == 4, &s_n[3], lane); if (kind == 4) A.big_idx[(size_t)blockIdx.x * A.tile + s] = (uint32_t)i; }
      if (PAIRS && A.wide_idx != nullptr) { s = wave_slot(kind == 5, &s_n[4], lane); if (kind == 5) A.wide_idx[(size_t)blockIdx.x * A.tile + s] = (uint32_t)i; }
   }
   __syncthreads();
   const int n_pair_lanes = PAIRS ? 2 * (int)s_n[0] : 0, total = n_pair_lanes + (int)s_n[1];
   constexpr bool KEEP_PLANES = PAIRS && NCH <= 2;      // (longer reads: the registers cost more than the loads, 4.6 -> 4.9 ms at 150 nt)
   uint32_t steps_sum = 0, windows = 0;
   // The entry and the suffix array's entry of a lane's NEXT round are fetched before this round's window is: a round then
   // waits for one trip to memory (the window, the next round's suffix-array entry beside it) instead of three in a row.
   auto fetch = [&](int base_, Node &nd_, int64_t &sa_) {
      const int idx_ = base_ + lane;
      sa_ = 0;
      if (idx_ < total) {
         const bool pair_ = idx_ < n_pair_lanes;
         const int e_ = pair_ ? s_pairs[idx_ >> 1] : s_singles[idx_ - n_pair_lanes];
         nd_ = A.list[tile0 + e_];
         sa_ = sa_entry(A.ix.sa, A.ix.sar_bits, ((int64_t)nd_.w[0] | ((int64_t)(nd_.w[3] & 0xff) << 32)) + (pair_ ? (idx_ & 1) : 0));
      }
   };
   Node nd, nd_next;
   int64_t sa_val, sa_next;
   fetch((int)(threadIdx.x & ~63u), nd, sa_val);
   for (int base = (int)(threadIdx.x & ~63u); base < total; base += 256) {      // (per wave: a wave with nothing left to do leaves)
      const int idx = base + lane;
      const bool mine = idx < total;                    // this lane replays a suffix
      const bool pair = idx < n_pair_lanes;
      const int q = pair ? (idx & 1) : 0;
      fetch(base + 256, nd_next, sa_next);
      uint32_t steps = 0;                 // this walk's steps (0 for a lane without a suffix)
      unsigned long long key = ~0ull;
      int64_t p0 = 0;
      uint32_t read = 0;
      int row = 0;
      int phase0 = 0, pos0 = 0, end0 = 0;
      bool fw0 = false;
      uint64_t E[NCH], S[NCH], QN[NCH];
#pragma unroll
      for (int c = 0; c < NCH; c++) { E[c] = 0; S[c] = 0; QN[c] = 0; }
      uint32_t tw[3][2 * NCH];            // the window's planes, kept for the pair (where the two places of the text differ)
#pragma unroll
      for (int p = 0; p < 3; p++)
#pragma unroll
         for (int w = 0; w < 2 * NCH; w++) tw[p][w] = 0;
      bool planes = false;                // tw holds this lane's window
      if (mine) verify_replay<NCH, KEEP_PLANES>(A, nd, q, sa_val, E, S, QN, tw, planes, steps, key, p0, read, row, phase0, pos0, end0, fw0);
      if (PAIRS) {
         // ---- the pair: what the partner found (lanes 2k and 2k+1 of a wave; every lane takes part in the exchange)
         const uint32_t steps_o = pair_swap(steps);
         const int64_t p0_o = pair_swap(p0);
         uint64_t Eo[NCH];
#pragma unroll
         for (int c = 0; c < NCH; c++) Eo[c] = pair_swap(E[c]);
         // where the two windows differ, letter by letter, from the planes in registers (both lanes on the fast path)
         uint64_t TD[NCH];
#pragma unroll
         for (int c = 0; c < NCH; c++) TD[c] = 0;
         bool planes_both = false;
         if (KEEP_PLANES) {
            const uint32_t planes_o = pair_swap((uint32_t)planes);
            planes_both = planes && planes_o != 0;
            uint32_t dw[2 * NCH];
#pragma unroll
            for (int w = 0; w < 2 * NCH; w++) {
               dw[w] = 0;
#pragma unroll
               for (int p = 0; p < 3; p++) dw[w] |= tw[p][w] ^ pair_swap(tw[p][w]);
            }
#pragma unroll
            for (int c = 0; c < NCH; c++) TD[c] = (uint64_t)dw[2 * c] | ((uint64_t)dw[2 * c + 1] << 32);
         }
         if (pair && q == 1) {
            const uint32_t wk = planes_both ? verify_walked_td<NCH>(A, TD, row, phase0, pos0, end0, fw0)
                                            : verify_walked<NCH>(A, E, Eo, QN, TD, false, p0, p0_o, row, phase0, pos0, end0, fw0);
            steps -= min(wk, min(steps, steps_o));
         }
      }
      steps_sum += steps;
      windows += mine ? 1u : 0u;
      const bool found = mine && key != ~0ull;
      const uint32_t s = wave_slot(found, &s_n[5], lane);
      if (found) A.stage[(size_t)blockIdx.x * A.stage_per + s] = key;
      nd = nd_next; sa_val = sa_next;
   }
   verify_block_sums(steps_sum, windows, A.n_ext);         // (barrier inside)
   if (threadIdx.x == 0) {
      A.stage_cnt[blockIdx.x] = s_n[5];
      A.iv_cnt[blockIdx.x] = s_n[2];
      if (PAIRS && A.big_idx != nullptr) A.big_cnt[blockIdx.x] = s_n[3];
      if (PAIRS && A.wide_idx != nullptr) A.wide_cnt[blockIdx.x] = s_n[4];
   }
}

// The per-block slots of the staging area made dense: out[off[b] ..) = the first cnt[b] keys of slot b.
__global__ void __launch_bounds__(256) k_stage_compact(const unsigned long long *stage, const uint32_t *cnt, const uint32_t *off, int per,
                                                       unsigned long long *out)
{
   const uint32_t n = cnt[blockIdx.x];
   for (uint32_t t = threadIdx.x; t < n; t += blockDim.x) out[off[blockIdx.x] + t] = stage[(size_t)blockIdx.x * per + t];
}

// The same for the per-block lists of entry numbers (slots of `per` numbers).
__global__ void k_list_compact(const uint32_t *idx, const uint32_t *cnt, const uint32_t *off, int per, uint32_t *dense)
{
   const uint32_t n = cnt[blockIdx.x];
   for (uint32_t t = threadIdx.x; t < n; t += blockDim.x) dense[off[blockIdx.x] + t] = idx[(size_t)blockIdx.x * per + t];
}

// Candidates of three to G suffixes: G lanes per entry, lane q replays suffix q.  The walks of one entry are paths
// of the reference's search tree from one node; the bwt_query_all calls they stand for are the nodes of the union
// of those paths: walk q adds its own steps minus the most it shares with any walk before it (two paths share
// their nodes up to and including the first position where their texts differ: verify_walked).  The keys of the hits
// go to the block's slot of the staging area (A.stage, A.stage_cnt: those of this launch), as in k_verify.
__device__ __forceinline__ int __reduce_max_csz(int v)
{
   for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
   return v;
}

template <int NCH, int G>
__global__ void __launch_bounds__(256) k_verify_group(VerifyArgs A, const uint32_t *dense, int64_t n_big, unsigned long long *n_ext_out)
{
   // 64 / G entries per wave, G adjacent lanes each (G = 3: 21 entries, the last lane of the wave idle)
   constexpr int PER_WAVE = 64 / G;
   const int lane = threadIdx.x & 63, slot = lane / G, base = slot * G;
   const int q = lane - base;
   const int64_t g = ((blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6) * PER_WAVE + slot;
   uint32_t steps = 0;
   unsigned long long key = ~0ull;
   int64_t p0 = 0, i = 0;
   uint32_t read = 0;
   int row = 0, phase0 = 0, pos0 = 0, end0 = 0, csz = 0;
   bool fw0 = false, planes = false;
   uint64_t E[NCH], S[NCH], QN[NCH];
#pragma unroll
   for (int c = 0; c < NCH; c++) { E[c] = 0; S[c] = 0; QN[c] = 0; }
   uint32_t tw[3][2 * NCH];
#pragma unroll
   for (int p = 0; p < 3; p++)
#pragma unroll
      for (int w = 0; w < 2 * NCH; w++) tw[p][w] = 0;
   constexpr bool KEEP_PLANES = NCH <= 3;               // (150 nt: 6.33 -> 5.38 ms for k_verify + k_verify_group; with k_verify keeping them too 5.71)
   const bool valid = slot < PER_WAVE && g < n_big;
   if (valid) { i = dense[g]; csz = (int)A.list[i].w[2]; }
   const bool mine = valid && q < csz;
   if (mine) {
      Node nd = A.list[i];
      const int64_t sa_val = sa_entry(A.ix.sa, A.ix.sar_bits, ((int64_t)nd.w[0] | ((int64_t)(nd.w[3] & 0xff) << 32)) + q);
      verify_replay<NCH, KEEP_PLANES>(A, nd, q, sa_val, E, S, QN, tw, planes, steps, key, p0, read, row, phase0, pos0, end0, fw0);
   }
   uint32_t best = 0;
   if (KEEP_PLANES) {
      // Every walk holds its window's planes (the letter-by-letter path fills them too).  Copies of a repeat mostly spell
      // the same letters over the whole read: a walk whose window equals that of the lane before is the same path of
      // the search tree and adds nothing; only the first walk of each distinct window (a representative) is compared
      // with the representatives before it.  Every lane takes part in the exchanges.
      const int m = mine ? (int)A.row_len[row] : 0;
      uint64_t TD[NCH];
      auto differ = [&](int src) {
         uint32_t dw[2 * NCH];
#pragma unroll
         for (int w = 0; w < 2 * NCH; w++) {
            dw[w] = 0;
#pragma unroll
            for (int p = 0; p < 3; p++) dw[w] |= tw[p][w] ^ (uint32_t)__shfl((int)tw[p][w], src);
         }
#pragma unroll
         for (int c = 0; c < NCH; c++) TD[c] = (uint64_t)dw[2 * c] | ((uint64_t)dw[2 * c + 1] << 32);
      };
      differ(lane > 0 ? lane - 1 : 0);
      bool dup = false;
      if (mine && q > 0) {
         uint64_t any = 0;
#pragma unroll
         for (int c = 0; c < NCH; c++) {
            const int hi = m - 64 * c;
            any |= TD[c] & (hi >= 64 ? ~0ull : hi > 0 ? (1ull << hi) - 1ull : 0ull);
         }
         dup = any == 0;
      }
      if (dup) best = steps;
      const bool rep = mine && !dup;
      const uint32_t reps = (uint32_t)((__ballot(rep) >> base) & ((1ull << G) - 1ull));
      uint32_t todo = (rep && q > 0) ? (reps & ((1u << q) - 1u)) : 0u;
      while (__any(todo != 0)) {
         const int src = todo ? base + (__ffs((int)todo) - 1) : lane;
         const uint32_t steps_o = (uint32_t)__shfl((int)steps, src);
         differ(src);
         if (todo) {
            const uint32_t wk = verify_walked_td<NCH>(A, TD, row, phase0, pos0, end0, fw0);
            best = max(best, min(wk, min(steps, steps_o)));
            todo &= todo - 1;
         }
      }
   } else {
      const int rounds = __reduce_max_csz(csz) - 1;      // (wave-uniform: the most suffixes any entry of the wave holds)
#pragma unroll 1
      for (int j = 0; j < rounds; j++) {             // every lane takes part in the exchanges
         const int src = base + j;
         const uint32_t steps_o = (uint32_t)__shfl((int)steps, src);
         uint64_t Eo[NCH], TD[NCH];
#pragma unroll
         for (int c = 0; c < NCH; c++) { Eo[c] = __shfl(E[c], src); TD[c] = 0; }
         const int64_t p0_o = __shfl(p0, src);
         if (mine && q > j) {
            const uint32_t wk = verify_walked<NCH>(A, E, Eo, QN, TD, false, p0, p0_o, row, phase0, pos0, end0, fw0);
            best = max(best, min(wk, min(steps, steps_o)));
         }
      }
   }
   steps -= best;
   const bool found = valid && key != ~0ull;
   const unsigned long long foundm = __ballot(found);
   __shared__ uint32_t s_found[4];
   if (lane == 0) s_found[threadIdx.x >> 6] = (uint32_t)__popcll(foundm);
   verify_block_sums(steps, mine ? 1u : 0u, n_ext_out);       // (barrier inside)
   uint32_t off = 0, total = 0;
#pragma unroll
   for (int k = 0; k < 4; k++) { if (k < (int)(threadIdx.x >> 6)) off += s_found[k]; total += s_found[k]; }
   if (found) A.stage[(size_t)blockIdx.x * STAGE_PER_BLOCK + off + (uint32_t)__popcll(foundm & ((1ull << lane) - 1ull))] = key;
   if (threadIdx.x == 0) A.stage_cnt[blockIdx.x] = total;
}

// The steps of a k_verify launch: 64 blocks add up the per-block counts, one sum per block (what the host adds up).
__global__ void __launch_bounds__(256) k_sum_steps(const unsigned long long *part, int64_t n, unsigned long long *out)
{
   unsigned long long v = 0;
   for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) v += part[i];
   for (int o = 32; o > 0; o >>= 1) v += (unsigned long long)__shfl_xor((long long)v, o);
   __shared__ unsigned long long s_sum[4];
   if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = v;
   __syncthreads();
   if (threadIdx.x == 0) out[blockIdx.x] = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
}

// Selection by a per-lane index, written over values (not over the array) so that the arrays stay
// in registers: a select chain over array elements is turned into an indexed stack access.
template <class T>
__device__ __forceinline__ T sel5v(T v0, T v1, T v2, T v3, T v4, int j)
{
   T r = v0;
   r = j == 1 ? v1 : r;
   r = j == 2 ? v2 : r;
   r = j == 3 ? v3 : r;
   r = j == 4 ? v4 : r;
   return r;
}
#define sel5(v, j) sel5v((v)[0], (v)[1], (v)[2], (v)[3], (v)[4], (j))

template <int CAP = QCAP>
__device__ __forceinline__ void lds_store_node(Node *q, uint32_t slot, const Node &nd)
{
   uint4 *d = reinterpret_cast<uint4 *>(q + (slot & (CAP - 1)));
   d[0] = make_uint4(nd.w[0], nd.w[1], nd.w[2], nd.w[3]);
   d[1] = make_uint4(nd.w[4], nd.w[5], nd.w[6], nd.w[7]);
}

template <int CAP = QCAP>
__device__ __forceinline__ void lds_load_node(const Node *q, uint32_t slot, Node &nd)
{
   const uint4 *s = reinterpret_cast<const uint4 *>(q + (slot & (CAP - 1)));
   uint4 a = s[0], b = s[1];
   nd.w[0] = a.x; nd.w[1] = a.y; nd.w[2] = a.z; nd.w[3] = a.w;
   nd.w[4] = b.x; nd.w[5] = b.y; nd.w[6] = b.z; nd.w[7] = b.w;
}

__device__ __forceinline__ void wave_sync()
{
   __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
   __builtin_amdgcn_wave_barrier();
}

struct WaveQueue {
   Node *q;
   Node *spill;
   uint32_t bot, top, spill_top, spill_cap;
   uint32_t n_spilled;        // chunks this wave moved to its spill region
   int *error;
};

// Converged call: every lane of the wave executes it.  Lanes with `flag` append `nd`.
template <int CAP = QCAP, int CHUNK = SPILL_CHUNK>
__device__ __forceinline__ void wave_push(WaveQueue &wq, bool flag, const Node &nd, int lane)
{
   if (wq.top - wq.bot + 64 > CAP) {
      // Spill the oldest CHUNK nodes (LIFO order is kept: they return only when the deque is empty).
      bool room = wq.spill_top + CHUNK <= wq.spill_cap;
      if (room) {
         for (int i = lane; i < CHUNK; i += 64) {
            Node t;
            lds_load_node<CAP>(wq.q, wq.bot + i, t);
            uint4 *d = reinterpret_cast<uint4 *>(wq.spill + wq.spill_top + i);
            d[0] = make_uint4(t.w[0], t.w[1], t.w[2], t.w[3]);
            d[1] = make_uint4(t.w[4], t.w[5], t.w[6], t.w[7]);
         }
         wq.spill_top += CHUNK;
         wq.n_spilled++;
      } else if (lane == 0) {
         *wq.error = 1;      // results are incomplete; the host reports the failure
      }
      wq.bot += CHUNK;
      wave_sync();
   }
   unsigned long long m = __ballot(flag);
   if (flag) lds_store_node<CAP>(wq.q, wq.top + (uint32_t)__popcll(m & ((1ull << lane) - 1ull)), nd);
   wq.top += (uint32_t)__popcll(m);
}

// (k_search itself is in gm_search.hpp.)

// ------------------------------------------------------------------------------------------------
// K3: locate.  The hits k_verify / k_verify_group finished are keys already (k_stage_compact).  The hits of k_search
// itself are intervals: hit_off = exclusive scan of their sizes; one thread per interval reads its suffix-array
// entries (sar_get_range, src/index_sar.c:121-161), up to PLACE_INLINE of them here, longer ones go to a list that
// k_place_long works through a workgroup per interval.  After a verified search the intervals are few and come as a
// list of entry numbers (`dense`); without k_verify every entry of the list is looked at.
// key = read << (pos_bits + 4) | text position << 4 | score: one radix sort then orders the hits by
// read and, inside a read, by text position.
// ------------------------------------------------------------------------------------------------
constexpr int PLACE_INLINE = 16;

// `dense` (may be nullptr): the entries to look at, as a list of entry numbers; sizes / hit_off / longer then count
// in that list.
__global__ void k_hit_sizes(const Node *hits, const uint32_t *dense, int64_t n, int64_t *sizes)
{
   int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
   if (t >= n) return;
   const int64_t i = dense ? dense[t] : t;
   sizes[t] = (int64_t)hits[i].w[2] | ((int64_t)((hits[i].w[3] >> 16) & 0xff) << 32);
}

__global__ void __launch_bounds__(256) k_place(IndexDev ix, const Node *hits, const uint32_t *dense, int64_t n,
                                               const int64_t *sizes, const int64_t *hit_off, int pos_bits,
                                               unsigned long long *keys, unsigned int *longer, unsigned int *n_longer, unsigned int longer_cap,
                                               const uint8_t *score_add /* per read (blocksc_trail adds the N count, :94-97) or nullptr */)
{
   const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
   if (t >= n) return;
   const int64_t sz = sizes[t];
   if (sz == 0) return;
   if (sz > PLACE_INLINE) {
      const unsigned int slot = atomicAdd(n_longer, 1u);
      if (slot < longer_cap) longer[slot] = (unsigned int)t;      // (cap = slots / PLACE_INLINE: cannot overflow)
      return;
   }
   NodeR h;
   unpack_node(hits[dense ? dense[t] : t], h);
   const int64_t off = hit_off[t];
   const unsigned long long hi = ((unsigned long long)h.read << (pos_bits + 4)) | (unsigned long long)(h.score + (score_add ? score_add[h.read] : 0));
   for (int64_t j = 0; j < sz; j++)
      keys[off + j] = hi | ((unsigned long long)sa_entry(ix.sa, ix.sar_bits, h.fp + j) << 4);
}

__global__ void __launch_bounds__(256) k_place_long(IndexDev ix, const Node *hits, const uint32_t *dense, const int64_t *sizes,
                                                    const int64_t *hit_off, int pos_bits,
                                                    unsigned long long *keys, const unsigned int *longer, const unsigned int *n_longer,
                                                    const uint8_t *score_add)
{
   const unsigned int n = *n_longer;
   for (unsigned int e = blockIdx.x; e < n; e += gridDim.x) {
      const int64_t t = longer[e];
      NodeR h;
      unpack_node(hits[dense ? dense[t] : t], h);
      const int64_t off = hit_off[t], sz = sizes[t];
      const unsigned long long hi = ((unsigned long long)h.read << (pos_bits + 4)) | (unsigned long long)(h.score + (score_add ? score_add[h.read] : 0));
      for (int64_t j = threadIdx.x; j < sz; j += blockDim.x)    // consecutive threads read consecutive packed entries
         keys[off + j] = hi | ((unsigned long long)sa_entry(ix.sa, ix.sar_bits, h.fp + j) << 4);
   }
}

// Sorted keys -> structure of arrays; strand and coordinates by the rule of txt_pos_to_str
// (src/index_txt.c:492-505) with seq_bisect (:770-787).  Also the CSR offsets of the reads: the thread of
// the first key of a read writes that read's offset and those of the reads without hits before it.
__global__ void k_unpack_locs(IndexDev ix, const unsigned long long *keys, int64_t n_locs, int pos_bits, int64_t n_reads,
                              int64_t *read_first, int64_t *loc_read, int64_t *loc_pos, int64_t *loc_fwd, int64_t *loc_seq,
                              int64_t *loc_seqpos, uint8_t *loc_strand, uint8_t *loc_score)
{
   int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
   if (t >= n_locs) return;
   unsigned long long k = keys[t];
   int64_t pos = (int64_t)((k >> 4) & ((1ull << pos_bits) - 1ull));
   const int64_t read = (int64_t)(k >> (pos_bits + 4));
   const int64_t before = t == 0 ? -1 : (int64_t)(keys[t - 1] >> (pos_bits + 4));
   for (int64_t r = before + 1; r <= read; r++) read_first[r] = t;
   if (t == n_locs - 1)
      for (int64_t r = read + 1; r <= n_reads; r++) read_first[r] = n_locs;
   loc_read[t] = read;
   loc_pos[t] = pos;
   loc_score[t] = (uint8_t)(k & 15);
   int strand = pos >= ix.n / 2;
   int64_t f = strand ? ix.n - 2 - pos : pos;
   loc_strand[t] = (uint8_t)strand;
   loc_fwd[t] = f;
   int64_t lo = 0, hi = ix.seq_cnt;
   while (hi - lo >= 2) { int64_t mid = (lo + hi) / 2; if (f < ix.seq_beg[mid]) hi = mid; else lo = mid; }
   loc_seq[t] = lo;
   loc_seqpos[t] = f - ix.seq_beg[lo] + 1;
}

// Interval hits -> sortable keys (read, fp) and, after the sort, structure of arrays with the
// mismatch bitmap rebuilt from the recorded positions and the query's N positions.
// alias (or nullptr): the query a hit is reported under (gpumap_ann_build: a k-mer searched as two queries).
__global__ void k_hit_keys(const Node *hits, int64_t n_hits, unsigned long long *keys, unsigned int *idx, const uint32_t *alias)
{
   int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
   if (i >= n_hits) return;
   NodeR h;
   unpack_node(hits[i], h);
   // reads < 2^24 - 1 per run, fp < 2^39 (n < 2^34); holes of the list (sz = 0) sort behind every hit, also where only the
   // bits [39, 63) are sorted (gpumap_ann_build; not [40, 64): hipcub's radix sort with begin_bit > 0 and end_bit = 64
   // returns unsorted keys for 10^4 .. 10^6 items on this ROCm, tools/sortpairs_check.hip)
   keys[i] = h.sz > 0 ? (((unsigned long long)(alias ? alias[h.read] : h.read) << 39) | (unsigned long long)h.fp) : ~0ull;
   idx[i] = (unsigned int)i;
}

__global__ void k_unpack_hits(const Node *hits, const unsigned int *idx, int64_t n_hits, const uint8_t *reads,
                              const uint16_t *read_len, int stride, int64_t *hit_read, int64_t *hit_fp,
                              int64_t *hit_rp, int64_t *hit_sz, int64_t *hit_dp, int64_t *hit_score,
                              uint64_t *hit_align_out, const uint8_t *score_add, const uint32_t *alias)
{
   int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
   if (i >= n_hits) return;
   NodeR h;
   unpack_node(hits[idx[i]], h);
   int m = read_len[h.read];
   hit_read[i] = alias ? alias[h.read] : h.read; hit_fp[i] = h.fp; hit_rp[i] = h.rp; hit_sz[i] = h.sz;
   hit_dp[i] = m;                       // one extension per letter (dp + 1 per bwt_query_all)
   hit_score[i] = h.score + (score_add ? score_add[h.read] : 0);
   uint64_t al[4];
   hit_align(h.mm, h.score, reads + (size_t)h.read * stride, m, al);
   for (int w = 0; w < 4; w++) hit_align_out[4 * i + w] = al[w];
}

__global__ void k_status(const uint16_t *read_len, const uint16_t *len_row, int64_t n_reads, int64_t *status)
{
   int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
   if (r >= n_reads) return;
   int m = read_len[r];
   status[r] = (m <= MAX_READ && len_row[m] != NO_ROW) ? 0 : -1;
}

// The radix sort may leave out the lowest bits of the key when that saves it a pass over the keys (the score, which
// the text position determines, and up to three bits of the position): keys that agree in all bits from `shift` on are
// neighbours then, in any order.  The thread of the first key of such a run puts the run in order; runs are short (at
// most 2^(shift - 4) places of the text, each found from a few seeds).  The other threads only look at bits that no
// reordering inside a run changes.
__device__ __forceinline__ void order_run_from(unsigned long long *keys, int64_t n, int shift, int64_t t)
{
   const unsigned long long g = keys[t] >> shift;
   if (t > 0 && (keys[t - 1] >> shift) == g) return;
   int64_t e = t + 1;
   while (e < n && (keys[e] >> shift) == g) e++;
   for (int64_t i = t + 1; i < e; i++) {
      const unsigned long long k = keys[i];
      int64_t j = i;
      while (j > t && keys[j - 1] > k) { keys[j] = keys[j - 1]; j--; }
      if (j != i) keys[j] = k;
   }
}

__global__ void k_order_runs(unsigned long long *keys, int64_t n, int shift)
{
   const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
   if (t < n) order_run_from(keys, n, shift, t);
}

// ------------------------------------------------------------------------------------------------
// Round 3: the radix sort only brings the hits of a read together (the read's bits: 3 passes of 8 bits for 12.5 M
// reads instead of 7 over the whole key); the hits of a read are then put in order where they lie.  A read has 3.3 hits
// on average at GRCh38 size and a few have thousands:
//   k_order_first   a thread per key: in a run (keys that agree from `shift` up) of up to ORDER_SMALL keys every key
//                   finds its place by counting the keys of the run below it; longer runs are noted in the slot of
//                   the block their first key lies in (no global atomics: a block of 256 keys can hold the starts of
//                   ORDER_SLOT such runs at most)
//   k_order_big     a workgroup per noted run: up to ORDER_LDS keys by a bitonic network in LDS, longer runs by radix
//                   passes of 8 bits through the sort's other buffer
// ------------------------------------------------------------------------------------------------
constexpr int ORDER_SMALL = 32, ORDER_SLOT = 8, ORDER_LDS = 4096;

// src -> dst (the sort's two buffers): every key of a run of up to ORDER_SMALL keys goes to its place in the run, which
// is the number of keys of the run below it (a thread per key, the block's keys and ORDER_SMALL on either side in LDS:
// no thread waits for another one's chain of loads, and a wavefront takes as many rounds as its longest run has keys).
// The keys of longer runs are copied as they are and the runs noted for k_order_big.
__global__ void __launch_bounds__(256) k_order_first(const unsigned long long *src, unsigned long long *dst, int64_t n, int shift,
                                                     uint32_t *big_start, uint32_t *big_cnt)
{
   __shared__ unsigned long long s_keys[256 + 2 * ORDER_SMALL];
   __shared__ uint32_t s_cnt;
   const int tid = threadIdx.x;
   const int64_t base = blockIdx.x * (int64_t)256 - ORDER_SMALL;        // s_keys[j] = src[base + j]
   if (tid == 0) s_cnt = 0;
   for (int j = tid; j < 256 + 2 * ORDER_SMALL; j += 256) {
      const int64_t i = base + j;
      s_keys[j] = (i >= 0 && i < n) ? src[i] : 0ull;
   }
   __syncthreads();
   const int64_t t = blockIdx.x * (int64_t)256 + tid;
   if (t < n) {
      const int me = ORDER_SMALL + tid;
      const unsigned long long k = s_keys[me], g = k >> shift;
      int back = 0, fwd = 0;
      while (back < ORDER_SMALL && t - back - 1 >= 0 && (s_keys[me - back - 1] >> shift) == g) back++;
      while (fwd < ORDER_SMALL && t + fwd + 1 < n && (s_keys[me + fwd + 1] >> shift) == g) fwd++;
      if (back + fwd + 1 > ORDER_SMALL) {
         dst[t] = k;
         if (back == 0) {
            const uint32_t at = atomicAdd(&s_cnt, 1u);                  // (LDS)
            big_start[(size_t)blockIdx.x * ORDER_SLOT + at] = (uint32_t)tid;
         }
      } else {
         int rank = 0;
         for (int j = me - back; j <= me + fwd; j++) rank += s_keys[j] < k;      // (the keys of a run differ: read and place)
         dst[t - back + rank] = k;
      }
   }
   __syncthreads();
   if (tid == 0) big_cnt[blockIdx.x] = s_cnt;
}

// One workgroup sorts keys[s, e) by the bits [lo_bit, hi_bit) with passes of 8 bits through alt[s, e) (stable: the
// keys of a tile of 256 are ranked by ballots inside their wavefront and by the wavefronts' counts across them).
__device__ __forceinline__ void block_radix_run(unsigned long long *keys, unsigned long long *alt, int64_t s, int64_t e, int lo_bit, int hi_bit,
                                                uint32_t *s_hist /* [256] */, uint32_t (*s_wcnt)[256] /* [4][256] */, uint32_t *s_part /* [4] */)
{
   const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
   const unsigned long long lt = (1ull << lane) - 1ull;
   unsigned long long *src = keys, *dst = alt;
   for (int bit = lo_bit; bit < hi_bit; bit += 8) {
      const int nb = min(8, hi_bit - bit);
      const uint32_t mask = (1u << nb) - 1u;
      s_hist[tid] = 0;
      for (int w = 0; w < 4; w++) s_wcnt[w][tid] = 0;
      __syncthreads();
      for (int64_t i = s + tid; i < e; i += 256) atomicAdd(&s_hist[(uint32_t)(src[i] >> bit) & mask], 1u);
      __syncthreads();
      {  // exclusive scan of the 256 counts
         const uint32_t v = s_hist[tid];
         uint32_t incl = v;
         for (int o = 1; o < 64; o <<= 1) { const uint32_t x = (uint32_t)__shfl_up((int)incl, o); if (lane >= o) incl += x; }
         if (lane == 63) s_part[wid] = incl;
         __syncthreads();
         uint32_t before = 0;
         for (int w = 0; w < wid; w++) before += s_part[w];
         s_hist[tid] = before + incl - v;
         __syncthreads();
      }
      for (int64_t base = s; base < e; base += 256) {
         const int64_t i = base + tid;
         const bool have = i < e;
         const unsigned long long k = have ? src[i] : 0ull;
         const uint32_t d = (uint32_t)(k >> bit) & mask;
         unsigned long long same = __ballot(have);
         for (int b = 0; b < nb; b++) {
            const unsigned long long bb = __ballot((d >> b) & 1u);
            same &= ((d >> b) & 1u) ? bb : ~bb;
         }
         const uint32_t rank = (uint32_t)__popcll(same & lt);
         if (have && rank == 0) s_wcnt[wid][d] = (uint32_t)__popcll(same);
         __syncthreads();
         if (have) {
            uint32_t off = s_hist[d] + rank;
            for (int w = 0; w < wid; w++) off += s_wcnt[w][d];
            dst[s + off] = k;
         }
         __syncthreads();
         {
            uint32_t tot = 0;
            for (int w = 0; w < 4; w++) { tot += s_wcnt[w][tid]; s_wcnt[w][tid] = 0; }
            s_hist[tid] += tot;
         }
         __syncthreads();
      }
      unsigned long long *x = src; src = dst; dst = x;
   }
   if (src != keys)
      for (int64_t i = s + tid; i < e; i += 256) keys[i] = src[i];
   __syncthreads();
}

__global__ void __launch_bounds__(256) k_order_big(unsigned long long *keys, unsigned long long *alt, int64_t n, int shift, int lo_bit,
                                                   const uint32_t *big_start, const uint32_t *big_cnt, int64_t n_blocks)
{
   __shared__ unsigned long long s_keys[ORDER_LDS];
   __shared__ uint32_t s_hist[256];
   __shared__ uint32_t s_wcnt[4][256];
   __shared__ uint32_t s_part[4];
   __shared__ int64_t s_end;
   const int tid = threadIdx.x;
   for (int64_t b = blockIdx.x; b < n_blocks; b += gridDim.x) {
      const uint32_t cnt = big_cnt[b];
      for (uint32_t j = 0; j < cnt; j++) {
         const int64_t s = b * 256 + (int64_t)big_start[(size_t)b * ORDER_SLOT + j];
         if (tid == 0) {
            // the end of the run: the keys agree from `shift` up and are sorted by those bits
            const unsigned long long g = keys[s] >> shift;
            int64_t step = ORDER_SMALL, lo = s + ORDER_SMALL;           // keys[lo] belongs to the run
            while (lo + step < n && (keys[lo + step] >> shift) == g) { lo += step; step <<= 1; }
            int64_t hi = min(n, lo + step);                             // keys[hi] does not (or hi = n)
            while (hi - lo > 1) { const int64_t m = (lo + hi) >> 1; if ((keys[m] >> shift) == g) lo = m; else hi = m; }
            s_end = hi;
         }
         __syncthreads();
         const int64_t e = s_end;
         const int len = (int)min(e - s, (int64_t)ORDER_LDS + 1);
         if (e - s <= ORDER_LDS) {
            int m = 64;
            while (m < len) m <<= 1;
            for (int i = tid; i < m; i += 256) s_keys[i] = i < len ? keys[s + i] : ~0ull;
            __syncthreads();
            for (int k = 2; k <= m; k <<= 1)
               for (int jj = k >> 1; jj > 0; jj >>= 1) {
                  for (int i = tid; i < m; i += 256) {
                     const int l = i ^ jj;
                     if (l > i) {
                        const unsigned long long a = s_keys[i], c = s_keys[l];
                        const bool up = (i & k) == 0;
                        if ((a > c) == up) { s_keys[i] = c; s_keys[l] = a; }
                     }
                  }
                  __syncthreads();
               }
            for (int i = tid; i < len; i += 256) keys[s + i] = s_keys[i];
            __syncthreads();
         } else {
            block_radix_run(keys, alt, s, e, lo_bit, shift, s_hist, s_wcnt, s_part);
         }
      }
   }
}

// ------------------------------------------------------------------------------------------------
// Round 3, second form: two radix passes over the top 16 bits of the key bring the hits of a SEGMENT of reads
// together (256 consecutive reads where a chunk has 2^24 of them); a workgroup then orders each segment in LDS by
// the rest of the key.  The traffic of the ordering stage goes from 7 passes over the keys to 2 + 1; what a
// segment costs depends on its number of keys only, not on how they spread over its reads.
//   k_seg_first      where each segment begins (a thread per key, as k_read_first)
//   k_order_segments<E>  a workgroup per segment of up to 256 * E keys (launched for E = 4, 16, 32, each taking the segments
//                    the one before left); longer segments by radix passes through the sort's other buffer
// ------------------------------------------------------------------------------------------------
__global__ void k_seg_first(const unsigned long long *keys, int64_t n, int shift, int64_t n_seg, uint32_t *seg_first)
{
   const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
   if (t >= n) return;
   const int64_t s = (int64_t)(keys[t] >> shift);
   const int64_t before = t == 0 ? -1 : (int64_t)(keys[t - 1] >> shift);
   for (int64_t x = before + 1; x <= s && x <= n_seg; x++) seg_first[x] = (uint32_t)t;
   if (t == n - 1)
      for (int64_t x = s + 1; x <= n_seg; x++) seg_first[x] = (uint32_t)n;
}

// E keys per thread in registers (a segment of up to 256 * E keys), LSD passes of 8 bits over [lo_bit, shift): a wavefront
// owns a quarter of the segment in order; a key's rank among the wavefront's keys of its digit comes from ballots
// (which lanes hold the same digit) and a running count per wavefront and digit in LDS, the digit's base from a scan
// over the 256 digits and the four wavefronts; the keys change places through one LDS array.  Stable, so the passes
// add up to the order by all the bits.  (A bitonic network in LDS was the first form: 110 LDS accesses per key at
// 1024 keys against 15 here; it ran at the speed of the LDS pipe, 1.0 ms for the segments of a GRCh38-size chunk.)
// What follows the sort of a segment, by the workgroup that holds it: the keys that agree from `lo_bit` up (bits the
// passes left out) are put in order, and the CSR offsets of the segment's reads are written (what k_read_first does
// for the whole array: the thread of the first key of a read writes that read's offset and those of the reads
// without hits before it; the reads behind the segment's last key get its end).
struct SegOut {
   int64_t *read_first;       // nullptr: the caller derives the offsets itself
   int64_t n_reads, n_keys;
   int read_shift;            // the read's bits begin here
   int seg_reads_log;         // a segment holds 2^this reads
   uint8_t *packed;           // not nullptr: the segment's keys leave as the packed result (5 bytes each: the bits below the
                              // read's), and the 8-byte keys are not written back
};

// The keys k[0, len) of the segment that begins at key `first_key`, as bytes [5 * first_key, 5 * (first_key + len)) of the
// packed result: whole words by a thread each, the few bytes before the first and behind the last whole word one by one
// (the neighbouring segments' workgroups write theirs the same way: no word is written by two of them).
__device__ __forceinline__ void segment_write_packed(const unsigned long long *k, int len, int64_t first_key, int low_bits, uint8_t *packed, int tid)
{
   const unsigned long long m = (1ull << low_bits) - 1ull;
   const int64_t g0 = 5 * first_key, g1 = g0 + 5 * (int64_t)len;
   const int64_t w0 = (g0 + 3) >> 2, w1 = g1 >> 2;                 // whole words [w0, w1)
   auto byte_at = [&](uint32_t l) -> uint32_t {                    // byte l of the segment's part
      const uint32_t i = l / 5u, r = l - 5u * i;
      return (uint32_t)((k[i] & m) >> (8u * r)) & 0xffu;
   };
   if (w0 < w1) {
      uint32_t *words = reinterpret_cast<uint32_t *>(packed);
      for (int64_t w = w0 + tid; w < w1; w += 256) {
         const uint32_t l = (uint32_t)(4 * w - g0);
         words[w] = byte_at(l) | (byte_at(l + 1) << 8) | (byte_at(l + 2) << 16) | (byte_at(l + 3) << 24);
      }
      const int head = (int)(4 * w0 - g0), tail = (int)(g1 - 4 * w1);
      if (tid < head) packed[g0 + tid] = (uint8_t)byte_at((uint32_t)tid);
      if (tid < tail) packed[4 * w1 + tid] = (uint8_t)byte_at((uint32_t)(4 * w1 - g0) + (uint32_t)tid);
   } else {
      for (int64_t g = g0 + tid; g < g1; g += 256) packed[g] = (uint8_t)byte_at((uint32_t)(g - g0));
   }
}

__device__ __forceinline__ void segment_finish(unsigned long long *k /* LDS or global */, int len, int64_t first_key, int64_t seg, int lo_bit,
                                               const SegOut &o, int tid)
{
   if (lo_bit > 4) {
      for (int i = tid; i < len; i += 256) {
         const unsigned long long g = k[i] >> lo_bit;
         if (i > 0 && (k[i - 1] >> lo_bit) == g) continue;
         int e = i + 1;
         while (e < len && (k[e] >> lo_bit) == g) e++;
         for (int x = i + 1; x < e; x++) {
            const unsigned long long v = k[x];
            int j = x;
            while (j > i && k[j - 1] > v) { k[j] = k[j - 1]; j--; }
            if (j != x) k[j] = v;
         }
      }
      __syncthreads();
   }
   if (o.read_first == nullptr) return;
   const int64_t r_lo = seg << o.seg_reads_log, r_hi = min(o.n_reads, (seg + 1) << o.seg_reads_log);
   for (int i = tid; i < len; i += 256) {
      const int64_t r = (int64_t)(k[i] >> o.read_shift);
      const int64_t before = i == 0 ? r_lo - 1 : (int64_t)(k[i - 1] >> o.read_shift);
      for (int64_t x = before + 1; x <= r; x++) o.read_first[x] = first_key + i;
   }
   if (tid == 0) {
      const int64_t last = len == 0 ? r_lo - 1 : (int64_t)(k[len - 1] >> o.read_shift);
      for (int64_t x = last + 1; x < r_hi; x++) o.read_first[x] = first_key + len;
      if (r_hi == o.n_reads) o.read_first[o.n_reads] = o.n_keys;
   }
}

// E keys per thread in registers (a segment of up to 256 * E keys), LSD passes of 8 bits over [lo_bit, shift): a wavefront
// owns a quarter of the segment in order; a key's rank among the wavefront's keys of its digit comes from ballots
// (which lanes hold the same digit) and a running count per wavefront and digit in LDS, the digit's base from a scan
// over the 256 digits and the four wavefronts; the keys change places through one LDS array.  Stable, so the passes
// add up to the order by all the bits.  (A bitonic network in LDS was the first form: 110 LDS accesses per key at
// 1024 keys against 15 here; it ran at the speed of the LDS pipe, 1.0 ms for the segments of a GRCh38-size chunk.)
// A launch takes the segments of more than len_above and at most 256 * E keys (-1: from empty ones on), the launch with
// longer_too also the longer ones (radix passes through the sort's other buffer).
template <int E>
__global__ void __launch_bounds__(256) k_order_segments(unsigned long long *keys, unsigned long long *alt, const uint32_t *seg_first, int lo_bit, int shift,
                                                        int len_above, bool longer_too, SegOut out)
{
   constexpr int CAP = 256 * E;
   __shared__ unsigned long long s_keys[CAP];
   __shared__ uint32_t s_wcnt[4][256];
   __shared__ uint32_t s_hist[256];
   __shared__ uint32_t s_part[4];
   const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
   const int64_t b = seg_first[blockIdx.x], e = seg_first[blockIdx.x + 1];
   const int len = (int)min(e - b, (int64_t)CAP + 1);
   if (len <= len_above) return;
   if (e - b > CAP) {
      if (longer_too) {
         block_radix_run(keys, alt, b, e, lo_bit, shift, s_hist, s_wcnt, s_part);
         __threadfence_block();
         __syncthreads();
         segment_finish(keys + b, (int)(e - b), b, (int64_t)blockIdx.x, lo_bit, out, tid);
         if (out.packed) {
            __threadfence_block();
            __syncthreads();
            segment_write_packed(keys + b, (int)(e - b), b, out.read_shift, out.packed, tid);
         }
      }
      return;
   }
   if (len < 2) {
      if (tid == 0 && len == 1) s_keys[0] = keys[b];
      __syncthreads();
      segment_finish(s_keys, len, b, (int64_t)blockIdx.x, 4, out, tid);
      if (out.packed && len == 1) segment_write_packed(s_keys, 1, b, out.read_shift, out.packed, tid);
      return;
   }
   const int rounds = (len + 255) >> 8;                      // per wavefront; its keys are [wid * rounds * 64, (wid + 1) * rounds * 64) of the segment
   const int first = wid * rounds * 64 + lane;
   const unsigned long long lt = (1ull << lane) - 1ull;
   unsigned long long k[E];
   uint32_t rank[E];
#pragma unroll
   for (int r = 0; r < E; r++) {
      const int i = first + r * 64;
      k[r] = (r < rounds && i < len) ? keys[b + i] : 0ull;
   }
   for (int bit = lo_bit; bit < shift; bit += 8) {
      const int nb = min(8, shift - bit);
      const uint32_t mask = (1u << nb) - 1u;
      for (int d = lane; d < 256; d += 64) s_wcnt[wid][d] = 0;
      wave_sync();
#pragma unroll
      for (int r = 0; r < E; r++) {
         if (r < rounds) {
            const bool valid = first + r * 64 < len;
            const uint32_t d = (uint32_t)(k[r] >> bit) & mask;
            unsigned long long same = __ballot(valid);
            for (int q = 0; q < nb; q++) {
               const unsigned long long bb = __ballot((d >> q) & 1u);
               same &= ((d >> q) & 1u) ? bb : ~bb;
            }
            const uint32_t below = (uint32_t)__popcll(same & lt);
            uint32_t pre = 0;
            if (valid) pre = s_wcnt[wid][d];
            rank[r] = pre + below;
            wave_sync();
            if (valid && below == 0) s_wcnt[wid][d] = pre + (uint32_t)__popcll(same);
            wave_sync();
         }
      }
      __syncthreads();
      {  // digit tid: its keys in the four wavefronts -> where each wavefront's keys of the digit begin
         const uint32_t c0 = s_wcnt[0][tid], c1 = s_wcnt[1][tid], c2 = s_wcnt[2][tid], c3 = s_wcnt[3][tid];
         const uint32_t v = c0 + c1 + c2 + c3;
         uint32_t incl = v;
         for (int o = 1; o < 64; o <<= 1) { const uint32_t x = (uint32_t)__shfl_up((int)incl, o); if (lane >= o) incl += x; }
         if (lane == 63) s_part[wid] = incl;
         __syncthreads();
         uint32_t before = 0;
         for (int w = 0; w < wid; w++) before += s_part[w];
         const uint32_t ex = before + incl - v;
         s_wcnt[0][tid] = ex; s_wcnt[1][tid] = ex + c0; s_wcnt[2][tid] = ex + c0 + c1; s_wcnt[3][tid] = ex + c0 + c1 + c2;
      }
      __syncthreads();
#pragma unroll
      for (int r = 0; r < E; r++)
         if (r < rounds && first + r * 64 < len) {
            const uint32_t d = (uint32_t)(k[r] >> bit) & mask;
            s_keys[s_wcnt[wid][d] + rank[r]] = k[r];
         }
      __syncthreads();
      if (bit + 8 < shift) {
#pragma unroll
         for (int r = 0; r < E; r++) {
            const int i = first + r * 64;
            if (r < rounds && i < len) k[r] = s_keys[i];
         }
      }
   }
   segment_finish(s_keys, len, b, (int64_t)blockIdx.x, lo_bit, out, tid);
   __syncthreads();
   if (out.packed) segment_write_packed(s_keys, len, b, out.read_shift, out.packed, tid);
   else for (int i = tid; i < len; i += 256) keys[b + i] = s_keys[i];
}

// Packed results: of every key the 40 bits below the read's (text position << 4 | score), four keys = five words per
// thread; the offsets as 32-bit numbers and the status as a byte.
__global__ void __launch_bounds__(256) k_pack_results(const unsigned long long *keys, int64_t n_locs, int low_bits, const int64_t *read_first,
                                                      const int64_t *status, int64_t n_reads, uint32_t *packed, uint32_t *first32, uint8_t *status8)
{
   const int64_t t = blockIdx.x * (int64_t)256 + threadIdx.x;
   if (keys != nullptr && 4 * t < n_locs) {                        // (nullptr: the ordering's last kernel has written the hits)
      const unsigned long long m = (1ull << low_bits) - 1ull;
      unsigned long long v[4];
#pragma unroll
      for (int j = 0; j < 4; j++) v[j] = 4 * t + j < n_locs ? keys[4 * t + j] & m : 0ull;
      uint32_t *o = packed + 5 * t;
      o[0] = (uint32_t)v[0];
      o[1] = (uint32_t)(v[0] >> 32) | (uint32_t)(v[1] << 8);
      o[2] = (uint32_t)(v[1] >> 24) | (uint32_t)(v[2] << 16);
      o[3] = (uint32_t)(v[2] >> 16) | (uint32_t)(v[3] << 24);
      o[4] = (uint32_t)(v[3] >> 8);
   }
   if (t <= n_reads) first32[t] = (uint32_t)read_first[t];
   if (t < n_reads) status8[t] = status[t] != 0 ? 1 : 0;
}

// Keeps one wavefront busy for `cycles` of the shader clock (gm_ctx_pipe_streams: do two streams run side by side?).
__global__ void k_spin(long long cycles)
{
   const long long t0 = clock64();
   while (clock64() - t0 < cycles) __builtin_amdgcn_s_sleep(8);
}

// Compact results: the sorted keys are the output; only the CSR offsets of the reads are derived from them
// (as k_unpack_locs does: the thread of the first key of a read writes that read's offset and those of the
// reads without hits before it).
// run_shift > 0: the sort left out the bits below it, and this kernel orders the runs as k_order_runs does (the read's
// bits, which the offsets are made of, lie above and stay where they are).
__global__ void k_read_first(unsigned long long *keys, int64_t n_locs, int pos_bits, int64_t n_reads, int64_t *read_first, int run_shift)
{
   int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
   if (t >= n_locs) return;
   if (run_shift > 0) order_run_from(keys, n_locs, run_shift, t);
   const int64_t read = (int64_t)(keys[t] >> (pos_bits + 4));
   const int64_t before = t == 0 ? -1 : (int64_t)(keys[t - 1] >> (pos_bits + 4));
   for (int64_t r = before + 1; r <= read; r++) read_first[r] = t;
   if (t == n_locs - 1)
      for (int64_t r = read + 1; r <= n_reads; r++) read_first[r] = n_locs;
}

}  // namespace gm
