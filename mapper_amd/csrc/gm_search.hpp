// gm_search.hpp -- k_search: the block search of a batch of reads, one search node per lane.
//
// Replaces blocksearch_trail_rec + seqsearch_fw/bw + seqdash_fw/bw (src/blocksearch.c:106-496) with
// bwt_query_all / get_occ_all (src/index_bwt.c:198-304, :840-905) inlined.
//
//  * A node = one pending call of seqsearch_* / one iteration of seqdash_*: bi-interval, next query
//    position, packed phase bookkeeping (gm_core.hpp).  A wave advances 64 nodes per pass of its loop.
//  * fetch: the 64-byte Occ block of a node is read by the node's quad of lanes, one 16-byte load per
//    lane (four load instructions per rank, each touching 16 lines per wave).  A lane reading its own
//    line with four loads runs 3x slower on footprints of several GiB
//    (profiles/r1_gather_microbench.jsonl, shape lane64 against quad64).
//  * hand-over: the four sub-blocks reach the owning lane through a per-wave LDS staging area
//    (80-byte slots: conflict-free 16-byte writes and reads).
//  * rank, bwt_query_all and the child rules run per lane.  A pass follows ONE child per node: the
//    first valid candidate stays in the lane's registers; if more remain, the node itself goes back to
//    the wave's LDS deque as a continuation (meta.first = next candidate) and is re-extended when
//    popped.  Siblings are rare (repeats, query N); a loop over children would cost every lane of the
//    wave a pass per sibling of any lane.
//  * The recursion of the reference (seqsearch_* calling itself, pathstack_t per tree level) is that
//    deque: pushes compacted with __ballot + popcount prefix, pops hand the newest nodes to idle lanes
//    (LIFO keeps the frontier small); its oldest part spills to a per-wave global region when full.
//  * Finished nodes go straight to a piece of the global list reserved by the wave; phase records (what a node becomes at the end
//    of a phase) and the superblock counts sit in LDS.
//  * History of the design (a node per quad of lanes, a loop over children, one atomic per emitting
//    pass: 13 G extension steps/s; this kernel: 44-65 G) is in DESIGN.md section 3.
#pragma once
#include "gm_kernels.hpp"

namespace gm {

constexpr int STAGE_SLOT_U4 = 5;     // staging slot per lane, in 16-byte units (4 used + 1 pad)
// Finished nodes go straight into a piece of the global list that the wave has reserved for itself:
// HIT_CHUNK entries per atomic on the list's counter.  Every atomic on that one address costs all
// waves about 10 ns (one per emitting pass was the whole kernel time once; one per 64 hits still a
// third of it).  A piece whose rest is too short for the nodes of a pass, and the last piece of a wave,
// are filled up with empty entries (sz = 0): the list has holes, its consumers skip them.
constexpr uint32_t HIT_CHUNK = 512;
struct HitOut {
   unsigned long long out_base;      // the reserved piece: next free entry, entries left
   uint32_t out_left;
   unsigned long long n_real;        // entries written that are not holes
};

__device__ __forceinline__ void hit_holes(const SearchArgs &A, const HitOut &hr, int lane)
{
   for (uint32_t i = lane; i < hr.out_left; i += 64)
      if (hr.out_base + i < A.hit_cap) {
         uint4 *d = reinterpret_cast<uint4 *>(A.hits + hr.out_base + i);
         d[0] = make_uint4(0, 0, 0, 0);
         d[1] = make_uint4(0, 0, 0, 0);
      }
}

// End of the wave: the rest of the reserved piece becomes holes.
__device__ __forceinline__ void hit_finish(const SearchArgs &A, HitOut &hr, int lane)
{
   hit_holes(A, hr, lane);
   if (lane == 0 && hr.n_real) atomicAdd(A.stats + 3, hr.n_real);
}

// Converged call: lanes with `flag` append `nd`.
__device__ __forceinline__ void hit_emit(const SearchArgs &A, HitOut &hr, bool flag, const Node &nd, int lane)
{
   const unsigned long long m = __ballot(flag);
   if (m == 0) return;
   const uint32_t add = (uint32_t)__popcll(m);
   if (hr.out_left < add) {
      hit_holes(A, hr, lane);
      unsigned long long base = 0;
      if (lane == 0) base = atomicAdd(A.hit_count, (unsigned long long)HIT_CHUNK);
      hr.out_base = __shfl(base, 0);
      hr.out_left = HIT_CHUNK;
   }
   const unsigned long long at = hr.out_base + (unsigned long long)__popcll(m & ((1ull << lane) - 1ull));
   if (flag && at < A.hit_cap) {          // beyond the list: counted, not stored (the host runs again with a longer list)
      uint4 *d = reinterpret_cast<uint4 *>(A.hits + at);
      d[0] = make_uint4(nd.w[0], nd.w[1], nd.w[2], nd.w[3]);
      d[1] = make_uint4(nd.w[4], nd.w[5], nd.w[6], nd.w[7]);
   }
   hr.out_base += add; hr.out_left -= add; hr.n_real += add;
}

// Lanes whose quad's lane I is set in `m`: the execution mask of fetch slot I (scalar work only).
template <int I>
__device__ __forceinline__ unsigned long long quad_spread(unsigned long long m)
{
   unsigned long long t = (m >> I) & 0x1111111111111111ull;
   t |= t << 1;
   return t | (t << 2);
}

// Fetch slot I: the quad reads the block of its lane I, 16 bytes per lane.  `off` is the byte offset
// of the lane's own block; Occ tables are below 4 GiB (n < 2^33), so a 32-bit offset from the
// scalar base addresses them.
template <int I>
__device__ __forceinline__ void quad_fetch(const char *occ, uint32_t off, unsigned long long want, uint32_t sub16, uint4 &v)
{
   uint32_t o = (uint32_t)__builtin_amdgcn_mov_dpp((int)off, I * 0x55, 0xF, 0xF, true) + sub16;
   if (__builtin_amdgcn_inverse_ballot_w64(quad_spread<I>(want)))
      v = *reinterpret_cast<const uint4 *>(occ + o);
}

template <int I>
__device__ __forceinline__ void quad_stage(uint4 *slot /* stage + quad*4*SLOT + sub */, unsigned long long want, const uint4 &v)
{
   if (__builtin_amdgcn_inverse_ballot_w64(quad_spread<I>(want))) slot[I * STAGE_SLOT_U4] = v;
}

// Rank inside one block held by one lane: counts (relative to the superblock) of the five symbols in
// BWT[superblock start, block start + o), o = 0..127.  Same result as block_rank (gm_core.hpp).
__device__ __forceinline__ void lane_rank(const uint4 &s0, const uint4 &s1, const uint4 &s2, const uint4 &s3, int o,
                                          uint32_t rel[SYMS])
{
   const unsigned long long t = (1ull << (o & 63)) - 1ull;
   const unsigned long long lo = o >= 64 ? ~0ull : t, hi = o >= 64 ? t : 0ull;
   const uint32_t m0 = (uint32_t)lo, m1 = (uint32_t)(lo >> 32), m2 = (uint32_t)hi, m3 = (uint32_t)(hi >> 32);
   uint32_t a = s0.w & CNT_MASK, c = s1.w & CNT_MASK, g = s2.w & CNT_MASK, tt = s3.w & CNT_MASK;
   uint32_t n = (s0.w >> 25) | ((s1.w >> 25) << 7) | ((s2.w >> 25) << 14) | ((s3.w >> 25) << 21);
   // v_bitop3_b32: result bit = imm[(a << 2) | (b << 1) | c]; one instruction per symbol mask, one
   // v_bcnt_u32_b32 (popcount + add) per count.  S.x, S.y, S.z = planes b0, b1, b2 of 32 positions.
#define GM_SUB(S, M) { \
      const uint32_t x = __builtin_amdgcn_bitop3_b32(M, S.z, S.z, 0x10);            /* M & ~b2: A C G T */ \
      a  += __popc(__builtin_amdgcn_bitop3_b32(x, S.y, S.x, 0x10));                 /* x & ~b1 & ~b0 */ \
      c  += __popc(__builtin_amdgcn_bitop3_b32(x, S.y, S.x, 0x20));                 /* x & ~b1 &  b0 */ \
      g  += __popc(__builtin_amdgcn_bitop3_b32(x, S.y, S.x, 0x40));                 /* x &  b1 & ~b0 */ \
      tt += __popc(__builtin_amdgcn_bitop3_b32(x, S.y, S.x, 0x80));                 /* x &  b1 &  b0 */ \
      n  += __popc(__builtin_amdgcn_bitop3_b32(S.z, S.x, M, 0x20)); }               /* b2 & ~b0 & M  */
   GM_SUB(s0, m0) GM_SUB(s1, m1) GM_SUB(s2, m2) GM_SUB(s3, m3)
#undef GM_SUB
   rel[0] = a; rel[1] = c; rel[2] = g; rel[3] = tt; rel[4] = n;
}

// Keeps the compiler from carrying values derived from a block across the point where some lanes
// replace the block (it would merge the two versions with a register move per derived value).
__device__ __forceinline__ void pin(uint4 &v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }
// Registers that the lanes which load nothing never look at: "defined" for the compiler without an instruction (48 moves
// of zero per pass were 8 % of the kernel's vector instructions; plainly uninitialised they cost 90 registers).
__device__ __forceinline__ void any_value(uint4 &v) { asm volatile("" : "=v"(v.x), "=v"(v.y), "=v"(v.z), "=v"(v.w)); }

constexpr int PH_LDS_ROWS = 8;       // phase-record rows (read lengths) kept in LDS; others are read from global memory

// SC = the search of blocksc_trail (src/blocksearch.c:7-103): the schedule of a query is the block search of its
// left part followed by one bounded forward extension (build_sc_phase_table); a hit of the left part only goes on if
// it starts below the bound of the query's own left block (:76-85); inside the extension a node on the boundary
// follows the symbols up to the query's only (scsearch_fw, :338) and stays on it through the query's symbol (:366).
template <bool SC>
__global__ void __launch_bounds__(WG_THREADS) k_search(SearchArgs A)
{
   __shared__ Node s_q[WG_WAVES][QCAP];
   __shared__ uint4 s_stage[WG_WAVES][64 * STAGE_SLOT_U4];
   __shared__ uint2 s_ph[PH_LDS_ROWS * MAX_PHASES];
   extern __shared__ unsigned long long s_cb[];              // C[j] + count of j before superblock s

   const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, sub = lane & 3;
   const uint32_t gw = blockIdx.x * WG_WAVES + wid;
   for (int i = tid; i < (int)A.ix.n_super * SYMS; i += WG_THREADS)
      s_cb[i] = A.ix.tables[TAB_SBASE + i] + A.ix.tables[TAB_C + i % SYMS];
   for (int i = tid; i < min(A.n_rows, PH_LDS_ROWS) * MAX_PHASES; i += WG_THREADS) s_ph[i] = A.phase_rec[i];
   __syncthreads();

   WaveQueue wq;
   wq.q = s_q[wid];
   wq.spill = A.spill + (size_t)gw * A.spill_cap;
   wq.bot = wq.top = wq.spill_top = 0;
   wq.n_spilled = 0;
   wq.spill_cap = A.spill_cap;
   wq.error = A.error;
   uint4 *stage = s_stage[wid];
   uint4 *stage_w = stage + (lane & ~3) * STAGE_SLOT_U4 + sub;      // where this lane writes slot 0 of its quad
   const uint4 *stage_r = stage + lane * STAGE_SLOT_U4;             // this lane's own block
   HitOut hr;
   hr.out_base = 0; hr.out_left = 0; hr.n_real = 0;
   unsigned long long n_iter = 0, n_cont = 0;
   unsigned long long res_base = 0;      // reads reserved from the global cursor (A.cursor_chunk per atomic)
   uint32_t res_left = 0;

   const int seeds_per = A.tau + 1;
   const int reads_per_refill = 64 / seeds_per;
   const unsigned long long lt_mask = (1ull << lane) - 1ull;
   const char *occ = reinterpret_cast<const char *>(A.ix.occ);
   const uint32_t sub16 = (uint32_t)sub << 4;

   bool reads_left = true;
   bool have = false;
   // the node of this lane
   int64_t fp = 0, rp = 0, sz = 0;
   uint64_t mm = 0;
   uint32_t read = 0, meta = 0;
   int pos = 0;
   uint4 qword = make_uint4(0, 0, 0, 0);   // the query symbols around pos (one load per 16 letters of a walk, SC, or per 32, the planes:
   int qword_at = -1;                      // a load per four letters was a fifth of the kernel's requests to the L2); pos >> 4 (>> 5) of them, -1 = none
   unsigned long long ext_count = 0;
   uint32_t jump_steps = 0;     // extension steps of this lane's seeds that the jump table stood in for
#ifdef GM_STEP_HIST
   unsigned long long h_cnt[6] = {0, 0, 0, 0, 0, 0};
#endif

   for (;;) {
      // ------------------------------------------------------------ hand nodes to idle lanes
      unsigned long long needm = __ballot(!have);
      if (needm) {
         int nneed = __popcll(needm);
         int count = (int)(wq.top - wq.bot);
         if (count < nneed) {
            if (wq.spill_top > 0) {
               wq.spill_top -= SPILL_CHUNK;
               for (int i = lane; i < SPILL_CHUNK; i += 64) {
                  const uint4 *sp = reinterpret_cast<const uint4 *>(wq.spill + wq.spill_top + i);
                  uint4 x = sp[0], y = sp[1];
                  Node t;
                  t.w[0] = x.x; t.w[1] = x.y; t.w[2] = x.z; t.w[3] = x.w;
                  t.w[4] = y.x; t.w[5] = y.y; t.w[6] = y.z; t.w[7] = y.w;
                  lds_store_node<QCAP>(wq.q, wq.top + i, t);
               }
               wq.top += SPILL_CHUNK;
            } else if (reads_left) {
               // Admit reads: one seed per leaf block of the schedule (src/blocksearch.c:138-144).
               if (res_left == 0) {
                  unsigned long long got_base = 0;
                  if (lane == 0) got_base = atomicAdd(A.cursor, (unsigned long long)A.cursor_chunk);
                  res_base = __shfl(got_base, 0);
                  res_left = res_base < (unsigned long long)A.n_reads
                                ? (uint32_t)min((unsigned long long)A.cursor_chunk, (unsigned long long)A.n_reads - res_base) : 0u;
               }
               if (res_left == 0) {
                  reads_left = false;
               } else {
                  const int got = (int)min((uint32_t)reads_per_refill, res_left);
                  const unsigned long long base = res_base;
                  res_base += (unsigned long long)got; res_left -= (uint32_t)got;
                  int r = lane / seeds_per, leaf = lane - r * seeds_per;
                  bool mk = r < got, seed_cand = false;
                  Node packed;
                  for (int k = 0; k < 8; k++) packed.w[k] = 0;
                  if (mk) {
                     uint32_t rd = (uint32_t)(base + r);
                     int m = A.read_len[rd];
                     int row = SC ? A.read_row[rd] : (m <= MAX_READ ? A.len_row[m] : NO_ROW);
                     mk = row != NO_ROW;
                     if (SC && mk) mk = leaf < (int)A.row_leaves[row];          // tau_l + 1 leaves, by the query's class and N count
                     // (gpumap_ann_build with the reference's reuse of searches: the leaves this query starts from, and the
                     // first phase whose steps the reference executes for it; kept in byte 6 of the mismatch list, tau <= 5 there)
                     uint32_t count_from = 0;
                     if (SC && mk && A.sc_flags != nullptr) {
                        const uint32_t f = A.sc_flags[rd];
                        mk = (f >> leaf) & 1;
                        count_from = f >> 2;
                     }
                     if (mk) {
                        NodeR sd;
                        sd.fp = 0; sd.rp = 0; sd.sz = A.ix.n;        // bwt_new_query, src/index_bwt.c:48-52
                        sd.read = rd; sd.score = 0; sd.mm = ((uint64_t)row << 56) | ((uint64_t)count_from << 48);
                        sd.sref = 0; sd.pos = sd.end = sd.tau = sd.diff = sd.phase = 0; sd.dash = sd.fw = false;
                        enter_phase(sd, leaf, A.phases[row * MAX_PHASES + leaf]);
                        // The first K steps of a leaf are an exact search from the root interval: one look-up in
                        // the jump table instead (counted as the steps the reference executes).  Not for a query N
                        // among the K symbols (it follows all symbols), nor when the block has no letter left after.
                        const uint4 *jt = nullptr;
                        int jk = 0;
                        if (A.jump != nullptr && sd.pos - sd.end >= A.jump_k) { jt = A.jump; jk = A.jump_k; }
                        else if (A.jump2 != nullptr && sd.pos - sd.end >= A.jump2_k) { jt = A.jump2; jk = A.jump2_k; }
                        if (jt != nullptr) {
                           uint32_t idx = 0, bad = 0;
                           if constexpr (SC) {
                              const uint8_t *qp = A.reads + (size_t)rd * A.stride + sd.pos;
                              for (int t = 0; t < jk; t++) {
                                 const uint32_t sym = qp[-t];
                                 bad |= sym & ~3u;
                                 idx |= (sym & 3u) << (2 * t);
                              }
                           } else {
                              // the jk letters that end at sd.pos out of the row's bit planes (two runs of 12 bytes hold them):
                              // bit t of plane p, reversed, is bit p of the letter at sd.pos - t
                              const int lo = sd.pos - jk + 1;
                              const int wb = (sd.pos >> 5) > 0 ? (sd.pos >> 5) - 1 : 0;
                              const uint32_t *qp = A.planes + (size_t)rd * (size_t)(3 * (A.stride >> 5)) + 3 * wb;
                              const uint32_t sh = (uint32_t)(lo - 32 * wb), keep = (1u << jk) - 1u;     // (jk <= 16)
                              uint32_t x[3];
#pragma unroll
                              for (int pl = 0; pl < 3; pl++)
                                 x[pl] = (uint32_t)((((uint64_t)qp[3 + pl] << 32) | (uint64_t)qp[pl]) >> sh) & keep;
                              bad = x[2];                                      // N, an unknown letter
                              auto spread = [](uint32_t v) {                   // bit t -> bit 2t
                                 v = (v | (v << 8)) & 0x00ff00ffu; v = (v | (v << 4)) & 0x0f0f0f0fu;
                                 v = (v | (v << 2)) & 0x33333333u; return (v | (v << 1)) & 0x55555555u;
                              };
                              idx = spread(__brev(x[0]) >> (32 - jk)) | (spread(__brev(x[1]) >> (32 - jk)) << 1);
                           }
                           if (!bad) {
                              const uint4 e = jt[idx];
                              sd.fp = (int64_t)e.x | ((int64_t)(e.w & 0xff) << 32);
                              sd.rp = (int64_t)e.y | ((int64_t)((e.w >> 8) & 0xff) << 32);
                              sd.sz = (int64_t)e.z | ((int64_t)((e.w >> 16) & 0xff) << 32);
                              if (!SC || (uint32_t)leaf >= count_from) jump_steps += e.w >> 24;
                              sd.pos -= jk;
                              mk = sd.sz > 0;
                              // one suffix left at the table's entry: a candidate for k_verify as it stands (the pass that took it
                              // there one letter later cost a line of the Occ table per seed of a unique place)
                              seed_cand = !SC && A.seed_cand && sd.sz == 1;
                           }
                        }
                        pack_node(sd, packed);
                        if (seed_cand) packed.w[5] |= 1u << META_CAND;
                     }
                  }
                  wave_push<QCAP, SPILL_CHUNK>(wq, mk && !seed_cand, packed, lane);
                  if (!SC && A.seed_cand) hit_emit(A, hr, mk && seed_cand, packed, lane);
               }
            }
            wave_sync();
            count = (int)(wq.top - wq.bot);
         }
         if (!have) {
            int rank = __popcll(needm & lt_mask);
            if (rank < count) {
               Node nd;
               lds_load_node<QCAP>(wq.q, wq.top - 1 - (uint32_t)rank, nd);
               fp = (int64_t)nd.w[0] | ((int64_t)(nd.w[3] & 0xff) << 32);
               rp = (int64_t)nd.w[1] | ((int64_t)((nd.w[3] >> 8) & 0xff) << 32);
               sz = (int64_t)nd.w[2] | ((int64_t)((nd.w[3] >> 16) & 0xff) << 32);
               pos = (int)(nd.w[3] >> 24);
               read = nd.w[4]; meta = nd.w[5];
               mm = (uint64_t)nd.w[6] | ((uint64_t)nd.w[7] << 32);
               have = true;
               qword_at = -1;
            }
         }
         wq.top -= (uint32_t)min(nneed, count);
         wave_sync();
      }
      const unsigned long long havem = __ballot(have);
      if (!havem) {
         if (!reads_left && wq.spill_top == 0 && wq.top == wq.bot) break;
         continue;
      }
#ifdef GM_STEP_HIST
      {  // diagnostic build: extension steps by the size of the interval they extend
         const bool fresh = have && ((meta >> META_FIRST) & 7) == 0;
         h_cnt[0] += __popcll(__ballot(fresh && sz == 2));
         h_cnt[1] += __popcll(__ballot(fresh && sz >= 3 && sz <= 4));
         h_cnt[2] += __popcll(__ballot(fresh && sz >= 5 && sz <= 16));
         h_cnt[3] += __popcll(__ballot(fresh && sz >= 17 && sz <= 256));
         h_cnt[4] += __popcll(__ballot(fresh && sz > 256));
         h_cnt[5] += __popcll(__ballot(fresh && sz == 1));
      }
#endif
      const int first = (int)(meta >> META_FIRST) & 7;
      ext_count += (unsigned long long)__popcll(__ballot(have && first == 0 && (!SC || (meta & 63u) >= (uint32_t)((mm >> 48) & 0xff))));   // a continuation was counted at its first visit
      n_iter++;

      // ------------------------------------------------------------ fetch: all loads first, then the hand-over
      const bool fw = (meta >> META_FW) & 1;
      const int64_t a = fw ? rp : fp, other = fw ? fp : rp;
      const int64_t e0 = a, e1 = a + sz;                           // ranks at a-1 and a+sz-1, exclusive form
      const uint32_t b0 = (uint32_t)(e0 >> BLOCK_POS_LOG), b1 = (uint32_t)(e1 >> BLOCK_POS_LOG);
      const unsigned long long need1m = __ballot(have && b1 != b0);
      uint4 f00, f01, f02, f03, f10, f11, f12, f13;
      any_value(f00); any_value(f01); any_value(f02); any_value(f03); any_value(f10); any_value(f11); any_value(f12); any_value(f13);
      quad_fetch<0>(occ, b0 << 6, havem, sub16, f00); quad_fetch<1>(occ, b0 << 6, havem, sub16, f01);
      quad_fetch<2>(occ, b0 << 6, havem, sub16, f02); quad_fetch<3>(occ, b0 << 6, havem, sub16, f03);
      if (need1m) {
         quad_fetch<0>(occ, b1 << 6, need1m, sub16, f10); quad_fetch<1>(occ, b1 << 6, need1m, sub16, f11);
         quad_fetch<2>(occ, b1 << 6, need1m, sub16, f12); quad_fetch<3>(occ, b1 << 6, need1m, sub16, f13);
      }
      if constexpr (SC) {
         if (have && (pos >> 4) != qword_at) {
            qword_at = pos >> 4;
            qword = *reinterpret_cast<const uint4 *>(A.reads + (size_t)read * A.stride + (pos & ~15));    // (rows are multiples of 128 bytes)
         }
      } else if (have && (pos >> 5) != qword_at) {
         // the mapping search reads the rows' bit planes (12 bytes per 32 letters; the byte matrix is not written for it)
         qword_at = pos >> 5;
         const uint32_t *qp = A.planes + (size_t)read * (size_t)(3 * (A.stride >> 5)) + 3 * (pos >> 5);
         qword.x = qp[0]; qword.y = qp[1]; qword.z = qp[2];
      }

      uint32_t rel0[SYMS], rel1[SYMS];
      uint4 g0, g1, g2, g3;
      any_value(g0); any_value(g1); any_value(g2); any_value(g3);
      quad_stage<0>(stage_w, havem, f00); quad_stage<1>(stage_w, havem, f01);
      quad_stage<2>(stage_w, havem, f02); quad_stage<3>(stage_w, havem, f03);
      wave_sync();
      if (have) { g0 = stage_r[0]; g1 = stage_r[1]; g2 = stage_r[2]; g3 = stage_r[3]; }
      lane_rank(g0, g1, g2, g3, (int)(e0 & (BLOCK_POS - 1)), rel0);
      wave_sync();
      quad_stage<0>(stage_w, need1m, f10); quad_stage<1>(stage_w, need1m, f11);
      quad_stage<2>(stage_w, need1m, f12); quad_stage<3>(stage_w, need1m, f13);
      wave_sync();
      if (have && b1 != b0) { g0 = stage_r[0]; g1 = stage_r[1]; g2 = stage_r[2]; g3 = stage_r[3]; }
      pin(g0); pin(g1); pin(g2); pin(g3);
      lane_rank(g0, g1, g2, g3, (int)(e1 & (BLOCK_POS - 1)), rel1);
      wave_sync();
      int qc;
      if constexpr (SC) {
         const uint32_t qsel = (pos & 8) ? ((pos & 4) ? qword.w : qword.z) : ((pos & 4) ? qword.y : qword.x);
         qc = (int)((qsel >> (8 * (pos & 3))) & 0xff);
      } else {
         const uint32_t qb = (uint32_t)pos & 31u;
         qc = (int)(((qword.x >> qb) & 1u) | (((qword.y >> qb) & 1u) << 1) | (((qword.z >> qb) & 1u) << 2));
      }

      // ------------------------------------------------------------ bwt_query_all, src/index_bwt.c:262-273
      const int sb0 = (int)(e0 >> SUPER_LOG), sb1 = (int)(e1 >> SUPER_LOG);
      int64_t szn[SYMS];
      for (int j = 0; j < SYMS; j++) szn[j] = (int64_t)rel1[j] - (int64_t)rel0[j];
      if (have && sb1 != sb0)
         for (int j = 0; j < SYMS; j++) szn[j] += (int64_t)(s_cb[sb1 * SYMS + j] - s_cb[sb0 * SYMS + j]);
      const int64_t tot = szn[0] + szn[1] + szn[2] + szn[3] + szn[4];

      // non-empty children by internal symbol, then by query symbol (forward: complemented)
      uint32_t alive = (szn[0] >= 1 ? 1u : 0u) | (szn[1] >= 1 ? 2u : 0u) | (szn[2] >= 1 ? 4u : 0u) |
                       (szn[3] >= 1 ? 8u : 0u) | (szn[4] >= 1 ? 16u : 0u);
      if (fw) alive = (__brev(alive & 15u) >> 28) | (alive & 16u);
      uint32_t cand = have ? valid_children(SC ? meta & ~(1u << META_BOUND) : meta, pos, qc, alive) & (0x1fu << first) : 0u;
      if (SC && ((meta >> META_BOUND) & 1) && !((meta >> META_DASH) & 1)) cand &= (2u << qc) - 1u;      // nt <= query[pos] (:338)

      // ------------------------------------------------------------ this pass follows the first candidate;
      // the others stay behind as a continuation of the same node (re-extended when it is popped)
      const bool cont = (cand & (cand - 1)) != 0;
      if (__any(cont)) {
         Node out;
         out.w[0] = (uint32_t)fp; out.w[1] = (uint32_t)rp; out.w[2] = (uint32_t)sz;
         out.w[3] = (uint32_t)((fp >> 32) & 0xff) | ((uint32_t)((rp >> 32) & 0xff) << 8) |
                    ((uint32_t)((sz >> 32) & 0xff) << 16) | ((uint32_t)pos << 24);
         out.w[4] = read;
         const uint32_t rest = cand & (cand - 1);
         out.w[5] = (meta & ~(7u << META_FIRST)) | ((uint32_t)(__ffs((int)rest) - 1) << META_FIRST);
         out.w[6] = (uint32_t)mm; out.w[7] = (uint32_t)(mm >> 32);
         wave_push<QCAP, SPILL_CHUNK>(wq, cont, out, lane);
         n_cont += (unsigned long long)__popcll(__ballot(cont));
      }
      bool emit = false, stable = false;
      if (cand != 0) {
         const int nt = __ffs((int)cand) - 1;
         const int js = fw ? complement(nt) : nt;
         // the pointer on the side that was not extended accumulates in the order T,G,C,A,N (:270-273)
         const int64_t r3 = other + (sz - tot), r2 = r3 + szn[3], r1 = r2 + szn[2], r0 = r1 + szn[1], r4 = r0 + szn[0];
         const int64_t fpn = (int64_t)(s_cb[sb0 * SYMS + js] + sel5(rel0, js));
         const int64_t rpn = sel5v(r0, r1, r2, r3, r4, js);
         const int64_t sz_parent = sz;
         sz = sel5(szn, js);
         stable = sz == sz_parent;
         fp = fw ? rpn : fpn;
         rp = fw ? fpn : rpn;
         uint32_t meta_n;
         const bool done = child_meta(meta, pos, qc, nt, mm, meta_n, mm);
         meta = meta_n;
         if (SC && nt != qc) meta &= ~(1u << META_BOUND);                        // off the boundary (:366)
         pos = fw ? pos + 1 : pos - 1;
         if (done) {
            const int idx = (int)(mm >> 56) * MAX_PHASES + (int)(meta & 63);
            const uint2 rec = idx < PH_LDS_ROWS * MAX_PHASES ? s_ph[idx] : A.phase_rec[idx];
            if ((rec.y >> 8) & 1) {
               emit = true;
               pos = 0;
               if (SC) meta &= ~(1u << META_BOUND);
            } else {
               const uint32_t s = (meta >> META_SCORE) & 7;
               meta = rec.x | (s << META_SCORE) | (s << META_SREF);
               pos = (int)(rec.y & 255);
               if (SC && ((rec.y >> 9) & 1)) {
                  // a hit of the left block enters the bounded extension: those at or beyond the end of the query's
                  // own left block are dropped (:76-85), the others start on the boundary (:91)
                  if (fp >= A.sc_bound[read]) have = false;
                  meta |= 1u << META_BOUND;
                  // (a hit that already has more mismatches than this query allows, possible only where the left block was
                  // searched with another query's budget: scsearch_fw extends it once and follows no child, :345-346;
                  // without the dash flag no child is valid for it)
                  if (s > ((rec.x >> META_TAU) & 7u)) meta &= ~(1u << META_DASH);
               }
            }
         }
      } else {
         have = false;
      }
      // A node whose interval has shrunk to one suffix (A.verify = 2: to one or two) leaves the search: k_verify
      // finishes it against the text (a suffix-array entry and a window of the text per suffix instead of
      // Occ lines per letter).
      // (three and four suffixes only once the interval has survived a step unchanged: what a coincidence of the
      // seed's letters adds to an interval is gone after a letter or two and is cheaper to shed on the Occ table
      // than to replay; copies of a repeat stay)
      const bool toverify = A.verify && have && !emit && (sz <= 2 ? sz <= A.verify : (sz <= A.verify && (stable || !A.verify_stable)));
      if (__any(emit || toverify)) {
         Node out;
         out.w[0] = (uint32_t)fp; out.w[1] = (uint32_t)rp; out.w[2] = (uint32_t)sz;
         out.w[3] = (uint32_t)((fp >> 32) & 0xff) | ((uint32_t)((rp >> 32) & 0xff) << 8) |
                    ((uint32_t)((sz >> 32) & 0xff) << 16) | ((uint32_t)pos << 24);
         out.w[4] = read; out.w[5] = meta | ((uint32_t)toverify << META_CAND);
         out.w[6] = (uint32_t)mm; out.w[7] = (uint32_t)(mm >> 32);
         hit_emit(A, hr, emit || toverify, out, lane);
         if (emit || toverify) have = false;
      }
   }
   hit_finish(A, hr, lane);
   if (lane == 0 && wq.n_spilled) atomicAdd(A.stats + 4, (unsigned long long)wq.n_spilled);
#ifdef GM_STEP_HIST
   if (lane == 0) for (int k = 0; k < 6; k++) atomicAdd(A.stats + 5 + k, h_cnt[k]);
#endif
   for (int o = 32; o > 0; o >>= 1) jump_steps += (uint32_t)__shfl_xor((int)jump_steps, o);
   ext_count += jump_steps;
   if (lane == 0 && ext_count) {
      atomicAdd(A.n_ext, ext_count);
      atomicAdd(A.stats + 0, n_iter);
      atomicAdd(A.stats + 1, n_cont);
      atomicAdd(A.stats + 2, (unsigned long long)jump_steps);
   }
}

}  // namespace gm
