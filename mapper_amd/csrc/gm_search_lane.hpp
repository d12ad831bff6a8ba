// gm_search_lane.hpp -- k_search_lane: the block search with one search node per lane.
//
// Same schedule, node format, queue and output as k_search (gm_kernels.hpp); what differs is the
// mapping of nodes to lanes.  k_search gives a node to a quad of lanes, so a wave advances 16 nodes
// per pass of the loop and is bound by instruction issue (about 750 vector instructions per pass,
// profiles/r1_*).  Here a wave advances 64 nodes per pass:
//
//  * fetch: the 64-byte Occ block of a node is still read by its quad, one 16-byte load per lane
//    (four load instructions per rank, each touching 16 lines per wave).  A lane reading its own
//    line with four loads runs 3x slower on footprints of several GiB
//    (profiles/r1_gather_microbench.jsonl, shape lane64 against quad64).
//  * hand-over: the four sub-blocks reach the owning lane through a per-wave LDS staging area
//    (80-byte slots: conflict-free 16-byte writes and reads).
//  * rank, bwt_query_all and the child rules run per lane; the first child stays in the lane's
//    registers, further children go to the wave's LDS deque.
#pragma once
#include "gm_kernels.hpp"

namespace gm {

constexpr int LQCAP = 128;           // deque capacity per wave (nodes), power of two
constexpr int LCHUNK = 64;           // nodes moved per spill / reload
constexpr int STAGE_SLOT_U4 = 5;     // staging slot per lane, in 16-byte units (4 used + 1 pad)
constexpr int HITBUF = 128;          // finished nodes staged per wave before they go to the global list

// Finished nodes are staged in a per-wave LDS ring and leave it 64 at a time: one atomic on the
// global counter and one contiguous 2-KB store per 64 hits.  (One atomic per emitting pass, as in
// k_search, serialises all waves on one address: about 10 ns per atomic, the whole kernel time.)
struct HitRing {
   Node *ring;
   uint32_t head, tail;
};

__device__ __forceinline__ void hit_flush(const SearchArgs &A, HitRing &hr, int lane, uint32_t n)
{
   unsigned long long base = 0;
   if (lane == 0) base = atomicAdd(A.hit_count, (unsigned long long)n);
   base = __shfl(base, 0);
   if ((uint32_t)lane < n && base + lane < A.hit_cap) {
      Node t;
      lds_load_node<HITBUF>(hr.ring, hr.head + lane, t);
      uint4 *d = reinterpret_cast<uint4 *>(A.hits + base + lane);
      d[0] = make_uint4(t.w[0], t.w[1], t.w[2], t.w[3]);
      d[1] = make_uint4(t.w[4], t.w[5], t.w[6], t.w[7]);
   }
   hr.head += n;
   wave_sync();
}

// Converged call: lanes with `flag` append `nd`.
__device__ __forceinline__ void hit_emit(const SearchArgs &A, HitRing &hr, bool flag, const Node &nd, int lane)
{
   unsigned long long m = __ballot(flag);
   if (m == 0) return;
   if (flag) lds_store_node<HITBUF>(hr.ring, hr.tail + (uint32_t)__popcll(m & ((1ull << lane) - 1ull)), nd);
   hr.tail += (uint32_t)__popcll(m);
   wave_sync();
   if (hr.tail - hr.head >= 64) hit_flush(A, hr, lane, 64);
}

// Broadcast of lane i's value to its quad (DPP quad_perm [i,i,i,i]); every lane must be active.
template <int I>
__device__ __forceinline__ uint32_t quad_bcast(uint32_t x)
{
   return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, I * 0x55, 0xF, 0xF, true);
}

template <int I>
__device__ __forceinline__ void quad_fetch(const uint4 *occ4, uint32_t blk, bool want, int sub, uint4 &v)
{
   uint32_t b = quad_bcast<I>(blk);
   bool w = quad_bcast<I>((uint32_t)want) != 0;
   if (w) v = occ4[(size_t)b * 4 + sub];
}

template <int I>
__device__ __forceinline__ void quad_stage(uint4 *stage, bool want, int lane, int sub, const uint4 &v)
{
   bool w = quad_bcast<I>((uint32_t)want) != 0;
   if (w) stage[((lane & ~3) + I) * STAGE_SLOT_U4 + sub] = v;
}

__device__ __forceinline__ void stage_read(const uint4 *stage, int lane, uint32_t blk[BLOCK_DWORDS])
{
   const uint4 *s = stage + lane * STAGE_SLOT_U4;
#pragma unroll
   for (int k = 0; k < 4; k++) {
      uint4 t = s[k];
      blk[4 * k] = t.x; blk[4 * k + 1] = t.y; blk[4 * k + 2] = t.z; blk[4 * k + 3] = t.w;
   }
}

__global__ void __launch_bounds__(WG_THREADS) k_search_lane(SearchArgs A)
{
   __shared__ Node s_q[WG_WAVES][LQCAP];
   __shared__ uint4 s_stage[WG_WAVES][64 * STAGE_SLOT_U4];
   __shared__ Node s_hit[WG_WAVES][HITBUF];
   extern __shared__ unsigned long long s_cb[];              // C[j] + count of j before superblock s

   const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, sub = lane & 3;
   const uint32_t gw = blockIdx.x * WG_WAVES + wid;
   for (int i = tid; i < (int)A.ix.n_super * SYMS; i += WG_THREADS)
      s_cb[i] = A.ix.tables[TAB_SBASE + i] + A.ix.tables[TAB_C + i % SYMS];
   __syncthreads();

   WaveQueue wq;
   wq.q = s_q[wid];
   wq.spill = A.spill + (size_t)gw * A.spill_cap;
   wq.bot = wq.top = wq.spill_top = 0;
   wq.spill_cap = A.spill_cap;
   wq.error = A.error;
   uint4 *stage = s_stage[wid];
   HitRing hr;
   hr.ring = s_hit[wid];
   hr.head = hr.tail = 0;
   unsigned long long n_iter = 0, n_child_iter = 0;

   const int seeds_per = A.tau + 1;
   const int reads_per_refill = 64 / seeds_per;
   const unsigned long long lt_mask = (1ull << lane) - 1ull;
   const uint4 *occ4 = reinterpret_cast<const uint4 *>(A.ix.occ);

   bool reads_left = true;
   bool have = false;
   NodeR cur;
   cur.fp = cur.rp = cur.sz = 0; cur.read = 0; cur.mm = 0;
   cur.pos = cur.end = cur.phase = cur.score = cur.sref = cur.tau = cur.diff = 0;
   cur.dash = cur.fw = false;
   uint32_t qword = 0;          // four query symbols around cur.pos
   int qword_at = -1;           // cur.pos >> 2 of the cached word, -1 = none
   unsigned long long ext_count = 0;

   for (;;) {
      // ------------------------------------------------------------ hand nodes to idle lanes
      unsigned long long needm = __ballot(!have);
      int nneed = __popcll(needm);
      if (nneed) {
         int count = (int)(wq.top - wq.bot);
         if (count < nneed) {
            if (wq.spill_top > 0) {
               wq.spill_top -= LCHUNK;
               for (int i = lane; i < LCHUNK; i += 64) {
                  const uint4 *sp = reinterpret_cast<const uint4 *>(wq.spill + wq.spill_top + i);
                  uint4 x = sp[0], y = sp[1];
                  Node t;
                  t.w[0] = x.x; t.w[1] = x.y; t.w[2] = x.z; t.w[3] = x.w;
                  t.w[4] = y.x; t.w[5] = y.y; t.w[6] = y.z; t.w[7] = y.w;
                  lds_store_node<LQCAP>(wq.q, wq.top + i, t);
               }
               wq.top += LCHUNK;
            } else if (reads_left) {
               // Admit reads: one seed per leaf block of the schedule (src/blocksearch.c:138-144).
               unsigned long long base = 0;
               if (lane == 0) base = atomicAdd(A.cursor, (unsigned long long)reads_per_refill);
               base = __shfl(base, 0);
               if (base >= (unsigned long long)A.n_reads) {
                  reads_left = false;
               } else {
                  int got = (int)min((unsigned long long)reads_per_refill, (unsigned long long)A.n_reads - base);
                  int r = lane / seeds_per, leaf = lane - r * seeds_per;
                  bool mk = r < got;
                  Node packed;
                  for (int k = 0; k < 8; k++) packed.w[k] = 0;
                  if (mk) {
                     uint32_t read = (uint32_t)(base + r);
                     int m = A.read_len[read];
                     int row = m <= MAX_READ ? A.len_row[m] : 0xff;
                     mk = row != 0xff;
                     if (mk) {
                        NodeR sd;
                        sd.fp = 0; sd.rp = 0; sd.sz = A.ix.n;        // bwt_new_query, src/index_bwt.c:48-52
                        sd.read = read; sd.score = 0; sd.mm = (uint64_t)row << 56;
                        sd.sref = 0; sd.pos = sd.end = sd.tau = sd.diff = sd.phase = 0; sd.dash = sd.fw = false;
                        enter_phase(sd, leaf, A.phases[row * MAX_PHASES + leaf]);
                        pack_node(sd, packed);
                     }
                  }
                  wave_push<LQCAP, LCHUNK>(wq, mk, packed, lane);
               }
            }
            wave_sync();
            count = (int)(wq.top - wq.bot);
         }
         if (!have) {
            int rank = __popcll(needm & lt_mask);
            if (rank < count) {
               Node nd;
               lds_load_node<LQCAP>(wq.q, wq.top - 1 - (uint32_t)rank, nd);
               unpack_node(nd, cur);
               have = true;
               qword_at = -1;
            }
         }
         wq.top -= (uint32_t)min(nneed, count);
         wave_sync();
      }
      if (!__any(have)) {
         if (!reads_left && wq.spill_top == 0 && wq.top == wq.bot) break;
         continue;
      }
      ext_count += (unsigned long long)__popcll(__ballot(have));
      n_iter++;

      // ------------------------------------------------------------ fetch: all loads first, then the hand-over
      const int64_t a = cur.fw ? cur.rp : cur.fp, other = cur.fw ? cur.fp : cur.rp;
      const int64_t e0 = a, e1 = a + cur.sz;                       // ranks at a-1 and a+sz-1, exclusive form
      const uint32_t b0 = (uint32_t)(e0 >> BLOCK_POS_LOG), b1 = (uint32_t)(e1 >> BLOCK_POS_LOG);
      const bool need1 = have && b1 != b0;
      uint4 f00, f01, f02, f03, f10, f11, f12, f13;
      f00 = f01 = f02 = f03 = f10 = f11 = f12 = f13 = make_uint4(0, 0, 0, 0);
      quad_fetch<0>(occ4, b0, have, sub, f00); quad_fetch<1>(occ4, b0, have, sub, f01);
      quad_fetch<2>(occ4, b0, have, sub, f02); quad_fetch<3>(occ4, b0, have, sub, f03);
      quad_fetch<0>(occ4, b1, need1, sub, f10); quad_fetch<1>(occ4, b1, need1, sub, f11);
      quad_fetch<2>(occ4, b1, need1, sub, f12); quad_fetch<3>(occ4, b1, need1, sub, f13);
      if (have && (cur.pos >> 2) != qword_at) {
         qword_at = cur.pos >> 2;
         qword = *reinterpret_cast<const uint32_t *>(A.reads + (size_t)cur.read * A.stride + (cur.pos & ~3));
      }
      const int qc = (int)((qword >> (8 * (cur.pos & 3))) & 0xff);

      uint32_t blk[BLOCK_DWORDS];
      uint32_t rel0[SYMS], rel1[SYMS];
      quad_stage<0>(stage, have, lane, sub, f00); quad_stage<1>(stage, have, lane, sub, f01);
      quad_stage<2>(stage, have, lane, sub, f02); quad_stage<3>(stage, have, lane, sub, f03);
      wave_sync();
      for (int k = 0; k < BLOCK_DWORDS; k++) blk[k] = 0;
      if (have) stage_read(stage, lane, blk);
      block_rank(blk, (int)(e0 & (BLOCK_POS - 1)), rel0);
      wave_sync();
      quad_stage<0>(stage, need1, lane, sub, f10); quad_stage<1>(stage, need1, lane, sub, f11);
      quad_stage<2>(stage, need1, lane, sub, f12); quad_stage<3>(stage, need1, lane, sub, f13);
      wave_sync();
      if (need1) stage_read(stage, lane, blk);
      block_rank(blk, (int)(e1 & (BLOCK_POS - 1)), rel1);
      wave_sync();

      // ------------------------------------------------------------ bwt_query_all, src/index_bwt.c:262-273
      const int sb0 = (int)(e0 >> SUPER_LOG), sb1 = (int)(e1 >> SUPER_LOG);
      int64_t szn[SYMS], rpn[SYMS];
      for (int j = 0; j < SYMS; j++) szn[j] = (int64_t)rel1[j] - (int64_t)rel0[j];
      if (have && sb1 != sb0)
         for (int j = 0; j < SYMS; j++) szn[j] += (int64_t)(s_cb[sb1 * SYMS + j] - s_cb[sb0 * SYMS + j]);
      const int64_t tot = szn[0] + szn[1] + szn[2] + szn[3] + szn[4];
      rpn[3] = other + (cur.sz - tot);
      rpn[2] = rpn[3] + szn[3];
      rpn[1] = rpn[2] + szn[2];
      rpn[0] = rpn[1] + szn[1];
      rpn[4] = rpn[0] + szn[0];

      // candidate query symbols: non-empty children; the dash follows only the query symbol (N: all)
      uint32_t cand = 0;
      if (have) {
         uint32_t alive = (szn[0] >= 1 ? 1u : 0u) | (szn[1] >= 1 ? 2u : 0u) | (szn[2] >= 1 ? 4u : 0u) |
                          (szn[3] >= 1 ? 8u : 0u) | (szn[4] >= 1 ? 16u : 0u);           // by internal symbol
         if (cur.fw) alive = ((alive & 1u) << 3) | ((alive & 2u) << 1) | ((alive & 4u) >> 1) | ((alive & 8u) >> 3) | (alive & 16u);
         cand = alive;                                                                   // by query symbol
         if (cur.dash && qc != UNKNOWN) cand &= 1u << qc;
      }
      const int row = (int)(cur.mm >> 56);
      NodeR next = cur;
      bool kept = false;
      while (__any(cand != 0)) {
         n_child_iter++;
         bool push = false, emit = false;
         Node out;
         for (int k = 0; k < 8; k++) out.w[k] = 0;
         if (cand != 0) {
            const int nt = __ffs((int)cand) - 1;
            cand &= cand - 1;
            const int js = cur.fw ? complement(nt) : nt;
            const int64_t fpn = (int64_t)(s_cb[sb0 * SYMS + js] + sel5(rel0, js));
            const int64_t rp_ = sel5(rpn, js), sz_ = sel5(szn, js);
            NodeR child;
            int kind = make_child(cur, nt, qc, cur.fw ? rp_ : fpn, cur.fw ? fpn : rp_, sz_, child);
            if (kind == CHILD_DONE) {
               int nxt = phase_next(A.phases[row * MAX_PHASES + cur.phase]);
               if (nxt == PHASE_FINAL) { emit = true; child.pos = 0; pack_node(child, out); kind = CHILD_NONE; }
               else { enter_phase(child, nxt, A.phases[row * MAX_PHASES + nxt]); kind = CHILD_NODE; }
            }
            if (kind == CHILD_NODE) {
               if (!kept) { next = child; kept = true; }
               else { push = true; pack_node(child, out); }
            }
         }
         if (__any(push)) wave_push<LQCAP, LCHUNK>(wq, push, out, lane);
         hit_emit(A, hr, emit, out, lane);
      }
      if (have) {
         cur = next;
         have = kept;
      }
      wave_sync();
   }
   if (hr.tail != hr.head) hit_flush(A, hr, lane, hr.tail - hr.head);
   if (lane == 0 && ext_count) {
      atomicAdd(A.n_ext, ext_count);
      atomicAdd(A.stats + 0, n_iter);
      atomicAdd(A.stats + 1, n_child_iter);
   }
}

}  // namespace gm
