// gm_core.hpp -- data layouts and per-node logic of the search, shared by the HIP kernels
// (gm_device.hip) and by a host-compiled checker of the same code (tests/hostemu).  Everything
// here is `GM_HD`: plain integer code with no device intrinsics.
//
// Reference semantics restated here (cited per function):
//   bwt_query_all            src/index_bwt.c:198-304
//   get_occ_all              src/index_bwt.c:840-905      (rank; our block layout differs)
//   blocksearch_trail_rec    src/blocksearch.c:106-177    (flattened into a phase table)
//   seqsearch_fw/bw          src/blocksearch.c:184-311
//   seqdash_fw/bw            src/blocksearch.c:381-496
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define GM_HD __host__ __device__ __forceinline__
#else
#define GM_HD inline
#endif

namespace gm {

constexpr int SYMS = 5;               // A C G T N; separators are code 5 and are never counted
constexpr int UNKNOWN = 4;            // UNKNOWN_BASE, src/blocksearch.h:14
constexpr int MAX_READ = 256;         // src/blocksearch.h:12-13
constexpr int MAX_TAU = 7;

// ------------------------------------------------------------------------------------------------
// Occ blocks.  One block = 64 bytes = 128 BWT positions, cut into four 16-byte sub-blocks so that
// four neighbouring lanes fetch one block with one 16-byte load each:
//
//    sub-block k (k = 0..3), positions [32k, 32k+32) of the block, bit i = position 32k + i:
//       dword 0: bit plane b0      code = b0 + 2*b1 + 4*b2:  A=0 C=1 G=2 T=3 N=4 $=5
//       dword 1: bit plane b1
//       dword 2: bit plane b2
//       dword 3: bits  0..24  count of symbol k (A,C,G,T) in BWT[superblock start, block start)
//                bits 25..31  bits [7k, 7k+7) of the same count for N
//
// Counts are relative to the enclosing superblock (2^25 positions); the absolute counts at each
// superblock start (plus C[]) live in a small side table.  Density: 0.5 byte per position
// (the reference layout: 0.67, src/index_bwt.c:525-564).  One extra block at the end lets a rank
// at position n read totals.
// ------------------------------------------------------------------------------------------------
constexpr int BLOCK_POS_LOG = 7;
constexpr int BLOCK_POS = 1 << BLOCK_POS_LOG;       // 128
constexpr int BLOCK_DWORDS = 16;
constexpr int SUPER_LOG = 25;
constexpr uint32_t CNT_MASK = (1u << 25) - 1;

GM_HD int64_t num_blocks(int64_t n) { return (n >> BLOCK_POS_LOG) + 1; }
GM_HD int64_t num_supers(int64_t n) { return (n >> SUPER_LOG) + 1; }

// Counts of A,C,G,T,N among the first t (0..32) positions of one sub-block.
GM_HD void sub_counts(uint32_t b0, uint32_t b1, uint32_t b2, int t, uint32_t cnt[SYMS])
{
   uint32_t mask = t >= 32 ? 0xffffffffu : ((1u << t) - 1u);
   uint32_t lo = ~b2 & mask;         // A C G T
#if defined(__HIP_DEVICE_COMPILE__)
   cnt[0] = __popc(lo & ~b1 & ~b0);
   cnt[1] = __popc(lo & ~b1 & b0);
   cnt[2] = __popc(lo & b1 & ~b0);
   cnt[3] = __popc(lo & b1 & b0);
   cnt[4] = __popc(b2 & ~b0 & mask);
#else
   cnt[0] = (uint32_t)__builtin_popcount(lo & ~b1 & ~b0);
   cnt[1] = (uint32_t)__builtin_popcount(lo & ~b1 & b0);
   cnt[2] = (uint32_t)__builtin_popcount(lo & b1 & ~b0);
   cnt[3] = (uint32_t)__builtin_popcount(lo & b1 & b0);
   cnt[4] = (uint32_t)__builtin_popcount(b2 & ~b0 & mask);
#endif
}

// Whole-block rank, one thread: counts (relative to the superblock) of the five symbols in
// BWT[superblock start, e) for e inside block `blk` (64-byte aligned).  The kernels split this over
// four lanes; this form serves the host checker and the small operator kernels.
GM_HD void block_rank(const uint32_t *blk, int o /* 0..127 positions inside the block */, uint32_t rel[SYMS])
{
   uint32_t n_cnt = 0;
   for (int j = 0; j < SYMS; j++) rel[j] = 0;
   for (int k = 0; k < 4; k++) {
      const uint32_t *s = blk + 4 * k;
      int t = o - 32 * k; t = t < 0 ? 0 : (t > 32 ? 32 : t);
      uint32_t c[SYMS];
      sub_counts(s[0], s[1], s[2], t, c);
      for (int j = 0; j < SYMS; j++) rel[j] += c[j];
      rel[k] += s[3] & CNT_MASK;
      n_cnt |= (s[3] >> 25) << (7 * k);
   }
   rel[4] += n_cnt;
}

// Building one block from the reference's Occ array (src/index_bwt.c:525-564: word w's five planes
// sit at (w + w/intv + 1)*5, MSB first).  rel[j] = counts of symbol j in BWT[superblock start,
// block start), which the caller accumulates.
GM_HD uint32_t brev32(uint32_t v)
{
#if defined(__HIP_DEVICE_COMPILE__)
   return __brev(v);
#else
   v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
   v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
   v = ((v >> 4) & 0x0f0f0f0fu) | ((v & 0x0f0f0f0fu) << 4);
   v = ((v >> 8) & 0x00ff00ffu) | ((v & 0x00ff00ffu) << 8);
   return (v >> 16) | (v << 16);
#endif
}

GM_HD void pack_sub(uint32_t a, uint32_t c, uint32_t g, uint32_t t, uint32_t n, uint32_t cnt_word, uint32_t *out)
{
   uint32_t sep = ~(a | c | g | t | n);              // separators and padding past the text end
   out[0] = c | t | sep;                             // b0: C=1 T=3 $=5
   out[1] = g | t;                                   // b1: G=2 T=3
   out[2] = n | sep;                                 // b2: N=4 $=5
   out[3] = cnt_word;
}

GM_HD uint32_t count_word(const uint32_t rel[SYMS], int k)
{
   return (rel[k] & CNT_MASK) | (((rel[4] >> (7 * k)) & 0x7fu) << 25);
}

GM_HD void pack_block(const uint64_t *ref_occ, int64_t mark_intv, int64_t n_ref_words, int64_t blk,
                      const uint32_t rel[SYMS], uint32_t out[BLOCK_DWORDS])
{
   uint64_t p0[SYMS], p1[SYMS];                      // planes of the two reference words of this block
   const int64_t w0 = 2 * blk, w1 = 2 * blk + 1;
   for (int j = 0; j < SYMS; j++) {
      p0[j] = w0 < n_ref_words ? ref_occ[(w0 + w0 / mark_intv + 1) * SYMS + j] : 0;
      p1[j] = w1 < n_ref_words ? ref_occ[(w1 + w1 / mark_intv + 1) * SYMS + j] : 0;
   }
   // MSB-first words: positions 0..31 are the high half, bit-reversed into our LSB-first order.
   pack_sub(brev32((uint32_t)(p0[0] >> 32)), brev32((uint32_t)(p0[1] >> 32)), brev32((uint32_t)(p0[2] >> 32)),
            brev32((uint32_t)(p0[3] >> 32)), brev32((uint32_t)(p0[4] >> 32)), count_word(rel, 0), out + 0);
   pack_sub(brev32((uint32_t)p0[0]), brev32((uint32_t)p0[1]), brev32((uint32_t)p0[2]),
            brev32((uint32_t)p0[3]), brev32((uint32_t)p0[4]), count_word(rel, 1), out + 4);
   pack_sub(brev32((uint32_t)(p1[0] >> 32)), brev32((uint32_t)(p1[1] >> 32)), brev32((uint32_t)(p1[2] >> 32)),
            brev32((uint32_t)(p1[3] >> 32)), brev32((uint32_t)(p1[4] >> 32)), count_word(rel, 2), out + 8);
   pack_sub(brev32((uint32_t)p1[0]), brev32((uint32_t)p1[1]), brev32((uint32_t)p1[2]),
            brev32((uint32_t)p1[3]), brev32((uint32_t)p1[4]), count_word(rel, 3), out + 12);
}

// ------------------------------------------------------------------------------------------------
// Search nodes.  A node is one pending call of seqsearch_* / one iteration of seqdash_*: a
// bi-interval, the next query position and the bookkeeping of the enclosing phase.  32 bytes.
//
//   w0 fp[31:0]  w1 rp[31:0]  w2 sz[31:0]  w3 fp[39:32] | rp[39:32]<<8 | sz[39:32]<<16 | pos<<24
//   w4 read index
//   w5 phase(6) | dash(1)<<6 | score(3)<<7 | sref(3)<<10 | fw(1)<<13 | end(8)<<14 | tau(3)<<22 | diff(3)<<25 |
//      first(3)<<28: a node pushed back as a continuation resumes with the candidate symbols >= first
//   w6,w7 mismatch positions, one byte each, in the order found (at most 7); byte 7 of the pair
//         holds the row of the read's length in the phase tables (one row per read length of the batch)
// A finished node (hit) is stored in the same format (pos field unused).
// ------------------------------------------------------------------------------------------------
struct Node {
   uint32_t w[8];
};

struct NodeR {      // unpacked, in registers
   int64_t  fp, rp, sz;
   uint32_t read;
   uint64_t mm;     // mismatch positions + phase-table row << 56
   int pos, end, phase, score, sref, tau, diff;
   int first = 0;   // first candidate symbol still to be followed (continuations of k_search)
   bool dash, fw;
};

constexpr int META_DASH = 6, META_SCORE = 7, META_SREF = 10, META_FW = 13, META_END = 14, META_TAU = 22,
              META_DIFF = 25, META_FIRST = 28;
// Entries of the output list of k_search: a finished hit (an interval of the whole read), or, bit 31 set,
// a node whose interval holds one suffix and that k_verify finishes against the text (the list entry
// stays as it is; what becomes of it goes to the locate stage as a sort key).
constexpr int META_CAND = 31;
// The search of blocksc_trail (k_search<true>, no k_verify there): bit 31 of a node inside the bounded extension = every
// letter of the extension so far equals the query's (`boundary` of scsearch_fw, src/blocksearch.c:338, :366).
constexpr int META_BOUND = 31;

constexpr int PHASE_FINAL = 63;

GM_HD uint32_t pack_meta(const NodeR &n)
{
   return (uint32_t)n.phase | ((uint32_t)n.dash << META_DASH) | ((uint32_t)n.score << META_SCORE) |
          ((uint32_t)n.sref << META_SREF) | ((uint32_t)n.fw << META_FW) | ((uint32_t)n.end << META_END) |
          ((uint32_t)n.tau << META_TAU) | ((uint32_t)n.diff << META_DIFF) | ((uint32_t)n.first << META_FIRST);
}

GM_HD void pack_node(const NodeR &n, Node &o)
{
   o.w[0] = (uint32_t)n.fp; o.w[1] = (uint32_t)n.rp; o.w[2] = (uint32_t)n.sz;
   o.w[3] = (uint32_t)((n.fp >> 32) & 0xff) | ((uint32_t)((n.rp >> 32) & 0xff) << 8) |
            ((uint32_t)((n.sz >> 32) & 0xff) << 16) | ((uint32_t)n.pos << 24);
   o.w[4] = n.read;
   o.w[5] = pack_meta(n);
   o.w[6] = (uint32_t)n.mm; o.w[7] = (uint32_t)(n.mm >> 32);
}

GM_HD void unpack_node(const Node &i, NodeR &n)
{
   n.fp = (int64_t)i.w[0] | ((int64_t)(i.w[3] & 0xff) << 32);
   n.rp = (int64_t)i.w[1] | ((int64_t)((i.w[3] >> 8) & 0xff) << 32);
   n.sz = (int64_t)i.w[2] | ((int64_t)((i.w[3] >> 16) & 0xff) << 32);
   n.pos = (int)(i.w[3] >> 24);
   n.read = i.w[4];
   uint32_t m = i.w[5];
   n.phase = m & 63; n.dash = (m >> META_DASH) & 1; n.score = (m >> META_SCORE) & 7; n.sref = (m >> META_SREF) & 7;
   n.fw = (m >> META_FW) & 1; n.end = (m >> META_END) & 255; n.tau = (m >> META_TAU) & 7; n.diff = (m >> META_DIFF) & 7;
   n.first = (m >> META_FIRST) & 7;
   n.mm = (uint64_t)i.w[6] | ((uint64_t)i.w[7] << 32);
}

// ------------------------------------------------------------------------------------------------
// Phase table: blocksearch_trail_rec (src/blocksearch.c:138-170) flattened.  A search path never
// has to meet another one: a hit of sub-block X is extended over the rest of X's parent, so every
// path just walks up the recursion tree.  One 32-bit entry per phase:
//   start(8) | end(8)<<8 | fw(1)<<16 | tau(4)<<17 | diff(4)<<21 | next(6)<<25
// Phases 0..tau are the leaves (exact search of one block, right to left, from the root interval);
// the others extend hits of the left child to the right (diff 0) or of the right child to the
// left (diff = blocks of the left child).  next = PHASE_FINAL when the extension completes the read.
// ------------------------------------------------------------------------------------------------
constexpr int MAX_PHASES = 3 * MAX_TAU + 1;

GM_HD uint32_t phase_entry(int start, int end, int fw, int tau, int diff, int next)
{
   return (uint32_t)start | ((uint32_t)end << 8) | ((uint32_t)fw << 16) | ((uint32_t)tau << 17) |
          ((uint32_t)diff << 21) | ((uint32_t)next << 25);
}

struct PhaseBuilder {
   uint32_t entry[MAX_PHASES];
   uint8_t lo[MAX_PHASES];        // lowest query position already matched when the phase starts (forward phases)
   int n_leaf = 0, n_ext = 0, tau_total;
   bool error = false;
   // returns nothing; fills entries.  cont = phase entered when a path completes [pos,end]
   void build(int pos, int end, int blocks, int cont)
   {
      if (pos < 0 || pos > end || end < 0) { error = true; return; }     // src/blocksearch.c:125-127
      if (blocks == 1) {                                                 // :138-144
         lo[n_leaf] = 0;                                                  // backward phases: lo = node position + 1
         entry[n_leaf++] = phase_entry(end, pos, 0, 0, 0, cont);
         return;
      }
      int blk_l = (blocks >> 1) + (blocks & 1), blk_r = blocks >> 1;     // :147-148
      int slen = end - pos + 1;
      int pos_r = pos + (slen >> 1) + (slen & 1), end_l = pos_r - 1;      // :151-153
      int ext_l = tau_total + 1 + n_ext++, ext_r = tau_total + 1 + n_ext++;
      entry[ext_l] = phase_entry(pos_r, end, 1, blocks - 1, 0, cont);     // :159-162
      entry[ext_r] = phase_entry(end_l, pos, 0, blocks - 1, blk_l, cont); // :167-170
      lo[ext_l] = (uint8_t)pos; lo[ext_r] = 0;
      build(pos, end_l, blk_l, ext_l);                                    // :157
      if (error) return;
      build(pos_r, end, blk_r, ext_r);                                    // :165
   }
};

// Fills `table[0 .. 3*tau]`; returns false where the reference returns -1 for this (m, tau).
// lo (optional): per phase, the lowest query position matched before a forward phase starts.
inline bool build_phase_table(int m, int tau, uint32_t *table, uint8_t *lo = nullptr)
{
   if (m < 1 || m > MAX_READ || tau < 0 || tau > MAX_TAU) return false;
   PhaseBuilder b;
   b.tau_total = tau;
   b.build(0, m - 1, tau + 1, PHASE_FINAL);
   if (b.error) return false;
   for (int i = 0; i < 3 * tau + 1; i++) { table[i] = b.entry[i]; if (lo) lo[i] = b.lo[i]; }
   return true;
}

// blocksc_trail (src/blocksearch.c:7-103) as a phase table: the block search of the left part query[0, pos_r) with
// tau_l + 1 blocks (:74), whose hits continue into ONE more phase, the lexicographically bounded forward extension
// over [pos_r, slen) with up to tau_eff mismatches in all (scsearch_fw, :89-92).  tau_eff = tau - (N in the query),
// last = the query's interval lies at or behind its reverse complement's (:55).  Phases 0..3*tau_l as
// build_phase_table(pos_r, tau_l) gives them, phase 3*tau_l + 1 the bounded extension.  Returns 0 = search,
// 1 = no search and no hits (:64-66; an empty left block: :73), -1 = where the reference returns -1 (and for an
// empty right part, where it reads behind the query).
// tau_l_forced >= 0: the left part is searched with that many mismatches instead (gpumap_ann_build with the
// reference's reuse of the last k-mer's left block, whose search ran with the budget of the k-mer that made it).
constexpr int SC_SEARCH = 0, SC_NO_HITS = 1, SC_ERROR = -1;
GM_HD int sc_pos_r(int slen, int last) { return slen / 2 + (last ? slen % 2 : 0); }                       // :69
GM_HD int sc_tau_l(int tau_eff, int last) { return tau_eff / 2 - (last ? 0 : (1 - tau_eff % 2)); }       // :70
inline int build_sc_phase_table(int slen, int tau_arg, int tau_eff, int last, uint32_t *table, uint8_t *lo, int *sc_phase, int *leaves,
                                int tau_l_forced = -1)
{
   *sc_phase = 0; *leaves = 0;
   if (slen < 1 || slen > MAX_READ) return SC_ERROR;
   if (tau_eff < 0 || (tau_eff == 0 && !last)) return SC_NO_HITS;                  // :64-66
   if (tau_arg == 0) return SC_ERROR;               // the caller's tree, alloc_stack_tree(0), has no left subtree (:74 -> :123)
   const int pos_r = sc_pos_r(slen, last);                                         // :69
   const int tau_l = tau_l_forced >= 0 ? tau_l_forced : sc_tau_l(tau_eff, last);   // :70
   if (pos_r > slen - 1) return SC_ERROR;
   if (pos_r == 0) return SC_NO_HITS;                                              // :73 with trail = 0
   PhaseBuilder b;
   b.tau_total = tau_l;
   const int scp = 3 * tau_l + 1;
   b.build(0, pos_r - 1, tau_l + 1, scp);
   if (b.error) return SC_ERROR;
   for (int i = 0; i < scp; i++) { table[i] = b.entry[i]; lo[i] = b.lo[i]; }
   table[scp] = phase_entry(pos_r, slen - 1, 1, tau_eff, 0, PHASE_FINAL);         // :91
   lo[scp] = 0;
   *sc_phase = scp; *leaves = tau_l + 1;
   return SC_SEARCH;
}

GM_HD void enter_phase(NodeR &n, int phase, uint32_t e)
{
   n.phase = phase;
   n.pos = e & 255; n.end = (e >> 8) & 255; n.fw = (e >> 16) & 1;
   n.tau = (e >> 17) & 15; n.diff = (e >> 21) & 15;
   n.sref = n.score;
   n.dash = n.tau == 0;      // leaves: seqsearch_bw with tau 0 degenerates to the exact dash (:138-144)
}
GM_HD int phase_next(uint32_t e) { return (int)(e >> 25) & 63; }

// ------------------------------------------------------------------------------------------------
// One extension step.
// ------------------------------------------------------------------------------------------------

// bwt_query_all (src/index_bwt.c:198-304), given the two ranks.  occ0 / occ1: counts of each symbol
// in BWT[0, a) and BWT[0, a + sz), a = fw ? rp : fp, *without* C[] added; cb0[j] = C[j].
// Produces, for internal symbol j, the new pointer on the searched side (fpn), the interval size,
// and the pointer on the other side (rpn, accumulated in the order T,G,C,A,N :270-273).
GM_HD void extend_all(int64_t other, int64_t sz, const int64_t occ0[SYMS], const int64_t occ1[SYMS],
                      const uint64_t c[SYMS], int64_t fpn[SYMS], int64_t rpn[SYMS], int64_t szn[SYMS])
{
   int64_t tot = 0;
   for (int j = 0; j < SYMS; j++) {
      fpn[j] = (int64_t)c[j] + occ0[j];
      szn[j] = occ1[j] - occ0[j];
      tot += szn[j];
   }
   rpn[3] = other + (sz - tot);
   rpn[2] = rpn[3] + szn[3];
   rpn[1] = rpn[2] + szn[2];
   rpn[0] = rpn[1] + szn[1];
   rpn[4] = rpn[0] + szn[0];
}

GM_HD int complement(int s) { return s < 4 ? 3 - s : s; }      // src/index_sym.c:12

enum ChildKind { CHILD_NONE = 0, CHILD_NODE = 1, CHILD_DONE = 2 };

// The body of the `for nt` loops of seqsearch_* (src/blocksearch.c:207-238, :273-302) and of
// seqdash_* (:396-428, :455-487) for one symbol nt of the query alphabet.  `cur` is the node being
// extended, qc the query symbol at cur.pos, (cfp, crp, csz) the child interval for nt.
// CHILD_NODE: `child` continues inside the same phase.  CHILD_DONE: `child` has consumed the last
// position of the phase (the reference pushes it on the stack of the enclosing block); the caller
// moves it to the next phase or emits it.
GM_HD int make_child(const NodeR &cur, int nt, int qc, int64_t cfp, int64_t crp, int64_t csz, NodeR &child)
{
   if (csz < 1) return CHILD_NONE;                                  // :209, :400, :417
   int s = cur.score;
   bool dash = cur.dash;
   uint64_t mm = cur.mm;
   if (cur.dash) {
      if (qc != UNKNOWN && nt != qc) return CHILD_NONE;             // exact continuation; N follows all
   } else {
      int ds = (nt != qc && qc != UNKNOWN);                         // :212
      s += ds;
      if (s > cur.tau) return CHILD_NONE;                           // :215
      if (ds) mm |= (uint64_t)cur.pos << (8 * cur.score);           // aln_bit_set (:221); N bits are implied
      if (s == cur.tau) {
         if (s - cur.sref < cur.diff) return CHILD_NONE;            // :223-227
         dash = true;
      } else if (cur.pos == cur.end) {
         if (s - cur.sref < cur.diff) return CHILD_NONE;            // :229-233
      }
   }
   child = cur;
   child.first = 0;
   child.fp = cfp; child.rp = crp; child.sz = csz;
   child.score = s; child.dash = dash; child.mm = mm;
   if (cur.pos == cur.end) return CHILD_DONE;
   child.pos = cur.fw ? cur.pos + 1 : cur.pos - 1;
   return CHILD_NODE;
}

// ------------------------------------------------------------------------------------------------
// The same rules on the packed meta word (k_search_lane keeps a node as fp, rp, sz, pos, meta, mm).
// ------------------------------------------------------------------------------------------------

// Of the non-empty children `alive` (bit nt = query symbol nt), those that make_child accepts.
GM_HD uint32_t valid_children(uint32_t meta, int pos, int qc, uint32_t alive)
{
   const uint32_t match = qc == UNKNOWN ? 0x1fu : (1u << qc);
   if ((meta >> META_DASH) & 1) return alive & match;
   const int score = (meta >> META_SCORE) & 7, sref = (meta >> META_SREF) & 7, tau = (meta >> META_TAU) & 7,
             diff = (meta >> META_DIFF) & 7, end = (meta >> META_END) & 255;
   const bool last = pos == end;
   const int s0 = score, s1 = score + 1;
   const bool ok0 = s0 <= tau && (s0 == tau ? s0 - sref >= diff : !(last && s0 - sref < diff));
   const bool ok1 = s1 <= tau && (s1 == tau ? s1 - sref >= diff : !(last && s1 - sref < diff));
   return alive & ((ok0 ? match : 0u) | (ok1 ? ~match & 0x1fu : 0u));
}

// Child of a node through an accepted symbol nt: new meta (score, dash) and mismatch list; returns
// true when the child has consumed the last position of its phase (CHILD_DONE).
GM_HD bool child_meta(uint32_t meta, int pos, int qc, int nt, uint64_t mm, uint32_t &meta_out, uint64_t &mm_out)
{
   const uint32_t match = qc == UNKNOWN ? 0x1fu : (1u << qc);
   const bool dash = (meta >> META_DASH) & 1;
   const int score = (meta >> META_SCORE) & 7, tau = (meta >> META_TAU) & 7, end = (meta >> META_END) & 255;
   const bool ds = !dash && !((match >> nt) & 1);
   const int s = score + (ds ? 1 : 0);
   mm_out = ds ? (mm | ((uint64_t)pos << (8 * score))) : mm;
   meta_out = (meta & ~((7u << META_SCORE) | (1u << META_DASH) | (7u << META_FIRST))) | ((uint32_t)s << META_SCORE) |
              ((uint32_t)(dash || s == tau) << META_DASH);
   return pos == end;
}

// Phase records for k_search_lane: what a node becomes when it completes phase p.  Two words:
//   x: meta template of the next phase (phase, dash = tau == 0, fw, end, tau, diff; score fields 0)
//   y: start position of the next phase | final << 8 (final: the node is a hit of the read)
GM_HD void phase_record(const uint32_t *table /* row of build_phase_table */, int p, uint32_t &x, uint32_t &y)
{
   int nxt = phase_next(table[p]);
   if (nxt == PHASE_FINAL) { x = 0; y = 1u << 8; return; }
   uint32_t e = table[nxt];
   int start = e & 255, end = (e >> 8) & 255, fw = (e >> 16) & 1, tau = (e >> 17) & 15, diff = (e >> 21) & 15;
   x = (uint32_t)nxt | ((uint32_t)(tau == 0) << META_DASH) | ((uint32_t)fw << META_FW) | ((uint32_t)end << META_END) |
       ((uint32_t)tau << META_TAU) | ((uint32_t)diff << META_DIFF);
   y = (uint32_t)start;
}

// Mismatch bitmap of a finished hit: the recorded mismatch positions plus every query N
// (src/blocksearch.c:221, :406, :465 set the bit of an N position on every path through it).
GM_HD void hit_align(uint64_t mm, int score, const uint8_t *query, int m, uint64_t align[4])
{
   for (int w = 0; w < 4; w++) align[w] = 0;
   for (int i = 0; i < score; i++) {
      int p = (int)((mm >> (8 * i)) & 0xff);
      align[p >> 6] |= (uint64_t)1 << (p & 63);
   }
   for (int p = 0; p < m; p++)
      if (query[p] == UNKNOWN) align[p >> 6] |= (uint64_t)1 << (p & 63);
}

}  // namespace gm
