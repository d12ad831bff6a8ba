// hostemu.cpp -- TEST INFRASTRUCTURE.  Compiles the shared integer core of the HIP kernels
// (mapper_amd/csrc/gm_core.hpp: block layout, node packing, phase table, per-node step) for the
// host and runs it sequentially, one node at a time from an explicit stack.  tests/ compare its
// output with the oracle, so that layout and schedule mistakes are found without a GPU.  It is not
// part of libgpumap.so and is never a fallback for it.
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "gm_core.hpp"

using namespace gm;

struct EmuIndex {
   int64_t n;
   std::vector<uint32_t> blocks;
   std::vector<uint64_t> sbase;   // [n_super][5]
   uint64_t c[SYMS];
};

static void ref_rank_excl(const uint64_t *occ, int64_t intv, int64_t e, uint64_t cnt[SYMS])
{
   int64_t w = e >> 6;
   int bit = (int)(e & 63);
   int64_t mk = w / intv;
   const uint64_t *m = occ + mk * (intv + 1) * SYMS;
   for (int j = 0; j < SYMS; j++) cnt[j] = m[j];
   for (int64_t ww = mk * intv; ww < w; ww++) {
      const uint64_t *g = occ + (ww + ww / intv + 1) * SYMS;
      for (int j = 0; j < SYMS; j++) cnt[j] += __builtin_popcountll(g[j]);
   }
   if (bit) {
      const uint64_t *g = occ + (w + w / intv + 1) * SYMS;
      for (int j = 0; j < SYMS; j++) cnt[j] += __builtin_popcountll(g[j] >> (64 - bit));
   }
}

extern "C" void *emu_create(const uint64_t *ref_occ, int64_t mark_intv, const uint64_t *c, int64_t n)
{
   EmuIndex *x = new EmuIndex();
   x->n = n;
   for (int j = 0; j < SYMS; j++) x->c[j] = c[j];
   int64_t nb = num_blocks(n), ns = num_supers(n);
   int64_t n_ref_words = ((n + mark_intv * 64 - 1) / (mark_intv * 64)) * mark_intv;
   x->sbase.resize(ns * SYMS);
   for (int64_t s = 0; s < ns; s++) {
      uint64_t cnt[SYMS];
      int64_t e = s << SUPER_LOG; if (e > n) e = n;
      ref_rank_excl(ref_occ, mark_intv, e, cnt);
      for (int j = 0; j < SYMS; j++) x->sbase[s * SYMS + j] = cnt[j];
   }
   x->blocks.resize(nb * BLOCK_DWORDS);
   for (int64_t b = 0; b < nb; b++) {
      uint64_t cnt[SYMS]; uint32_t rel[SYMS];
      int64_t e = b << BLOCK_POS_LOG;
      ref_rank_excl(ref_occ, mark_intv, e, cnt);
      for (int j = 0; j < SYMS; j++) rel[j] = (uint32_t)(cnt[j] - x->sbase[(e >> SUPER_LOG) * SYMS + j]);
      pack_block(ref_occ, mark_intv, n_ref_words, b, rel, &x->blocks[b * BLOCK_DWORDS]);
   }
   return x;
}

extern "C" void emu_destroy(void *h) { delete (EmuIndex *)h; }

static void rank_abs(const EmuIndex *x, int64_t e, int64_t occ[SYMS])
{
   uint32_t rel[SYMS];
   block_rank(&x->blocks[(e >> BLOCK_POS_LOG) * BLOCK_DWORDS], (int)(e & (BLOCK_POS - 1)), rel);
   for (int j = 0; j < SYMS; j++) occ[j] = (int64_t)(x->sbase[(e >> SUPER_LOG) * SYMS + j] + rel[j]);
}

extern "C" void emu_occ_all(void *h, int64_t ptr, int64_t occ[SYMS]) { rank_abs((EmuIndex *)h, ptr + 1, occ); }

// Returns number of hits (or -1 where the reference returns -1); out = [hits][10]:
// fp, rp, sz, dp, score, align0..3, n_ext (same in every row).
extern "C" int64_t emu_search(void *h, const uint8_t *q, int m, int tau, int64_t **out)
{
   EmuIndex *x = (EmuIndex *)h;
   uint32_t table[MAX_PHASES];
   *out = nullptr;
   if (!build_phase_table(m, tau, table)) return -1;
   std::vector<Node> stack;
   std::vector<int64_t> res;
   int64_t n_ext = 0;
   for (int leaf = tau; leaf >= 0; leaf--) {
      NodeR sd;
      memset(&sd, 0, sizeof sd);
      sd.fp = 0; sd.rp = 0; sd.sz = x->n; sd.read = 0; sd.score = 0; sd.mm = 0;
      enter_phase(sd, leaf, table[leaf]);
      Node p; pack_node(sd, p);
      stack.push_back(p);
   }
   while (!stack.empty()) {
      Node pk = stack.back(); stack.pop_back();
      NodeR cur; unpack_node(pk, cur);
      n_ext++;
      int64_t a = cur.fw ? cur.rp : cur.fp, other = cur.fw ? cur.fp : cur.rp;
      int64_t occ0[SYMS], occ1[SYMS], fpn[SYMS], rpn[SYMS], szn[SYMS];
      rank_abs(x, a, occ0);
      rank_abs(x, a + cur.sz, occ1);
      extend_all(other, cur.sz, occ0, occ1, x->c, fpn, rpn, szn);
      int qc = q[cur.pos];
      // children are pushed in reverse symbol order so that the pop order is the reference's DFS order
      for (int nt = SYMS - 1; nt >= 0; nt--) {
         int js = cur.fw ? complement(nt) : nt;
         NodeR child;
         int kind = make_child(cur, nt, qc, cur.fw ? rpn[js] : fpn[js], cur.fw ? fpn[js] : rpn[js], szn[js], child);
         if (kind == CHILD_NONE) continue;
         if (kind == CHILD_DONE) {
            int nxt = phase_next(table[cur.phase]);
            if (nxt == PHASE_FINAL) {
               uint64_t al[4];
               hit_align(child.mm, child.score, q, m, al);
               int64_t row[10] = {child.fp, child.rp, child.sz, m, child.score, (int64_t)al[0], (int64_t)al[1],
                                  (int64_t)al[2], (int64_t)al[3], 0};
               res.insert(res.end(), row, row + 10);
               continue;
            }
            enter_phase(child, nxt, table[nxt]);
         }
         Node p; pack_node(child, p);
         stack.push_back(p);
      }
   }
   int64_t nh = (int64_t)res.size() / 10;
   int64_t *o = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nh > 0 ? nh * 10 : 1));
   for (int64_t i = 0; i < nh; i++) { memcpy(o + 10 * i, &res[10 * i], 80); o[10 * i + 9] = n_ext; }
   if (nh == 0) o[0] = n_ext;
   *out = o;
   return nh;
}

// The same search in the form k_search runs it (gm_search.hpp): nodes kept as (fp, rp, sz, pos, meta, mm),
// one child per pass chosen with valid_children / child_meta, the other candidates left behind as a
// continuation of the node (meta.first), phase changes through phase_record.  Same output format.
extern "C" int64_t emu_search_packed(void *h, const uint8_t *q, int m, int tau, int64_t **out)
{
   EmuIndex *x = (EmuIndex *)h;
   uint32_t table[MAX_PHASES];
   *out = nullptr;
   if (!build_phase_table(m, tau, table)) return -1;
   uint32_t rec_x[MAX_PHASES], rec_y[MAX_PHASES];
   for (int p = 0; p < 3 * tau + 1; p++) phase_record(table, p, rec_x[p], rec_y[p]);
   std::vector<Node> stack;
   std::vector<int64_t> res;
   int64_t n_ext = 0;
   for (int leaf = tau; leaf >= 0; leaf--) {
      NodeR sd;
      memset(&sd, 0, sizeof sd);
      sd.sz = x->n;
      enter_phase(sd, leaf, table[leaf]);
      Node p; pack_node(sd, p);
      stack.push_back(p);
   }
   while (!stack.empty()) {
      Node nd = stack.back(); stack.pop_back();
      int64_t fp = (int64_t)nd.w[0] | ((int64_t)(nd.w[3] & 0xff) << 32);
      int64_t rp = (int64_t)nd.w[1] | ((int64_t)((nd.w[3] >> 8) & 0xff) << 32);
      int64_t sz = (int64_t)nd.w[2] | ((int64_t)((nd.w[3] >> 16) & 0xff) << 32);
      int pos = (int)(nd.w[3] >> 24);
      uint32_t meta = nd.w[5];
      uint64_t mm = (uint64_t)nd.w[6] | ((uint64_t)nd.w[7] << 32);
      bool have = true;
      while (have) {                                    // the lane keeps its node from pass to pass
         const int first = (int)(meta >> META_FIRST) & 7;
         if (first == 0) n_ext++;
         const bool fw = (meta >> META_FW) & 1;
         int64_t a = fw ? rp : fp, other = fw ? fp : rp;
         int64_t occ0[SYMS], occ1[SYMS], fpn[SYMS], rpn[SYMS], szn[SYMS];
         rank_abs(x, a, occ0);
         rank_abs(x, a + sz, occ1);
         extend_all(other, sz, occ0, occ1, x->c, fpn, rpn, szn);
         uint32_t alive = 0;
         for (int j = 0; j < SYMS; j++) if (szn[j] >= 1) alive |= 1u << (fw ? complement(j) : j);
         const int qc = q[pos];
         uint32_t cand = valid_children(meta, pos, qc, alive) & (0x1fu << first);
         if (cand & (cand - 1)) {
            uint32_t rest = cand & (cand - 1);
            Node c = nd;
            c.w[0] = (uint32_t)fp; c.w[1] = (uint32_t)rp; c.w[2] = (uint32_t)sz;
            c.w[3] = (uint32_t)((fp >> 32) & 0xff) | ((uint32_t)((rp >> 32) & 0xff) << 8) | ((uint32_t)((sz >> 32) & 0xff) << 16) | ((uint32_t)pos << 24);
            c.w[5] = (meta & ~(7u << META_FIRST)) | ((uint32_t)__builtin_ctz(rest) << META_FIRST);
            c.w[6] = (uint32_t)mm; c.w[7] = (uint32_t)(mm >> 32);
            stack.push_back(c);
         }
         if (!cand) { have = false; break; }
         const int nt = __builtin_ctz(cand), js = fw ? complement(nt) : nt;
         fp = fw ? rpn[js] : fpn[js];
         rp = fw ? fpn[js] : rpn[js];
         sz = szn[js];
         uint32_t meta_n;
         const int phase = (int)(meta & 63);
         const bool done = child_meta(meta, pos, qc, nt, mm, meta_n, mm);
         meta = meta_n;
         pos = fw ? pos + 1 : pos - 1;
         if (done) {
            if (rec_y[phase] >> 8) {
               const int score = (int)(meta >> META_SCORE) & 7;
               uint64_t al[4];
               hit_align(mm, score, q, m, al);
               int64_t row[10] = {fp, rp, sz, m, score, (int64_t)al[0], (int64_t)al[1], (int64_t)al[2], (int64_t)al[3], 0};
               res.insert(res.end(), row, row + 10);
               have = false;
            } else {
               const uint32_t sc = (meta >> META_SCORE) & 7;
               meta = rec_x[phase] | (sc << META_SCORE) | (sc << META_SREF);
               pos = (int)rec_y[phase];
            }
         }
      }
   }
   int64_t nh = (int64_t)res.size() / 10;
   int64_t *o = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nh > 0 ? nh * 10 : 1));
   for (int64_t i = 0; i < nh; i++) { memcpy(o + 10 * i, &res[10 * i], 80); o[10 * i + 9] = n_ext; }
   if (nh == 0) o[0] = n_ext;
   *out = o;
   return nh;
}

extern "C" void emu_free(void *p) { free(p); }
