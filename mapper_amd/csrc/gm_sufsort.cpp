// gm_sufsort.cpp -- host suffix sorter for index construction (one-off, off the query hot path).
//
// The reference builds its suffix array with the vendored divsufsort over the text with symbols
// shifted so that the separator sorts first (src/index_sar.c:46-54).  A suffix array is unique
// given the text and the symbol order, so any correct sorter reproduces the reference's .sar
// byte for byte.  This one is prefix doubling (Manber-Myers / Larsson-Sadakane): an initial
// sort on the first 21 symbols packed into 63 bits, then rounds that re-sort only the groups
// still tied, by the rank of the suffix h positions further on.  Buckets and groups are
// independent, so both stages run on a small thread pool.
#include <algorithm>
#include <atomic>
#include <cstring>
#include <thread>
#include <vector>
#include "gm_host.hpp"

namespace {

template <class F>
void parallel_for(int64_t n_items, int threads, F fn)
{
   if (threads <= 1 || n_items <= 1) { for (int64_t i = 0; i < n_items; i++) fn(i); return; }
   std::atomic<int64_t> next(0);
   std::vector<std::thread> pool;
   for (int t = 0; t < threads; t++)
      pool.emplace_back([&]() { for (int64_t i; (i = next.fetch_add(1)) < n_items;) fn(i); });
   for (auto &th : pool) th.join();
}

constexpr int K0 = 21;          // symbols in the initial key, 3 bits each (value + 1; 0 = past the end)
constexpr int BUCKET_SYMS = 6;  // leading symbols used for the bucket split
constexpr int BUCKET_BITS = 3 * BUCKET_SYMS;

template <class IDX>
struct KeyIdx { uint64_t key; IDX idx; };

template <class IDX>
int suffix_sort_impl(const uint8_t *data, int64_t n, int64_t *sa_out, int threads)
{
   using KI = KeyIdx<IDX>;
   std::vector<IDX> sa(n), rank(n);
   {
      // ---- initial sort on 21-symbol keys --------------------------------------------------
      std::vector<KI> cur(n), tmp(n);
      uint64_t key = 0;
      for (int64_t i = n - 1; i >= 0; i--) {
         key = (key >> 3) | ((uint64_t)(data[i] + 1) << 60);
         cur[i].key = key; cur[i].idx = (IDX)i;
      }
      const int64_t NB = (int64_t)1 << BUCKET_BITS;
      std::vector<int64_t> cnt(NB + 1, 0);
      for (int64_t i = 0; i < n; i++) cnt[(cur[i].key >> (63 - BUCKET_BITS)) + 1]++;
      for (int64_t b = 0; b < NB; b++) cnt[b + 1] += cnt[b];
      {
         std::vector<int64_t> pos(cnt.begin(), cnt.end() - 1);
         for (int64_t i = 0; i < n; i++) tmp[pos[cur[i].key >> (63 - BUCKET_BITS)]++] = cur[i];
      }
      std::vector<int64_t> nonempty;
      for (int64_t b = 0; b < NB; b++) if (cnt[b + 1] - cnt[b] > 1) nonempty.push_back(b);
      // big buckets first for balance
      std::sort(nonempty.begin(), nonempty.end(), [&](int64_t a, int64_t b) {
         return cnt[a + 1] - cnt[a] > cnt[b + 1] - cnt[b]; });
      parallel_for((int64_t)nonempty.size(), threads, [&](int64_t k) {
         int64_t b = nonempty[k];
         std::sort(tmp.begin() + cnt[b], tmp.begin() + cnt[b + 1], [](const KI &x, const KI &y) {
            return x.key < y.key || (x.key == y.key && x.idx < y.idx); });
      });
      // ranks = index of the first member of each equal-key group
      int64_t g = 0;
      for (int64_t j = 0; j < n; j++) {
         if (j > 0 && tmp[j].key != tmp[j - 1].key) g = j;
         sa[j] = tmp[j].idx;
         rank[tmp[j].idx] = (IDX)g;
      }
   }

   // ---- doubling rounds over the groups that are still tied -----------------------------------
   struct Grp { int64_t beg, len; };
   std::vector<Grp> groups;
   {
      int64_t j = 0;
      while (j < n) {
         int64_t e = j + 1;
         while (e < n && rank[sa[e]] == rank[sa[j]]) e++;
         if (e - j > 1) groups.push_back({j, e - j});
         j = e;
      }
   }
   std::vector<uint8_t> split;   // split[j] = 1: a new group starts at j (filled per round)
   for (int64_t h = K0; !groups.empty(); h *= 2) {
      split.assign(n, 0);
      // phase A: order each group by the rank of the suffix h further on (ranks are read-only)
      parallel_for((int64_t)groups.size(), threads, [&](int64_t gi) {
         const Grp g = groups[gi];
         std::vector<KI> buf(g.len);
         for (int64_t t = 0; t < g.len; t++) {
            IDX i = sa[g.beg + t];
            int64_t nx = (int64_t)i + h;
            buf[t].key = nx < n ? (uint64_t)rank[nx] + 1 : 0;   // running off the end sorts first
            buf[t].idx = i;
         }
         std::sort(buf.begin(), buf.end(), [](const KI &x, const KI &y) {
            return x.key < y.key || (x.key == y.key && x.idx < y.idx); });
         for (int64_t t = 0; t < g.len; t++) {
            sa[g.beg + t] = buf[t].idx;
            if (t > 0 && buf[t].key != buf[t - 1].key) split[g.beg + t] = 1;
         }
      });
      // phase B: new ranks and the next round's groups
      std::vector<std::vector<Grp>> next(groups.size());
      parallel_for((int64_t)groups.size(), threads, [&](int64_t gi) {
         const Grp g = groups[gi];
         int64_t s = g.beg;
         for (int64_t j = g.beg; j <= g.beg + g.len; j++) {
            if (j == g.beg + g.len || (j > g.beg && split[j])) {
               if (j - s > 1) next[gi].push_back({s, j - s});
               s = j;
            }
            if (j < g.beg + g.len) rank[sa[j]] = (IDX)s;
         }
      });
      std::vector<Grp> merged;
      for (auto &v : next) merged.insert(merged.end(), v.begin(), v.end());
      groups.swap(merged);
      if (h > 2 * n && !groups.empty()) return -1;   // ties left after comparing more than n letters: a logic error, never a partial order handed on
   }
   for (int64_t j = 0; j < n; j++) sa_out[j] = (int64_t)sa[j];
   return 0;
}

}  // namespace

int gm_suffix_sort(const uint8_t *data, int64_t n, int64_t *sa, int threads)
{
   if (n <= 0) return 0;
   if (threads < 1) threads = 1;
   if (n < ((int64_t)1 << 31)) return suffix_sort_impl<uint32_t>(data, n, sa, threads);
   return suffix_sort_impl<uint64_t>(data, n, sa, threads);
}
