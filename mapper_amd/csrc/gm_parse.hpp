// gm_parse.hpp -- the two ends of mapper() (src/mapper.c:3-47) on the device, so that the host only moves bytes:
//
//   in:  a record-aligned block of a FASTQ / FASTA / raw read file, as the host cut it from the file (the chunk of
//        io_stream_read_seq, src/io.c:113-227), is parsed where it lands: k_line_count + k_line_write find the ends of
//        its lines, k_parse_records applies the rules of seqread_parse_fastq / _fasta (src/seqread.c:119-164) and of
//        io_stream_read_seq per record (the sequence is line 1 of 4, 1 of 2 or 0 of 1; a line's content ends at its
//        newline or at a NUL, as the reference's C strings do; FASTQ: the quality line must be as long as the
//        sequence, :47-49; the tag is line 0 without its first character, or the record's line number for raw
//        files, src/io.c:171-174) and leaves, per read, where its letters begin and end in the block: k_encode reads
//        them from there.
//   out: k_format_len + k_format_write turn the sorted keys of the located hits into the lines mapper() prints,
//            tag \t name:pos:strand \t score \n
//        with txt_pos_to_str's rule (src/index_txt.c:480-511: strand = pos >= n/2, then pos = n-2-pos; sequence by
//        seq_bisect, :770-787; 1-based position inside the sequence), two passes around one scan of the lines' lengths.
#pragma once
#include "gm_kernels.hpp"

namespace gm {

constexpr int LINE_ROW = 256 * 16;            // bytes a block looks at per round: 16 per thread
constexpr int LINE_ROUNDS = 4;
constexpr int LINE_TILE = LINE_ROW * LINE_ROUNDS;
constexpr int SEEN_QUAL = MAX_READ + 3;       // k_parse_records: a FASTQ record whose quality line has another length than its sequence

// 0x80 in every byte of w that equals the byte `pat` is made of.
__device__ __forceinline__ uint32_t eq_bytes(uint32_t w, uint32_t pat)
{
   const uint32_t x = w ^ pat;
   const uint32_t t = (x & 0x7f7f7f7fu) + 0x7f7f7f7fu;
   return ~(t | x | 0x7f7f7f7fu);
}

// bit k of the result = byte k of the 16 bytes is a newline; nul |= any byte is 0
__device__ __forceinline__ uint32_t newline_mask(const uint8_t *raw, int64_t at, int64_t n, uint32_t &nul)
{
   uint32_t m = 0;
   if (at + 16 <= n) {
      const uint4 v = *reinterpret_cast<const uint4 *>(raw + at);
      const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int j = 0; j < 4; j++) {
         const uint32_t e = eq_bytes(w[j], 0x0a0a0a0au);
         m |= ((((e >> 7) & 0x01010101u) * 0x01020408u) >> 24 & 0xfu) << (4 * j);
         nul |= eq_bytes(w[j], 0u);
      }
   } else {
      for (int k = 0; k < 16 && at + k < n; k++) {
         const uint8_t c = raw[at + k];
         m |= (uint32_t)(c == '\n') << k;
         nul |= (uint32_t)(c == 0);
      }
   }
   return m;
}

template <int THREADS>
__device__ __forceinline__ uint32_t block_sum(uint32_t v, uint32_t *s_part /* [THREADS / 64] */)
{
   for (int o = 32; o > 0; o >>= 1) v += (uint32_t)__shfl_xor((int)v, o);
   if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = v;
   __syncthreads();
   uint32_t t = 0;
   for (int k = 0; k < THREADS / 64; k++) t += s_part[k];
   __syncthreads();
   return t;
}

// Newlines per tile of LINE_TILE bytes; *nul_flag = 1 when the block holds a NUL byte anywhere.
__global__ void __launch_bounds__(256) k_line_count(const uint8_t *raw, int64_t n, uint32_t *tile_cnt, uint32_t *nul_flag)
{
   __shared__ uint32_t s_part[4];
   const int64_t tile = blockIdx.x * (int64_t)LINE_TILE;
   uint32_t cnt = 0, nul = 0;
#pragma unroll
   for (int j = 0; j < LINE_ROUNDS; j++) {
      const int64_t at = tile + j * (int64_t)LINE_ROW + threadIdx.x * 16;
      if (at < n) cnt += (uint32_t)__popc(newline_mask(raw, at, n, nul));
   }
   const uint32_t total = block_sum<256>(cnt, s_part);
   if (threadIdx.x == 0) tile_cnt[blockIdx.x] = total;
   if (nul) *nul_flag = 1;
}

// The positions of the newlines, ascending: nl_pos[tile_off[tile] ..].
__global__ void __launch_bounds__(256) k_line_write(const uint8_t *raw, int64_t n, const uint32_t *tile_off, uint32_t *nl_pos, uint32_t cap)
{
   __shared__ uint32_t s_wave[4];
   const int64_t tile = blockIdx.x * (int64_t)LINE_TILE;
   const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
   uint32_t base = tile_off[blockIdx.x];
   for (int j = 0; j < LINE_ROUNDS; j++) {
      const int64_t at = tile + j * (int64_t)LINE_ROW + threadIdx.x * 16;
      uint32_t nul = 0;
      uint32_t m = at < n ? newline_mask(raw, at, n, nul) : 0u;
      const uint32_t c = (uint32_t)__popc(m);
      uint32_t incl = c;                                   // inclusive scan over the wave, then over the four waves
      for (int o = 1; o < 64; o <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)incl, o); if (lane >= o) incl += t; }
      if (lane == 63) s_wave[wid] = incl;
      __syncthreads();
      uint32_t before = 0, all = 0;
      for (int k = 0; k < 4; k++) { if (k < wid) before += s_wave[k]; all += s_wave[k]; }
      uint32_t w = base + before + incl - c;
      while (m) { const int k = __ffs((int)m) - 1; m &= m - 1; if (w < cap) nl_pos[w] = (uint32_t)(at + k); w++; }
      base += all;
      __syncthreads();
   }
}

struct ParseArgs {
   const uint8_t *raw;
   int64_t n;                 // bytes of the block
   const uint32_t *nl_pos;
   const uint32_t *n_nl_dev;  // newlines of the block (the total of the scan of the tiles' counts)
   int64_t nl_cap;            // entries nl_pos holds (the host's count of the block's lines + 1)
   int lines_per_rec;         // 4 FASTQ, 2 FASTA, 1 raw
   int64_t n_rec;
   const uint32_t *nul_flag;
   int64_t *beg, *end;        // per read: its letters are raw[beg, end)
   uint16_t *read_len;
   uint8_t *seen;
   uint32_t *tag_off, *tag_len;      // lines_per_rec > 1
};

__device__ __forceinline__ void parse_line(const ParseArgs &P, bool any_nul, int64_t n_nl, int64_t i, int64_t &b, int64_t &e)
{
   if (i > n_nl) { b = e = P.n; return; }                  // (no such line: the host counted the block's lines otherwise)
   b = i > 0 ? (int64_t)P.nl_pos[i - 1] + 1 : 0;
   e = i < n_nl ? (int64_t)P.nl_pos[i] : P.n;              // (the last line of a file may come without a newline)
   if (any_nul)
      for (int64_t k = b; k < e; k++) if (P.raw[k] == 0) { e = k; break; }
}

__global__ void __launch_bounds__(256) k_parse_records(ParseArgs P)
{
   __shared__ uint32_t s_seen[SEEN_BYTES];
   for (int i = threadIdx.x; i < SEEN_BYTES; i += blockDim.x) s_seen[i] = 0;
   __syncthreads();
   const bool any_nul = *P.nul_flag != 0;
   const int L = P.lines_per_rec;
   const int64_t n_nl = min((int64_t)*P.n_nl_dev, P.nl_cap);
   for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < P.n_rec; r += (int64_t)gridDim.x * blockDim.x) {
      int64_t b, e;
      parse_line(P, any_nul, n_nl, L * r + (L == 1 ? 0 : 1), b, e);
      const int64_t len = e - b;
      P.beg[r] = b; P.end[r] = e;
      P.read_len[r] = (uint16_t)(len > 0xffff ? 0xffff : len);
      s_seen[len > MAX_READ ? SEEN_LONGER : (int)len] = 1;
      if (L > 1) {
         int64_t tb, te;
         parse_line(P, any_nul, n_nl, L * r, tb, te);
         const int64_t tl = te - tb;
         P.tag_off[r] = (uint32_t)(tb + (tl > 0 ? 1 : 0));                    // lines[0] + 1 (src/seqread.c:130, :155)
         P.tag_len[r] = (uint32_t)(tl > 0 ? tl - 1 : 0);
      }
      if (L == 4) {
         int64_t qb, qe;
         parse_line(P, any_nul, n_nl, L * r + 3, qb, qe);
         if (qe - qb != len) s_seen[SEEN_QUAL] = 1;                           // src/seqread.c:47-49
      }
   }
   __syncthreads();
   for (int i = threadIdx.x; i < SEEN_BYTES; i += blockDim.x)
      if (s_seen[i]) P.seen[i] = 1;
}

// ------------------------------------------------------------------------------------------------
// Hit lines
// ------------------------------------------------------------------------------------------------
struct FormatArgs {
   const unsigned long long *keys;      // sorted: read << (pos_bits + 4) | text position << 4 | score
   int64_t n_keys;
   int pos_bits;
   int64_t n;                           // text length
   const int64_t *seq_beg;
   int seq_cnt;
   const uint8_t *names;                // the sequences' names back to back, name i = names[name_off[i], name_off[i + 1])
   const uint32_t *name_off;
   const uint8_t *raw;                  // the block the batch was parsed from (the tags lie in it)
   const uint32_t *tag_off, *tag_len;   // nullptr: tags are line numbers, first_line + read
   int64_t first_line;
   int64_t *line_len;                   // [n_keys + 1] lengths, then (scanned in place) offsets
   uint8_t *out;
};

__device__ __forceinline__ int dec_digits(uint64_t v)
{
   int d = 1;
   while (v >= 10) { v /= 10; d++; }
   return d;
}

__device__ __forceinline__ uint8_t *put_dec(uint8_t *p, uint64_t v)
{
   const int d = dec_digits(v);
   for (int k = d - 1; k >= 0; k--) { p[k] = (uint8_t)('0' + v % 10); v /= 10; }
   return p + d;
}

__device__ __forceinline__ void key_fields(const FormatArgs &A, unsigned long long k, int64_t &read, int64_t &seq, int64_t &seqpos, bool &minus, int &score)
{
   read = (int64_t)(k >> (A.pos_bits + 4));
   const int64_t pos = (int64_t)((k >> 4) & ((1ull << A.pos_bits) - 1ull));
   score = (int)(k & 15);
   minus = pos >= A.n / 2;                                           // src/index_txt.c:492-496
   const int64_t f = minus ? A.n - 2 - pos : pos;
   int64_t a = 0, e = A.seq_cnt;                                      // seq_bisect, :770-787
   while (e - a >= 2) { const int64_t mid = (a + e) / 2; if (f < A.seq_beg[mid]) e = mid; else a = mid; }
   seq = a;
   seqpos = f - A.seq_beg[a] + 1;                                     // :505
}

__global__ void __launch_bounds__(256) k_format_len(FormatArgs A)
{
   const int64_t t = blockIdx.x * (int64_t)256 + threadIdx.x;
   if (t > A.n_keys) return;
   if (t == A.n_keys) { A.line_len[t] = 0; return; }
   int64_t read, seq, seqpos;
   bool minus;
   int score;
   key_fields(A, A.keys[t], read, seq, seqpos, minus, score);
   const int64_t tl = A.tag_len ? (int64_t)A.tag_len[read] : (int64_t)dec_digits((uint64_t)(A.first_line + read));
   A.line_len[t] = tl + 1 + (int64_t)(A.name_off[seq + 1] - A.name_off[seq]) + 1 + dec_digits((uint64_t)seqpos) + 3 + dec_digits((uint64_t)score) + 1;
}

__global__ void __launch_bounds__(256) k_format_write(FormatArgs A)
{
   const int64_t t = blockIdx.x * (int64_t)256 + threadIdx.x;
   if (t >= A.n_keys) return;
   int64_t read, seq, seqpos;
   bool minus;
   int score;
   key_fields(A, A.keys[t], read, seq, seqpos, minus, score);
   uint8_t *p = A.out + A.line_len[t];                               // (scanned: the line's offset)
   if (A.tag_len) {
      const uint8_t *tag = A.raw + A.tag_off[read];
      const uint32_t tl = A.tag_len[read];
      for (uint32_t k = 0; k < tl; k++) p[k] = tag[k];
      p += tl;
   } else {
      p = put_dec(p, (uint64_t)(A.first_line + read));
   }
   *p++ = '\t';
   const uint8_t *nm = A.names + A.name_off[seq];
   const uint32_t nl = A.name_off[seq + 1] - A.name_off[seq];
   for (uint32_t k = 0; k < nl; k++) p[k] = nm[k];
   p += nl;
   *p++ = ':';
   p = put_dec(p, (uint64_t)seqpos);
   *p++ = ':';
   *p++ = minus ? '-' : '+';
   *p++ = '\t';
   p = put_dec(p, (uint64_t)score);
   *p++ = '\n';
}

}  // namespace gm
