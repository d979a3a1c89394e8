// Short rows outside the window path (wide column hull, few products): count-only symbolic pass and a
// numeric pass that builds its own sorted pattern, so no column list travels between the two.
// Part of the gfx950 SpGEMM kernels; included through kernels.h.
#pragma once
#include "kernels_numeric.h"

namespace sg {

// A row takes this path when it has at most SH_MAX_NPROD products and at most SH_MAX_ALEN entries
// in its A row (the whole A row then sits in one wavefront, one entry per lane).
constexpr int SH_STEPS = SH_MAX_NPROD / 64; // steps of 64 products a row can take
// per-wavefront walk scratch: st[64] u32 | p0s[64] u64 | marks[64 * steps] u8 | avs[64] f64 (numeric only);
// steps: steps of 64 products the rows of the launch can take
__host__ __device__ constexpr uint32_t sh_scratch(bool numeric, uint32_t steps) {
  return 256u + 512u + 64u * steps + (numeric ? 512u : 0u);
}

// Monotone map of the columns of B onto [0, nnz(B)]: F(c) = number of entries of B in columns < c,
// piecewise linear over CDF_BINS equal column ranges (cdf[b] = entries in the bins before b, rounded to
// single precision: the map only has to be monotone - rounding is - and to tell a few hundred home
// slots apart, and a row pays for six evaluations and a division of it, which cost a tenth of the
// kernel's vector instructions in double precision).  A row's columns, drawn from B's rows, are spread evenly by F whatever
// the popularity of B's columns (R-MAT hubs), which keeps the clusters of the sorted table short.
constexpr uint32_t CDF_BINS = 2048;
struct ColMap {
  const float *__restrict__ cdf;  // CDF_BINS + 1 entries, then 1.0 if the entries of B are spread evenly over its columns (k_col_cdf)
  uint32_t shift;                 // bin of column c = c >> shift
  float inv_width;                // 2^-shift
  const uint4 *__restrict__ prec; // B rows as padded {column, 0, value} records (k_prec_fill), or null
  const uint32_t *__restrict__ pcol; // the columns of those records alone, at the same positions, or null
};
// value of the map at column c, given the prefix counts lo, hi at the two ends of c's bin
__device__ __forceinline__ float colmap_interp(const ColMap &cm, uint32_t c, float lo, float hi) {
  const uint32_t b = c >> cm.shift;
  const float frac = (float)(c - (b << cm.shift)) * cm.inv_width;
  return lo + (hi - lo) * frac;
}
// lin (wave-uniform): the entries of B are spread evenly over its columns: the column itself is the map, no table read
__device__ __forceinline__ float colmap_eval(const ColMap &cm, uint32_t c, bool lin) {
  if (lin) return (float)c;
  const uint32_t b = c >> cm.shift;
  return colmap_interp(cm, c, cm.cdf[b], cm.cdf[b + 1]);
}

// inclusive max-scan over the 64 lanes (0 is the identity)
__device__ __forceinline__ uint32_t wave_incl_max(uint32_t v) {
  v = max(v, dpp_move<0x111>(0u, v));
  v = max(v, dpp_move<0x112>(0u, v));
  v = max(v, dpp_move<0x114>(0u, v));
  v = max(v, dpp_move<0x118>(0u, v));
  v = max(v, dpp_move<0x142, 0xA>(0u, v));
  v = max(v, dpp_move<0x143, 0xC>(0u, v));
  return v;
}

// A wavefront takes `rpw` consecutive rows of its bin and pipelines them: while row i is worked on in
// LDS, the B entries of row i + 1 are in flight (their addresses need the A entries and the B row
// bounds, which were fetched while row i - 1 was worked on) and the A entries of row i + 2 are being
// fetched.  Without this a wavefront has loads in flight for a third of its life, and the kernels
// are bound by memory latency at the occupancy their LDS tables allow.
//
// Stage A (ShortFetch): lane t holds A entry t of the row: its value, start and length of B row k.
struct ShortFetch {
  uint32_t row, len, alen, n, cmin, cmax; // len: of the lane's B row; alen: of the A row
  uint64_t p0, off;
  double av;
};
template <bool NUMERIC>
__device__ __forceinline__ void short_fetch(const Csr &A, const Csr &B, const RowMeta &m, const ColMap &cm,
                                            const uint64_t *c_ptr, const uint32_t *rows, uint32_t idx, uint32_t lane,
                                            ShortFetch &f) {
  f.row = scalar_view(rows)[idx];
  const uint64_t a0 = scalar_view(A.ptr)[f.row], a1 = scalar_view(A.ptr)[f.row + 1];
  if constexpr (NUMERIC) {
    f.n = scalar_view((const uint32_t *)m.nnz)[f.row];
    f.off = scalar_view(c_ptr)[f.row];
    f.cmin = scalar_view(m.cmin)[f.row];
    f.cmax = scalar_view(m.cmax)[f.row];
  }
  const uint64_t ka = a0 + lane;
  f.alen = (uint32_t)(a1 - a0);
  f.len = 0;
  f.p0 = 0;
  f.av = 0.0;
  if (ka < a1) { // start and length of the B row: left per A entry by the analysis, read in a stream
    if constexpr (NUMERIC) f.p0 = cm.prec ? m.a_pk[ka] : m.a_p0[ka]; // wave-uniform choices
    else f.p0 = cm.pcol ? m.a_pk[ka] : m.a_p0[ka];
    f.len = m.a_len[ka];
    if constexpr (NUMERIC) f.av = A.val[ka];
  }
}

// Stage B (ShortSet): the products of the row, 64 per step, product q of the row (A order, then B
// row order) in lane q % 64 of step q / 64.  The exclusive prefix sums of the B row lengths go to
// LDS and a byte array marks, at each entry's first product, the entry (t + 1); a step reads its 64
// marks, and a running max over the lanes (and over the steps) gives every product its entry: no
// search.  Where the B rows are all short (up to 4, 8 or 16 entries) and the A row has few enough
// entries, every A entry gets a slot of 4, 8 or 16 lanes instead (product number = entry * slot +
// position, still in the CPU loop's order, with gaps): no prefix sums, marks or running max.
// qmax: product numbers the caller can take (a power of two).
// The loads of the B entries are issued here and consumed by stage C of the same row, one
// row later in program order.
// LDS scratch per wavefront: sh_scratch(NUMERIC, STEPS) bytes; the A values stay in avs for stage C,
// which reads them before the next row is expanded.
// STEPS: steps of 64 products the rows of the launch can take.
template <bool NUMERIC, int STEPS> struct ShortSet {
  uint32_t c[STEPS], t[STEPS];
  double bv[STEPS];
  uint32_t row, total, n, cmin, cmax;
  uint64_t off;
  float f[4]; // column map around cmin and cmax (numeric)
};
template <bool NUMERIC, int STEPS>
__device__ __forceinline__ void short_expand(const Csr &B, const ColMap &cm, const ShortFetch &f, uint32_t lane,
                                             unsigned char *scratch, uint32_t qmax, ShortSet<NUMERIC, STEPS> &s) {
  static_assert(STEPS == 4 || STEPS == 8, "marks are cleared with one 4- or 8-byte store per lane");
  uint32_t *st = reinterpret_cast<uint32_t *>(scratch);
  uint64_t *p0s = reinterpret_cast<uint64_t *>(scratch + 256);
  uint8_t *marks = scratch + 768;
  double *avs = reinterpret_cast<double *>(scratch + 768 + 64 * STEPS);
  s.row = f.row;
  if constexpr (NUMERIC) {
    s.n = f.n; s.off = f.off; s.cmin = f.cmin; s.cmax = f.cmax;
    const uint32_t b0 = f.cmin >> cm.shift, b1 = f.cmax >> cm.shift;
    s.f[0] = cm.cdf[b0]; s.f[1] = cm.cdf[b0 + 1]; s.f[2] = cm.cdf[b1]; s.f[3] = cm.cdf[b1 + 1];
  }
  lds_fence(); // the scratch of the row before has been read
  const uint32_t incl = group_incl_scan<64>(f.len);
  s.total = min(group_last<64>(incl), (uint32_t)STEPS * 64u);
  // slot per A entry: 1 << sh lanes, when every B row fits and the numbers stay below qmax
  const uint32_t sh = __any(f.len > 8u) ? 4u : __any(f.len > 4u) ? 3u : 2u;
  const bool slots = !__any(f.len > 16u) && (f.alen << sh) <= min(qmax, 64u * (uint32_t)STEPS);
  uint32_t nsteps;
  if (slots) {
    st[lane] = f.len;
    nsteps = ((f.alen << sh) + 63u) >> 6;
  } else {
    if constexpr (STEPS == 8) reinterpret_cast<uint64_t *>(marks)[lane] = 0ull; // 64 * STEPS bytes
    else reinterpret_cast<uint32_t *>(marks)[lane] = 0u;
    lds_fence();
    st[lane] = incl - f.len;
    if (f.len && incl - f.len < 64u * STEPS) marks[incl - f.len] = (uint8_t)(lane + 1u);
    nsteps = (s.total + 63u) >> 6;
  }
  p0s[lane] = f.p0;
  if constexpr (NUMERIC) avs[lane] = f.av;
  lds_fence();
  uint32_t carry = 1u;
#pragma unroll
  for (int u = 0; u < STEPS; ++u) {
    s.c[u] = EMPTY_KEY;
    s.bv[u] = 0.0;
    s.t[u] = 0;
    if ((uint32_t)u < nsteps) { // wave-uniform
      const uint32_t q = (uint32_t)u * 64u + lane;
      bool valid;
      uint32_t t;
      uint64_t p;
      if (slots) { // wave-uniform
        t = q >> sh;
        const uint32_t j = q & ((1u << sh) - 1u);
        valid = j < st[t]; // (t < 64; the lanes past the A row hold length 0)
        p = p0s[t] + j;
      } else {
        valid = q < s.total;
        const uint32_t mk = valid ? (uint32_t)marks[q] : 0u;
        const uint32_t tt = max(wave_incl_max(mk), carry);
        carry = (uint32_t)__builtin_amdgcn_readlane((int)tt, 63);
        t = tt - 1u;
        p = p0s[t] + (q - st[t]);
      }
      s.t[u] = t;
      if constexpr (NUMERIC) {
        if (cm.prec) { // the B rows as padded records: one 16-byte load, len / 8 lines per row
          if (valid) {
            const uint4 rec = cm.prec[p];
            s.c[u] = rec.x;
            s.bv[u] = __longlong_as_double((long long)(((unsigned long long)rec.w << 32) | rec.z));
          }
        } else if (valid) {
          s.c[u] = B.col[p];
          s.bv[u] = B.val[p];
        }
      } else {
        if (valid) s.c[u] = (cm.pcol ? cm.pcol : B.col)[p];
      }
    }
  }
}

// ---- symbolic: number of distinct columns of the row, nothing else --------------------------------
// LDS per wavefront: keys[S] u32 (S = 1 << log2s >= 2 * products) | walk scratch.
template <int STEPS>
__global__ void __launch_bounds__(256)
    k_sym_count(const uint32_t *__restrict__ rows, uint32_t nrows, Csr A, Csr B, RowMeta m, uint32_t log2s,
                uint32_t wpb, uint32_t rpw, const uint32_t *__restrict__ pcol) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint64_t first64 = ((uint64_t)xcd_remap(blockIdx.x, gridDim.x) * wpb + w) * rpw;
  if (w >= wpb || first64 >= nrows) return;
  const uint32_t first = (uint32_t)first64, last = (uint32_t)min((uint64_t)nrows, first64 + rpw);
  const uint32_t S = 1u << log2s, mask = S - 1u;
  const size_t region = (size_t)S * 4u + sh_scratch(false, STEPS);
  uint32_t *keys = reinterpret_cast<uint32_t *>(smem + w * region);
  unsigned char *scratch = smem + w * region + (size_t)S * 4u;
  const ColMap none{nullptr, 0, 0.0f, nullptr, pcol};
  ShortFetch fa;
  ShortSet<false, STEPS> cur, nxt;
  short_fetch<false>(A, B, m, none, nullptr, rows, first, lane, fa);
  short_expand<false, STEPS>(B, none, fa, lane, scratch, 64u * STEPS, cur);
  if (first + 1u < last) short_fetch<false>(A, B, m, none, nullptr, rows, first + 1u, lane, fa);
  for (uint32_t i = first; i < last; ++i) {
    if (i + 1u < last) {
      short_expand<false, STEPS>(B, none, fa, lane, scratch, 64u * STEPS, nxt);
      if (i + 2u < last) short_fetch<false>(A, B, m, none, nullptr, rows, i + 2u, lane, fa);
    }
    for (uint32_t j = lane * 4u; j < S; j += 256u)
      *reinterpret_cast<uint4 *>(&keys[j]) = make_uint4(EMPTY_KEY, EMPTY_KEY, EMPTY_KEY, EMPTY_KEY);
    lds_fence();
    // The probe chains of a lane's products advance together: the compare-and-swaps of a round are
    // issued back to back and their results read afterwards, without a branch in between (a lane
    // that is done swaps "empty" for "empty", which changes no slot; it starts from a slot of its
    // own so that the idle lanes do not meet on one address).
    uint32_t mine = 0, h[STEPS], c[STEPS];
#pragma unroll
    for (int u = 0; u < STEPS; ++u) {
      c[u] = cur.c[u];
      h[u] = c[u] != EMPTY_KEY ? hash_slot(c[u], log2s) : lane;
    }
    for (uint32_t round = 0;; ++round) {
      uint32_t old[STEPS];
#pragma unroll
      for (int u = 0; u < STEPS; ++u) old[u] = atomicCAS(&keys[h[u]], EMPTY_KEY, c[u]);
      bool open = false;
      uint32_t left = 0; // (wave-uniform) products whose chain goes on
      uint64_t bal[STEPS];
#pragma unroll
      for (int u = 0; u < STEPS; ++u) {
        mine += (old[u] == EMPTY_KEY && c[u] != EMPTY_KEY) ? 1u : 0u;
        c[u] = (old[u] == EMPTY_KEY || old[u] == c[u]) ? EMPTY_KEY : c[u];
        h[u] = c[u] != EMPTY_KEY ? (h[u] + 1u) & mask : lane; // the table is at most half full: an empty slot comes
        open |= c[u] != EMPTY_KEY;
        bal[u] = __ballot(c[u] != EMPTY_KEY);
        left += (uint32_t)__popcll(bal[u]);
      }
      if (left == 0u) break;
      // Most products find their slot at once; the few that go on (a quarter of the table is in use) are
      // handed one to a lane, through the walk scratch, which is free now: the rounds that follow cost one
      // compare-and-swap per round instead of one per step of the row.
      if (round == 0u && left <= 64u) {
        uint32_t *lst = reinterpret_cast<uint32_t *>(scratch); // {column, slot} pairs
        uint32_t before = 0;
#pragma unroll
        for (int u = 0; u < STEPS; ++u) {
          if (c[u] != EMPTY_KEY) {
            const uint32_t at = before + (uint32_t)__popcll(bal[u] & ((1ull << lane) - 1ull));
            lst[2u * at] = c[u];
            lst[2u * at + 1u] = h[u];
          }
          before += (uint32_t)__popcll(bal[u]);
        }
        lds_fence();
        uint32_t cc = EMPTY_KEY, hh = lane;
        if (lane < left) { cc = lst[2u * lane]; hh = lst[2u * lane + 1u]; }
        for (;;) {
          const uint32_t o1 = atomicCAS(&keys[hh], EMPTY_KEY, cc);
          mine += (o1 == EMPTY_KEY && cc != EMPTY_KEY) ? 1u : 0u;
          cc = (o1 == EMPTY_KEY || o1 == cc) ? EMPTY_KEY : cc;
          hh = cc != EMPTY_KEY ? (hh + 1u) & mask : lane;
          if (!__any(cc != EMPTY_KEY)) break;
        }
        break;
      }
    }
    const uint32_t n = group_sum<64>(mine);
    if (lane == 0) {
      m.nnz[cur.row] = n;
      m.num_key[cur.row] = (uint8_t)num_key_of(m.sym_key[cur.row], n, m.hash_num_max);
    }
    lds_fence();
    cur = nxt;
  }
}

// ---- numeric: expand, sort by (column, product number), sum runs, write -------------------------------
// PM = 1 << log2pm >= products of the row.  LDS per wavefront:
//   tab[3 PM] u64 | prodv[PM] f64 | walk scratch.
// 1. Every product q (numbered in the CPU loop's order: A entry after A entry, B row order inside)
//    puts the entry (column << 32 | q) into tab[] by sorted linear probing: the home slot is a
//    monotone function of the column (ColMap, scaled to [0, 2 * products)); a probe leaves the
//    smaller of {slot, carried entry} in the slot (64-bit atomic min) and carries the larger on.
//    A slot only ever decreases and an empty slot is never crossed, so at the end every run of
//    occupied slots is ascending and the runs are ordered among themselves: the table holds the
//    row's products sorted by column and, inside a column, in the order the CPU adds them.  No
//    wrap-around: at most `products` entries lie past the last home slot.  The product's value goes
//    to prodv[q].
// 2. The occupied slots are moved to the front of the table, keeping their order.  The first entry
//    of a column sums the column's entries one after the other (they are adjacent) - the CPU's
//    order of additions, bit for bit - and is written at the position given by the number of column
//    starts before it.
// K32: the table entry is 32 bits, column << log2pm | product number, for a B with fewer than
// 2^(32 - log2pm) columns: half the table, 32-bit atomics and compares.
// rows of up to 256 products go to the kernel with 4 steps, longer ones to the one with SH_STEPS
__host__ __device__ inline uint32_t num_short_steps(uint32_t log2pm) { return log2pm <= 8 ? 4u : (uint32_t)SH_STEPS; }
__host__ __device__ inline size_t num_short_region(uint32_t log2pm, bool k32) {
  return ((size_t)(k32 ? 20u : 32u) << log2pm) + sh_scratch(true, num_short_steps(log2pm));
}
template <int STEPS, bool K32>
__global__ void __launch_bounds__(256)
    k_num_short(const uint32_t *__restrict__ rows, uint32_t nrows, Csr A, Csr B, RowMeta m, NumOut o, ColMap cm,
                uint32_t log2pm, uint32_t wpb, uint32_t rpw) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  using key_t = typename std::conditional<K32, uint32_t, unsigned long long>::type;
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // Wavefront g of the launch's nw takes the rows g, g + nw, g + 2 nw, ... of the bin (at most rpw: the
  // grid is sized so): the rows in work at one time are neighbours in the bin, and so are the pieces of
  // C being written (with rpw rows in a block per wavefront the writes of a moment were spread over
  // rpw times as much of C, which cost a tenth of the kernel's time).
  const uint64_t g64 = (uint64_t)xcd_remap(blockIdx.x, gridDim.x) * wpb + w;
  const uint64_t nw = (uint64_t)gridDim.x * wpb;
  if (w >= wpb || g64 >= nrows) return;
  const uint32_t count = (uint32_t)min((uint64_t)rpw, ((uint64_t)nrows - g64 + nw - 1u) / nw);
  auto idx_of = [&](uint32_t i) { return (uint32_t)(g64 + (uint64_t)i * nw); };
  const uint32_t PM = 1u << log2pm;
  const uint32_t cshift = K32 ? log2pm : 32u; // an entry is column << cshift | product number
  const size_t region = ((size_t)(K32 ? 20u : 32u) << log2pm) + sh_scratch(true, STEPS);
  unsigned char *reg = smem + w * region;
  key_t *tab = reinterpret_cast<key_t *>(reg);
  double *prodv = reinterpret_cast<double *>(reg + (size_t)PM * 3u * sizeof(key_t));
  unsigned char *scratch = reg + (size_t)PM * (3u * sizeof(key_t) + 8u);
  constexpr key_t EMPTY64 = ~(key_t)0;
  auto col_of = [&](key_t e) { return (uint32_t)(e >> cshift); };
  auto q_of = [&](key_t e) { return K32 ? (uint32_t)e & (PM - 1u) : (uint32_t)e; };
  uint32_t nz = 0;
  const bool lin = scalar_view(cm.cdf)[CDF_BINS + 1u] != 0.0f; // wave-uniform
  ShortFetch fa;
  ShortSet<true, STEPS> cur, nxt;
  const double *avs = reinterpret_cast<const double *>(scratch + 768 + 64 * STEPS);
  short_fetch<true>(A, B, m, cm, o.c_ptr, rows, idx_of(0), lane, fa);
  short_expand<true, STEPS>(B, cm, fa, lane, scratch, PM, cur);
  if (1u < count) short_fetch<true>(A, B, m, cm, o.c_ptr, rows, idx_of(1), lane, fa);
  for (uint32_t i = 0; i < count; ++i) {
    const uint32_t total = min(cur.total, PM), n = cur.n;
    const uint64_t off = cur.off;
    const uint32_t H = 2u * total, T = (3u * total + 63u) & ~63u; // slots in use: a multiple of 64, <= 3 PM
    if constexpr (K32) {
      for (uint32_t j = lane * 4u; j < T; j += 256u)
        *reinterpret_cast<uint4 *>(&tab[j]) = make_uint4(~0u, ~0u, ~0u, ~0u);
    } else {
      for (uint32_t j = lane * 2u; j < T; j += 128u)
        *reinterpret_cast<ulonglong2 *>(&tab[j]) = make_ulonglong2(~0ull, ~0ull);
    }
    // the row's own column range stretched over the home slots
    // (the same expression as colmap_eval, so that a product in column cmin maps to exactly f_lo)
    const float f_lo = lin ? (float)cur.cmin : colmap_interp(cm, cur.cmin, cur.f[0], cur.f[1]);
    const float f_hi = lin ? (float)cur.cmax : colmap_interp(cm, cur.cmax, cur.f[2], cur.f[3]);
    const float scale = (float)H / (f_hi - f_lo + 1.0f);
    key_t x[STEPS];
    uint32_t h[STEPS];
    float fc[STEPS];
#pragma unroll
    for (int u = 0; u < STEPS; ++u) fc[u] = cur.c[u] != EMPTY_KEY ? colmap_eval(cm, cur.c[u], lin) : 0.0f;
    lds_fence();
#pragma unroll
    for (int u = 0; u < STEPS; ++u) {
      x[u] = EMPTY64;
      h[u] = lane; // (a lane without a product takes part with "empty", which changes no slot: a slot of its own)
      if (cur.c[u] != EMPTY_KEY) {
        const uint32_t q = (uint32_t)u * 64u + lane;
        prodv[q] = __dmul_rn(avs[cur.t[u]], cur.bv[u]);
        x[u] = ((key_t)cur.c[u] << cshift) | q;
        h[u] = min(H - 1u, (uint32_t)fmaxf(0.0f, (fc[u] - f_lo) * scale)); // (never below zero: the map is monotone)
      }
    }
    // the row's values and A entries are used up: the next row is expanded now (its B entries are in
    // flight during the rest of this row, and only half as many registers are live across the loads)
    if (i + 1u < count) {
      short_expand<true, STEPS>(B, cm, fa, lane, scratch, PM, nxt);
      if (i + 2u < count) short_fetch<true>(A, B, m, cm, o.c_ptr, rows, idx_of(i + 2u), lane, fa);
    }
    // 1. sorted insertion; the chains of a lane's products advance together: the atomics of a round
    //    are issued back to back, their results read afterwards, no branch in between (the minimum
    //    with "empty" changes no slot, so a lane whose chain has ended just keeps going)
    if (T) // (a lane's slots stay below T, which is at least 64 here)
      for (;;) {
        key_t old[STEPS];
#pragma unroll
        for (int u = 0; u < STEPS; ++u) old[u] = atomicMin(&tab[h[u]], x[u]);
        uint32_t left = 0; // (wave-uniform) chains that go on
        uint64_t bal[STEPS];
#pragma unroll
        for (int u = 0; u < STEPS; ++u) {
          x[u] = old[u] > x[u] ? old[u] : x[u]; // the larger one is carried on; "empty" ends the chain
          const bool last_slot = h[u] + 1u >= T; // cannot happen to a chain: at most `total` entries past the last home slot
          x[u] = last_slot ? EMPTY64 : x[u];
          h[u] = x[u] != EMPTY64 ? h[u] + 1u : lane;
          bal[u] = __ballot(x[u] != EMPTY64);
          left += (uint32_t)__popcll(bal[u]);
        }
        if (left == 0u) break;
        // once 64 chains or fewer go on, each gets a lane of its own (through the part of the walk scratch
        // that the next row's expansion has left free: its A values, kept behind it, stay): a round then
        // costs one atomic instead of one per step of the row
        if (left <= 64u) {
          key_t *lx = reinterpret_cast<key_t *>(scratch);
          uint32_t *lh = reinterpret_cast<uint32_t *>(scratch + 512);
          uint32_t before = 0;
#pragma unroll
          for (int u = 0; u < STEPS; ++u) {
            if (x[u] != EMPTY64) {
              const uint32_t at = before + (uint32_t)__popcll(bal[u] & ((1ull << lane) - 1ull));
              lx[at] = x[u];
              lh[at] = h[u];
            }
            before += (uint32_t)__popcll(bal[u]);
          }
          lds_fence();
          key_t xx = EMPTY64;
          uint32_t hh = lane;
          if (lane < left) { xx = lx[lane]; hh = lh[lane]; }
          for (;;) {
            const key_t o1 = atomicMin(&tab[hh], xx);
            xx = o1 > xx ? o1 : xx;
            xx = hh + 1u >= T ? EMPTY64 : xx;
            hh = xx != EMPTY64 ? hh + 1u : lane;
            if (!__any(xx != EMPTY64)) break;
          }
          break;
        }
      }
    lds_fence();
    // 2. the occupied slots are moved to the front of the table, in slot order (16 bytes of slots per
    //    lane and step; an entry moves to a slot at or before its own, and every lane has read its
    //    slots before any lane writes: in place)
    {
      constexpr uint32_t SPL = 16u / sizeof(key_t); // slots per lane
      uint32_t base = 0;
      for (uint32_t s0 = 0; s0 < T; s0 += 64u * SPL) {
        const uint32_t sl = s0 + lane * SPL;
        key_t e[SPL];
        if constexpr (K32) {
          const uint4 v = sl < T ? *reinterpret_cast<const uint4 *>(&tab[sl]) : make_uint4(~0u, ~0u, ~0u, ~0u);
          e[0] = v.x; e[1] = v.y; e[2] = v.z; e[3] = v.w;
        } else {
          const ulonglong2 v = sl < T ? *reinterpret_cast<const ulonglong2 *>(&tab[sl]) : make_ulonglong2(~0ull, ~0ull);
          e[0] = v.x; e[1] = v.y;
        }
        uint32_t c = 0;
#pragma unroll
        for (uint32_t k = 0; k < SPL; ++k) c += e[k] != EMPTY64 ? 1u : 0u;
        const uint32_t incl = group_incl_scan<64>(c);
        uint32_t p = base + incl - c;
        base += group_last<64>(incl);
        lds_fence();
#pragma unroll
        for (uint32_t k = 0; k < SPL; ++k)
          if (e[k] != EMPTY64) tab[p++] = e[k];
      }
    }
    lds_fence();
    // 3. runs of equal columns: the first entry of a column sums the column's entries, which follow it
    uint32_t base = 0;
    const bool plain = n == total; // no column twice: every entry is its own output
    for (uint32_t s0 = 0; s0 < total; s0 += 64u) {
      const uint32_t sl = s0 + lane;
      const bool occ = sl < total;
      const key_t e = occ ? tab[sl] : EMPTY64;
      const uint32_t col = col_of(e);
      const uint32_t prev_col = (sl && !plain && occ) ? col_of(tab[sl - 1u]) : col_of(EMPTY64);
      const bool start = occ && prev_col != col; // (no entry has the column an empty key reads as)
      const uint64_t bal = __ballot(start);
      if (start) {
        double v = prodv[q_of(e)];
        if (!plain)
          for (uint32_t j = sl + 1u; j < total; ++j) {
            const key_t g = tab[j];
            if (col_of(g) != col) break;
            v += prodv[q_of(g)];
          }
        const uint32_t pp = base + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
        if (pp < n) {
          o.c_col[off + pp] = col;
          o.c_val[off + pp] = v;
          nz += (v == 0.0) ? 1u : 0u;
        }
      }
      base += (uint32_t)__popcll(bal);
    }
    lds_fence();
    cur = nxt;
  }
  nz = group_sum<64>(nz);
  if (lane == 0 && nz) atomicAdd(o.zero_count, (unsigned long long)nz);
}

// ---- column map of a B matrix: histogram over CDF_BINS column ranges, then its prefix sums ----------
__global__ void __launch_bounds__(256)
    k_col_hist(uint64_t nnz, const uint32_t *__restrict__ col, uint32_t shift, uint32_t *__restrict__ hist) {
  __shared__ uint32_t s_h[CDF_BINS];
  for (uint32_t i = threadIdx.x; i < CDF_BINS; i += blockDim.x) s_h[i] = 0u;
  __syncthreads();
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nnz; i += (uint64_t)gridDim.x * blockDim.x)
    atomicAdd(&s_h[min(col[i] >> shift, CDF_BINS - 1u)], 1u);
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < CDF_BINS; i += blockDim.x)
    if (s_h[i]) atomicAdd(&hist[i], s_h[i]);
}
// one workgroup of 1024 threads, two bins per thread; cdf[b] = entries in the bins before b.
// cdf[CDF_BINS + 1] = 1 where the entries are spread evenly over the `used` column ranges that exist
// (no prefix count further than 1/256 of the entries from the straight line): the kernel then takes
// the column itself for the map and reads no table (Erdos-Renyi: a tenth of its numeric pass).
__global__ void __launch_bounds__(1024) k_col_cdf(const uint32_t *__restrict__ hist, float *__restrict__ cdf, uint32_t used) {
  __shared__ uint64_t s_wave[16];
  __shared__ unsigned long long s_dev;
  const uint32_t i = threadIdx.x;
  if (i == 0) s_dev = 0ull;
  const uint64_t a = hist[2u * i], b = hist[2u * i + 1u];
  uint64_t tot;
  const uint64_t ex = block_excl_scan64(a + b, s_wave, &tot); // (barriers inside: s_dev is set)
  cdf[2u * i] = (float)ex;
  cdf[2u * i + 1u] = (float)(ex + a);
  unsigned long long dev = 0;
#pragma unroll
  for (uint32_t h = 0; h < 2u; ++h) {
    const uint32_t bin = 2u * i + h;
    if (bin <= used) {
      const uint64_t have = h ? ex + a : ex;
      const uint64_t line = (uint64_t)((unsigned __int128)tot * bin / max(used, 1u));
      dev = max(dev, (unsigned long long)(have > line ? have - line : line - have));
    }
  }
  if (dev) atomicMax(&s_dev, dev);
  __syncthreads();
  if (i == 0) {
    cdf[CDF_BINS] = (float)tot;
    cdf[CDF_BINS + 1u] = s_dev <= (tot >> 8) ? 1.0f : 0.0f;
  }
}

} // namespace sg
