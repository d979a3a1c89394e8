// Per-multiply preparation: B row info, A row analysis, row groups, exclusive scan, stable partition of rows by bin.
// Part of the gfx950 SpGEMM kernels; included through kernels.h.
#pragma once
#include "kernels_common.h"

namespace sg {

// ---- B row info: two uint4 per row, {min col, max col, length, 1 if the row's columns are not strictly
// increasing} and {start of the row in B.col / B.val (u64), start of its padded records (u64, set by
// k_prec_patch)}.  The counter of rows that are not strictly increasing is zero on entry.  8 lanes per row.
// The two 16-byte halves of a row's record, lo = {min col, max col, length, unsorted} and hi =
// {first entry (u64), first padded record (u64)}: side by side (sh = 1, hi_base = 1: one 32-byte
// gather serves the analysis of matrices with short rows, which reads both halves per A entry), or
// in two arrays (sh = 0, hi_base = rows: the analysis of other matrices reads only lo, and twice as
// many rows then share a cache line).  p[2 * rows].x counts the unsorted rows either way.
struct RowInfo {
  uint4 *p;
  uint32_t sh;
  uint64_t hi_base;
  // one bit per row: the row has the column list of the row before it (same length, same columns in
  // the same, strictly increasing, order).  Consecutive entries of an A row that select such rows walk
  // one column list several times: the window kernels rank it once (kernels_numeric.h, RunWalk).
  uint32_t *same;
  __device__ __forceinline__ uint4 &lo(uint64_t k) const { return p[k << sh]; }
  __device__ __forceinline__ uint4 &hi(uint64_t k) const { return p[hi_base + (k << sh)]; }
};

// `same` is zero on entry; p[2 * rows].y receives the longest run of rows with one column list (capped at 255).
__global__ void __launch_bounds__(256) k_b_rowinfo(uint64_t rows, const uint64_t *__restrict__ b_ptr,
                                                   const uint32_t *__restrict__ b_col,
                                                   RowInfo info) {
  constexpr int G = 8;
  const uint32_t lg = threadIdx.x & (G - 1);
  // (a grid of a few thousand workgroups strides over the rows: a B of which only a few row sections
  // are present is mostly empty rows, and a workgroup per 32 of them would cost more than the rows do)
  for (uint64_t row0 = (uint64_t)blockIdx.x * (256 / G); row0 < rows; row0 += (uint64_t)gridDim.x * (256 / G)) {
  const uint64_t row = row0 + threadIdx.x / G;
  uint32_t mn = 0xFFFFFFFFu, mx = 0, unsorted = 0, differs = 0;
  uint64_t b0 = 0, b1 = 0, bp = 0;
  if (row < rows) {
    b0 = b_ptr[row];
    b1 = b_ptr[row + 1];
    bp = row ? b_ptr[row - 1] : b0;
  }
  const uint64_t len = b1 - b0;
  const bool cmp = row > 0 && len > 0 && b0 - bp == len; // the row before is as long: compare the columns
  for (uint64_t j = b0 + lg; j < b1; j += G) {
    uint32_t c = b_col[j];
    if (j > b0 && b_col[j - 1] >= c) unsorted = 1;
    if (cmp && b_col[j - len] != c) differs = 1;
    mn = min(mn, c);
    mx = max(mx, c);
  }
  mn = group_min<G>(mn);
  mx = group_max<G>(mx);
  unsorted = group_max<G>(unsorted);
  differs = group_max<G>(differs);
  if (row < rows && lg == 0) {
    info.lo(row) = make_uint4(mn, mx, len > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)len, unsorted);
    info.hi(row) = make_uint4((uint32_t)b0, (uint32_t)(b0 >> 32), 0u, 0u);
    if (unsorted) atomicAdd(&info.p[2 * rows].x, 1u);
    if (cmp && !differs && !unsorted && len <= 0xFFFFFFFFull) atomicOr(&info.same[row >> 5], 1u << (row & 31u));
  }
  }
}

// longest run of rows with one column list = 1 + the longest streak of set bits in `same` (a run is a
// row with its bit clear and the rows with set bits that follow it), capped at `cap`.  One thread per
// 32-bit word: the longest streak inside the word, and the streak that leaves the word at its top,
// followed into the next words.  Bits beyond the last row are clear.
__global__ void __launch_bounds__(256) k_b_maxrun(uint64_t rows, RowInfo info, uint32_t cap) {
  const uint64_t words = (rows + 31u) >> 5;
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t best = 0;
  if (t < words) {
    const uint32_t w = info.same[t];
    for (uint32_t x = w; x; x &= x << 1) ++best; // rounds until no set bit has a set bit below it
    uint32_t top = (uint32_t)__clz((int)~w); // set bits at the top of the word (32 when all are set)
    if (w == 0xFFFFFFFFu) top = 32u;
    if (top) {
      uint32_t streak = top;
      for (uint64_t k = t + 1; k < words && streak < cap; ++k) {
        const uint32_t n = info.same[k];
        const uint32_t low = n == 0xFFFFFFFFu ? 32u : (uint32_t)__builtin_ctz(~n); // set bits at the bottom
        streak += low;
        if (low < 32u) break;
      }
      best = max(best, streak);
    }
  }
  uint32_t run = best ? min(cap, best + 1u) : 0u;
  run = group_max<64>(run);
  if ((threadIdx.x & 63u) == 0 && run > 1 && run > __hip_atomic_load(&info.p[2 * rows].y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
    atomicMax(&info.p[2 * rows].y, run);
}

// ---- B rows as padded 16-byte records, for the short-row numeric kernel ------------------------------
// A row of up to PREC_MAX_LEN entries gets ceil(len / 8) * 8 records {column, 0, value}, starting
// at a multiple of 128 bytes: such a row is fetched as len / 8 whole 128-byte lines, where its
// 4-byte columns and 8-byte values in two arrays straddle three to four.  Longer rows get none (a
// short row of A, at most 512 products, cannot refer to them).  The record start of every row goes
// into the second half of its row-info record.  The columns alone are kept a second time at the
// same positions (a row starts at a multiple of 32 bytes; where all rows have 9 to 16 entries, of 64):
// the count-only symbolic pass fetches such a row as one line, where B's own column array straddles
// two half of the time.
constexpr uint32_t PREC_MAX_LEN = 512;
__global__ void __launch_bounds__(256) k_prec_count(uint64_t rows, RowInfo info, uint32_t *__restrict__ count) {
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  const uint32_t len = info.lo(r).z;
  count[r] = len <= PREC_MAX_LEN ? (len + 7u) & ~7u : 0u;
}
__global__ void __launch_bounds__(256) k_prec_patch(uint64_t rows, const uint64_t *__restrict__ start, RowInfo info) {
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  uint4 v = info.hi(r);
  v.z = (uint32_t)start[r];
  v.w = (uint32_t)(start[r] >> 32);
  info.hi(r) = v;
}
__global__ void __launch_bounds__(256) k_prec_fill(uint64_t rows, const uint64_t *__restrict__ b_ptr, const uint32_t *__restrict__ b_col,
                                                   const double *__restrict__ b_val, RowInfo info,
                                                   uint4 *__restrict__ prec, uint32_t *__restrict__ pcol) {
  constexpr int G = 8;
  const uint64_t row = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
  const uint32_t lg = threadIdx.x & (G - 1);
  if (row >= rows) return;
  const uint64_t b0 = b_ptr[row], b1 = b_ptr[row + 1];
  if (b1 - b0 > PREC_MAX_LEN) return;
  const uint4 hi = info.hi(row);
  const uint64_t start = ((uint64_t)hi.w << 32) | hi.z;
  uint4 *out = prec + start;
  uint32_t *out_col = pcol + start; // the columns alone, at the same positions: rows start at multiples of 32 bytes
  for (uint64_t j = b0 + lg; j < b1; j += G) {
    const unsigned long long v = (unsigned long long)__double_as_longlong(b_val[j]);
    const uint32_t c = b_col[j];
    out[j - b0] = make_uint4(c, 0u, (uint32_t)v, (uint32_t)(v >> 32));
    out_col[j - b0] = c;
  }
}

// ---- A row analysis --------------------------------------------------------------------------
// Per A row: nprod = sum of B row lengths, [cmin, cmax] = hull of the B rows' column ranges,
// the symbolic bin key and the upper bound `ub` on the row's distinct columns.
constexpr uint32_t NPROD_SLOTS = 256;
struct AnalyzeOut {
  uint32_t *r_cmin;
  uint32_t *r_cmax;
  uint32_t *r_ub;
  uint8_t *r_key;
  uint64_t *r_nprod;       // optional (may be null)
  uint64_t *total_nprod;   // NPROD_SLOTS partial counters, summed by the host
  uint32_t *long_rows;     // rows with more than long_len A entries (handled by k_analyze_long)
  uint32_t *long_count;
  uint8_t *r_same;         // 1 when the row's column list equals the previous row's (same order)
  uint64_t *a_p0;          // per A entry of a short row (kernels_short.h): start of its B row (may be null) ...
  uint32_t *a_len;         // ... and the B row's length (null: no short rows)
  uint64_t *a_pk;          // ... and the start of the B row's padded records (may be null)
};

__device__ __forceinline__ void analyze_finish(uint32_t row, uint64_t np, uint32_t mn, uint32_t mx,
                                               uint32_t len_a, const AnalyzeOut &o,
                                               const ClassParams &cp) {
  uint32_t span = np ? (mx - mn + 1u) : 0u;
  uint32_t key = sym_key_of(np, span, len_a, cp);
  uint64_t ub = np;
  if (key != KEY_NONE && (key < (uint32_t)SYM_HASH0 || key >= (uint32_t)SYM_GLOBAL0))
    ub = np < span ? np : span;
  else if (key != KEY_NONE && (key & 3u) == 0u)
    ub = 0; // short row: its symbolic pass only counts, no column list is kept
  o.r_cmin[row] = mn;
  o.r_cmax[row] = mx;
  o.r_ub[row] = (uint32_t)ub;
  o.r_key[row] = (uint8_t)key;
  if (o.r_nprod) o.r_nprod[row] = np;
}

template <int G>
__global__ void __launch_bounds__(256)
    k_analyze(uint32_t rows_a, const uint64_t *__restrict__ a_ptr, const uint32_t *__restrict__ a_col,
              RowInfo b_info, AnalyzeOut o, ClassParams cp, uint32_t long_len) {
  const uint64_t row64 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
  const uint32_t row = row64 < rows_a ? (uint32_t)row64 : rows_a; // rows_a: no row (ids beyond 2^32 / G must not alias)
  uint32_t lg = threadIdx.x & (G - 1);
  uint64_t np = 0;
  uint32_t mn = 0xFFFFFFFFu, mx = 0;
  uint64_t a0 = 0, a1 = 0;
  bool is_long = false;
  if (row < rows_a) {
    a0 = a_ptr[row];
    a1 = a_ptr[row + 1];
    is_long = (a1 - a0) > long_len;
  }
  uint32_t differs = 1;
  // a row of at most G entries (one per lane) keeps the records it reads for the per-entry arrays below
  const bool one_pass = o.a_len && (a1 - a0) <= (uint64_t)G;
  uint4 keep_lo = make_uint4(0u, 0u, 0u, 0u), keep_hi = keep_lo;
  if (!is_long) {
    // does the row repeat the previous row's column list?
    uint64_t p0 = 0;
    bool cmp = false;
    if (row > 0 && row < rows_a) {
      p0 = a_ptr[row - 1];
      cmp = (a0 - p0) == (a1 - a0) && a1 > a0;
    }
    differs = cmp ? 0u : 1u;
    // U entries per lane at a time: their column indices are loaded together, then their records
    // (one entry after the other, a row of 35 entries at 8 lanes waited for ten dependent loads)
    constexpr int U = 4;
    for (uint64_t kb = a0 + lg; kb < a1; kb += (uint64_t)G * U) {
      uint32_t kcol[U], pcol[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint64_t ka = kb + (uint64_t)u * G;
        kcol[u] = ka < a1 ? a_col[ka] : 0u;
        pcol[u] = (cmp && ka < a1) ? a_col[p0 + (ka - a0)] : kcol[u];
      }
      uint4 bi[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint64_t ka = kb + (uint64_t)u * G;
        bi[u] = ka < a1 ? b_info.lo(kcol[u]) : make_uint4(0u, 0u, 0u, 0u);
      }
      if (one_pass) { // (a single entry per lane: u = 0)
        keep_lo = bi[0];
        keep_hi = b_info.hi(kcol[0]); // (the other half of the record: the same line)
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (pcol[u] != kcol[u]) differs = 1;
        if (bi[u].z) {
          np += bi[u].z;
          mn = min(mn, bi[u].x);
          mx = max(mx, bi[u].y);
        }
      }
    }
  }
  differs = group_max<G>(differs);
  np = group_sum64<G>(np);
  mn = group_min<G>(mn);
  mx = group_max<G>(mx);
  // a short row leaves, per A entry, the start and length of the B row for its two kernels (the
  // records were just read: these loads hit the caches)
  if (row < rows_a && !is_long && o.a_len) {
    const uint32_t len_a = (uint32_t)(a1 - a0);
    const uint32_t key = sym_key_of(np, np ? (mx - mn + 1u) : 0u, len_a, cp);
    if (key >= (uint32_t)SYM_HASH0 && key < (uint32_t)SYM_GLOBAL0 && (key & 3u) == 0u) {
      if (one_pass) {
        const uint64_t ka = a0 + lg;
        if (ka < a1) {
          if (o.a_p0) o.a_p0[ka] = ((uint64_t)keep_hi.y << 32) | keep_hi.x;
          o.a_len[ka] = keep_lo.z;
          if (o.a_pk) o.a_pk[ka] = ((uint64_t)keep_hi.w << 32) | keep_hi.z;
        }
      } else {
        for (uint64_t ka = a0 + lg; ka < a1; ka += G) {
          const size_t k = a_col[ka];
          const uint4 lo = b_info.lo(k), hi = b_info.hi(k);
          if (o.a_p0) o.a_p0[ka] = ((uint64_t)hi.y << 32) | hi.x;
          o.a_len[ka] = lo.z;
          if (o.a_pk) o.a_pk[ka] = ((uint64_t)hi.w << 32) | hi.z;
        }
      }
    }
  }
  uint64_t contrib = 0;
  if (row < rows_a && lg == 0) {
    o.r_same[row] = differs ? 0 : 1;
    if (is_long) {
      uint32_t p = atomicAdd(o.long_count, 1u);
      o.long_rows[p] = row;
    } else {
      analyze_finish(row, np, mn, mx, (uint32_t)(a1 - a0), o, cp);
      contrib = np;
    }
  }
  // product total: reduce over the workgroup, then one atomic per workgroup, spread over
  // NPROD_SLOTS counters (atomics on a single word would serialise the whole grid)
  __shared__ uint64_t s_c[4];
  contrib = group_sum64<64>(contrib);
  if ((threadIdx.x & 63) == 0) s_c[threadIdx.x >> 6] = contrib;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint64_t t = s_c[0] + s_c[1] + s_c[2] + s_c[3];
    if (t) atomicAdd((unsigned long long *)&o.total_nprod[blockIdx.x & (NPROD_SLOTS - 1)], (unsigned long long)t);
  }
}

// One 256-thread workgroup per long A row.
__global__ void __launch_bounds__(256)
    k_analyze_long(const uint64_t *__restrict__ a_ptr, const uint32_t *__restrict__ a_col,
                   RowInfo b_info, AnalyzeOut o, ClassParams cp) {
  __shared__ uint64_t s_np[4];
  __shared__ uint32_t s_mn[4], s_mx[4];
  uint32_t nlong = *o.long_count;
  for (uint32_t idx = blockIdx.x; idx < nlong; idx += gridDim.x) {
    uint32_t row = o.long_rows[idx];
    uint64_t a0 = a_ptr[row], a1 = a_ptr[row + 1];
    uint64_t np = 0;
    uint32_t mn = 0xFFFFFFFFu, mx = 0;
    for (uint64_t ka = a0 + threadIdx.x; ka < a1; ka += blockDim.x) {
      uint4 bi = b_info.lo(a_col[ka]);
      if (bi.z) {
        np += bi.z;
        mn = min(mn, bi.x);
        mx = max(mx, bi.y);
      }
    }
    np = group_sum64<64>(np);
    mn = group_min<64>(mn);
    mx = group_max<64>(mx);
    uint32_t w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
      s_np[w] = np;
      s_mn[w] = mn;
      s_mx[w] = mx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      np = s_np[0] + s_np[1] + s_np[2] + s_np[3];
      mn = min(min(s_mn[0], s_mn[1]), min(s_mn[2], s_mn[3]));
      mx = max(max(s_mx[0], s_mx[1]), max(s_mx[2], s_mx[3]));
      analyze_finish(row, np, mn, mx, (uint32_t)(a1 - a0), o, cp);
      if (np) atomicAdd((unsigned long long *)&o.total_nprod[blockIdx.x & (NPROD_SLOTS - 1)], (unsigned long long)np);
    }
    __syncthreads();
  }
}

// ---- row groups ----------------------------------------------------------------------------------
// Consecutive rows of A with identical column lists (the unknowns of one mesh node, for example)
// produce identical patterns in C and walk the same B rows.  Such runs are cut into groups of
// up to GRP_MAX rows; the first row of a group is its leader.  The symbolic pass runs once per
// group, and the numeric pass loads every B entry and finds its output position once per group,
// then adds into one accumulator per row.  Only window-path rows with at most GRP_ALEN A
// entries are grouped.  gcnt: rows in the group for a leader (>= 1), 0 for a follower.
// Every row decides for itself from the `same` flags around it; the rule (position inside the
// run modulo GRP_MAX, runs longer than 56 rows stop grouping) gives the same answer from the
// leader's and from the follower's point of view.
__global__ void __launch_bounds__(256)
    k_group(uint32_t rows, const uint64_t *__restrict__ a_ptr, const uint8_t *__restrict__ same,
            uint8_t *__restrict__ key, uint32_t *__restrict__ ub, uint8_t *__restrict__ gcnt,
            uint32_t *__restrict__ lead, uint32_t enable, uint32_t *__restrict__ gmax_seen) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t c = 0;    // rows of the group this row leads (0: leads none that counts for the LDS sizing)
  bool wide = false;
  if (i < rows) {
    lead[i] = i;
    const uint32_t k = key[i];
    const uint64_t len = a_ptr[i + 1] - a_ptr[i];
    uint32_t g = 1; // gcnt of the row
    if (enable && k < (uint32_t)SYM_HASH0 && len <= GRP_ALEN_WIDE) { // a window row with products
      uint32_t pos = 0; // rows before i that belong to the same run
      while (pos < 56u && pos < i && same[i - pos]) ++pos;
      if (pos < 56u) {
        const uint32_t gmax = group_max_rows(len); // the rows of a run have the same length
        if (pos % gmax != 0) { // follower: handled by its leader
          g = 0;
          key[i] = (uint8_t)KEY_NONE;
          ub[i] = 0;
          lead[i] = i - pos % gmax;
        } else {
          while (g < gmax && i + g < rows && same[i + g]) ++g;
          c = g;
          wide = len > GRP_ALEN;
        }
      }
    }
    gcnt[i] = (uint8_t)g;
  }
  // largest group of the multiply, narrow ([0]) and wide ([1]) rows apart: the group kernel sizes its
  // LDS by it (five unknowns per mesh node make groups of five, not of GRP_MAX)
  // A row by itself counts as no group (the host takes the maximum to be at least 1), so a matrix
  // without groups issues no atomic at all; the others one per workgroup: thousands of wavefronts
  // on one address took longer than the rest of this kernel.
  __shared__ uint32_t s_mn[4], s_mw[4];
  const uint32_t mn = group_max<64>(wide || c < 2u ? 0u : c), mw = group_max<64>(wide && c >= 2u ? c : 0u);
  if ((threadIdx.x & 63u) == 0) { s_mn[threadIdx.x >> 6] = mn; s_mw[threadIdx.x >> 6] = mw; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t bn = max(max(s_mn[0], s_mn[1]), max(s_mn[2], s_mn[3]));
    const uint32_t bw = max(max(s_mw[0], s_mw[1]), max(s_mw[2], s_mw[3]));
    // the reads only save atomics; they bypass the CU's cache, which would keep showing zero
    if (bn && bn > __hip_atomic_load(&gmax_seen[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&gmax_seen[0], bn);
    if (bw && bw > __hip_atomic_load(&gmax_seen[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&gmax_seen[1], bw);
  }
}

// ---- start of a multiply: the counters and the per-row arrays that must start from a known value ---
// One launch instead of four fills (each of which is a launch or two of its own): zero[0..n_zero)
// = 0, nnz[0..rows) = 0, numkey[0..rows) = 0xFF.  nnz and numkey start on 256-byte boundaries and
// own the bytes up to the next one, so they are written 16 bytes at a time.
__global__ void __launch_bounds__(256)
    k_sym_clear(uint32_t *__restrict__ zero_a, uint32_t n_zero_a, uint32_t *__restrict__ zero_b, uint32_t n_zero_b,
                uint4 *__restrict__ nnz16, uint32_t n_nnz16, uint4 *__restrict__ numkey16, uint32_t n_numkey16) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x, stride = gridDim.x * 256u;
  for (uint32_t j = i; j < n_zero_a; j += stride) zero_a[j] = 0u;
  for (uint32_t j = i; j < n_zero_b; j += stride) zero_b[j] = 0u;
  for (uint32_t j = i; j < n_nnz16; j += stride) nnz16[j] = make_uint4(0u, 0u, 0u, 0u);
  for (uint32_t j = i; j < n_numkey16; j += stride) numkey16[j] = make_uint4(~0u, ~0u, ~0u, ~0u);
}

// ---- exclusive scan (uint32 in, uint64 out), three launches ----------------------------------
constexpr int SCAN_BLOCK = 256;
constexpr int SCAN_ITEMS = 8; // per thread -> 2048 per workgroup

__device__ __forceinline__ uint64_t block_excl_scan64(uint64_t v, uint64_t *s_wave /*>=4*/, uint64_t *total) {
  // inclusive scan inside the wavefront
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  uint64_t x = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint64_t t = __shfl_up(x, d, 64);
    if (lane >= d) x += t;
  }
  if (lane == 63) s_wave[w] = x;
  __syncthreads();
  uint64_t base = 0, tot = 0;
  const int nw = blockDim.x >> 6;
  for (int i = 0; i < nw; ++i) {
    if (i < w) base += s_wave[i];
    tot += s_wave[i];
  }
  __syncthreads();
  if (total) *total = tot;
  return base + x - v;
}

__global__ void __launch_bounds__(SCAN_BLOCK)
    k_scan_partials(const uint32_t *__restrict__ in, uint64_t n, uint64_t *__restrict__ partial) {
  __shared__ uint64_t s_wave[4];
  uint64_t base = (uint64_t)blockIdx.x * SCAN_BLOCK * SCAN_ITEMS;
  uint64_t s = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    uint64_t idx = base + (uint64_t)i * SCAN_BLOCK + threadIdx.x;
    if (idx < n) s += in[idx];
  }
  uint64_t tot;
  block_excl_scan64(s, s_wave, &tot);
  if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

// single workgroup: exclusive scan of the partials in place; total -> partial[nparts]
__global__ void __launch_bounds__(1024) k_scan_spine(uint64_t *partial, uint32_t nparts) {
  __shared__ uint64_t s_wave[16];
  __shared__ uint64_t s_carry;
  if (threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  for (uint32_t base = 0; base < nparts; base += 1024) {
    uint32_t idx = base + threadIdx.x;
    uint64_t v = idx < nparts ? partial[idx] : 0;
    uint64_t tot;
    uint64_t ex = block_excl_scan64(v, s_wave, &tot);
    uint64_t carry = s_carry;
    if (idx < nparts) partial[idx] = ex + carry;
    __syncthreads();
    if (threadIdx.x == 0) s_carry = carry + tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[nparts] = s_carry;
}

// The whole scan in one workgroup, for inputs of up to SCAN_SMALL_MAX items (the block histograms of
// a partition, the rows of a small matrix): one launch instead of two.  Every thread owns 16
// consecutive items and has their loads in flight together (a loop over a span of the thread's own
// waited for every load in turn: 13 us for 11,000 items).
constexpr uint32_t SCAN_SMALL_MAX = 16384;
__global__ void __launch_bounds__(1024)
    k_scan_small(const uint32_t *__restrict__ in, uint32_t n, uint64_t *__restrict__ out) {
  static_assert(SCAN_SMALL_MAX == 1024u * 16u, "16 items per thread");
  __shared__ uint64_t s_wave[16];
  const uint32_t lo = threadIdx.x * 16u;
  uint32_t v[16];
#pragma unroll
  for (uint32_t i = 0; i < 16u; ++i) v[i] = lo + i < n ? in[lo + i] : 0u;
  uint64_t s = 0;
#pragma unroll
  for (uint32_t i = 0; i < 16u; ++i) s += v[i];
  uint64_t tot;
  uint64_t ex = block_excl_scan64(s, s_wave, &tot);
#pragma unroll
  for (uint32_t i = 0; i < 16u; ++i) {
    if (lo + i < n) out[lo + i] = ex;
    ex += v[i];
  }
  if (threadIdx.x == 0) out[n] = tot;
}

// out[i] = exclusive prefix; out[n] = total.  Items are laid out thread-contiguous here so the
// per-thread prefix is a running sum.
// own_spine: `partial` holds the workgroups' raw totals (k_scan_spine has not run) and every
// workgroup adds up those before it on its own - one launch less for the short scans that are
// most of them; the sums of all workgroups together read nparts^2 / 2 values, so the host takes this
// form only up to SCAN_OWN_SPINE_MAX parts.
constexpr uint32_t SCAN_OWN_SPINE_MAX = 4096;
__global__ void __launch_bounds__(SCAN_BLOCK)
    k_scan_final(const uint32_t *__restrict__ in, uint64_t n, const uint64_t *__restrict__ partial,
                 uint64_t *__restrict__ out, uint32_t nparts, uint32_t own_spine) {
  __shared__ uint64_t s_wave[4];
  __shared__ uint32_t s_items[SCAN_BLOCK * SCAN_ITEMS];
  uint64_t base = (uint64_t)blockIdx.x * SCAN_BLOCK * SCAN_ITEMS;
  uint64_t before = 0; // sum of the items of the workgroups before this one
  if (own_spine) {
    uint64_t mine = 0;
    for (uint32_t j = threadIdx.x; j < blockIdx.x; j += SCAN_BLOCK) mine += partial[j];
    block_excl_scan64(mine, s_wave, &before);
  } else {
    before = partial[blockIdx.x];
  }
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    uint32_t l = i * SCAN_BLOCK + threadIdx.x;
    uint64_t idx = base + l;
    s_items[l] = idx < n ? in[idx] : 0u;
  }
  __syncthreads();
  uint32_t v[SCAN_ITEMS];
  uint64_t s = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    v[i] = s_items[threadIdx.x * SCAN_ITEMS + i];
    s += v[i];
  }
  uint64_t tot;
  uint64_t ex = block_excl_scan64(s, s_wave, &tot) + before;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    uint64_t idx = base + (uint64_t)threadIdx.x * SCAN_ITEMS + i;
    if (idx < n) out[idx] = ex;
    ex += v[i];
  }
  if (own_spine) {
    if (blockIdx.x == nparts - 1u && threadIdx.x == 0) out[n] = before + tot;
  } else if (blockIdx.x == 0 && threadIdx.x == 0) {
    out[n] = partial[nparts];
  }
}

// ---- stable partition of rows by an 8-bit key ---------------------------------------------------
constexpr int PART_BLOCK = 256;
constexpr int PART_ROUNDS = 4; // 1024 rows per workgroup

__global__ void __launch_bounds__(PART_BLOCK)
    k_part_hist(const uint8_t *__restrict__ key, uint32_t n, uint32_t nbins, uint32_t nblocks,
                uint32_t *__restrict__ blk_hist /* [nbins][nblocks] */) {
  __shared__ uint32_t s_h[MAX_BINS];
  for (uint32_t b = threadIdx.x; b < nbins; b += PART_BLOCK) s_h[b] = 0;
  __syncthreads();
  uint32_t base = blockIdx.x * PART_BLOCK * PART_ROUNDS;
  for (int r = 0; r < PART_ROUNDS; ++r) {
    uint32_t i = base + r * PART_BLOCK + threadIdx.x;
    if (i < n) {
      uint32_t k = key[i];
      if (k < nbins) atomicAdd(&s_h[k], 1u);
    }
  }
  __syncthreads();
  for (uint32_t b = threadIdx.x; b < nbins; b += PART_BLOCK) blk_hist[(size_t)b * nblocks + blockIdx.x] = s_h[b];
}

// blk_off = exclusive scan of blk_hist (bin-major), so bin b starts at blk_off[b*nblocks].
__global__ void __launch_bounds__(PART_BLOCK)
    k_part_scatter(const uint8_t *__restrict__ key, uint32_t n, uint32_t nbins, uint32_t nblocks,
                   const uint64_t *__restrict__ blk_off, uint32_t *__restrict__ rows_out,
                   uint64_t *__restrict__ bin_off) {
  __shared__ uint32_t s_base[MAX_BINS];
  __shared__ uint32_t s_wcnt[4][MAX_BINS];
  // bin_off[b] = start of bin b in rows_out (b = 0..nbins), read from the scanned histogram
  if (blockIdx.x == 0)
    for (uint32_t b = threadIdx.x; b <= nbins; b += PART_BLOCK) bin_off[b] = blk_off[(size_t)b * nblocks];
  for (uint32_t b = threadIdx.x; b < nbins; b += PART_BLOCK)
    s_base[b] = (uint32_t)blk_off[(size_t)b * nblocks + blockIdx.x];
  const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  uint32_t base = blockIdx.x * PART_BLOCK * PART_ROUNDS;
  for (int r = 0; r < PART_ROUNDS; ++r) {
    for (uint32_t b = threadIdx.x; b < nbins; b += PART_BLOCK) {
      s_wcnt[0][b] = 0; s_wcnt[1][b] = 0; s_wcnt[2][b] = 0; s_wcnt[3][b] = 0;
    }
    __syncthreads();
    uint32_t i = base + r * PART_BLOCK + threadIdx.x;
    uint32_t k = i < n ? key[i] : KEY_NONE;
    bool valid = k < nbins;
    uint32_t rank = 0;
    unsigned long long left = __ballot(valid);
    while (left) {
      int leader = __ffsll((long long)left) - 1;
      uint32_t kb = __shfl(k, leader, 64);
      unsigned long long m = __ballot(valid && k == kb);
      if (valid && k == kb) rank = __popcll(m & ((1ull << lane) - 1ull));
      if ((int)lane == leader) s_wcnt[w][kb] = __popcll(m);
      left &= ~m;
    }
    __syncthreads();
    if (valid) {
      uint32_t off = s_base[k] + rank;
      for (uint32_t ww = 0; ww < w; ++ww) off += s_wcnt[ww][k];
      rows_out[off] = i;
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < nbins; b += PART_BLOCK)
      s_base[b] += s_wcnt[0][b] + s_wcnt[1][b] + s_wcnt[2][b] + s_wcnt[3][b];
    __syncthreads();
  }
}

} // namespace sg
