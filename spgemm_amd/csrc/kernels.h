// Device kernels of the gfx950 SpGEMM engine (row-wise Gustavson, C = A * B, fp64).
//
// Execution model used throughout: 64-lane wavefronts; a "group" is G consecutive lanes
// (G in {8,16,32,64}) inside one wavefront.  LDS operations of one wavefront execute in issue
// order, so lanes of a wavefront hand data to each other through LDS with a compiler-only
// fence (lds_fence) and no s_barrier.
//
// Ordering rule that makes the result bit-identical to the reference's CPU loop
// (src/mults/drop_parallel.cpp:46-57, accumulator include/matrix_impl.hpp:81-95): every output
// row is owned by ONE group; the group walks the A row's entries sequentially in storage order
// (the k order of the CPU loop) and, for one k, its lanes take different entries of B row k,
// which have distinct columns.  Every (i,j) therefore receives its additions in the same order
// as on the CPU, products and sums are rounded separately (no fused multiply-add), and no
// two lanes of one instruction ever add to the same accumulator.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sg {

constexpr uint32_t EMPTY_KEY = 0xFFFFFFFFu;
constexpr uint32_t KEY_NONE = 255u; // row needs no kernel (no products)

// ---- bin keys ------------------------------------------------------------------------------
// symbolic:  [0,24)  window, key = span_tier*4 + gsel, span <= 2048 << span_tier
//            [24,52) hash,   key = 24 + tier*4 + gsel, nprod <= 64 << tier (table of 128 << tier slots)
//            [52,56) global, key = 52 + gsel
// numeric:   [0,96)   window, key = (span_tier*4 + cap_tier)*4 + gsel, nnz <= 128 << (2*cap_tier)
//            [96,124) hash,   key = 96 + tier*4 + gsel,                 nnz <= 64 << tier
//            124      global
//            [125,149) window, row groups: key = 125 + span_tier*4 + cap_tier (one wavefront per group
//                      of up to GRP_MAX consecutive rows of A with identical column lists)
constexpr int HASH_TIERS = 7; // table sizes 128 << tier
constexpr int SYM_WIN0 = 0, SYM_HASH0 = 24, SYM_GLOBAL0 = SYM_HASH0 + 4 * HASH_TIERS, SYM_NBINS = SYM_GLOBAL0 + 4;
constexpr int NUM_WIN0 = 0, NUM_HASH0 = 96, NUM_GLOBAL = NUM_HASH0 + 4 * HASH_TIERS, NUM_GRP0 = NUM_GLOBAL + 1,
              NUM_NBINS = NUM_GRP0 + 24;
constexpr int MAX_BINS = 192;
static_assert(NUM_NBINS <= MAX_BINS && NUM_NBINS < 255, "bin keys are 8 bit");
constexpr uint32_t GRP_MAX = 8;   // rows per group
constexpr uint32_t GRP_ALEN = 64; // longest A row of such a group (the rows' values are staged in LDS)
// "wide" groups: A rows of 65..GRP_ALEN_WIDE entries are grouped by GRP_MAX_WIDE at most (the same
// LDS holds fewer, longer value lists)
constexpr uint32_t GRP_MAX_WIDE = 4, GRP_ALEN_WIDE = 128;
__host__ __device__ inline uint32_t group_max_rows(uint64_t len_a) { return len_a > GRP_ALEN ? GRP_MAX_WIDE : GRP_MAX; }

struct ClassParams {
  uint32_t win_max_span;   // window path allowed up to this column span (<= 65536)
  uint32_t win_ratio;      // ... and only if span <= win_ratio * nprod
  uint32_t hash_max_nprod; // LDS hash symbolic up to this many products (<= 4096)
  uint32_t g_thresh0, g_thresh1, g_thresh2; // mean B-row length thresholds for 8/16/32 lanes
};

__host__ __device__ inline uint32_t ceil_log2_u32(uint32_t x) {
  uint32_t l = 0;
  while ((1u << l) < x && l < 31) ++l;
  return l;
}

__host__ __device__ inline uint32_t gsel_of(uint64_t nprod, uint32_t len_a, const ClassParams &p) {
  uint32_t avg = len_a ? (uint32_t)((nprod + len_a - 1) / len_a) : 0;
  return avg <= p.g_thresh0 ? 0u : avg <= p.g_thresh1 ? 1u : avg <= p.g_thresh2 ? 2u : 3u;
}

__host__ __device__ inline uint32_t sym_key_of(uint64_t nprod, uint32_t span, uint32_t len_a,
                                               const ClassParams &p) {
  if (nprod == 0) return KEY_NONE;
  uint32_t g = gsel_of(nprod, len_a, p);
  if (span <= p.win_max_span && (uint64_t)span <= (uint64_t)p.win_ratio * nprod) {
    uint32_t t = ceil_log2_u32(span < 2048u ? 2048u : span) - 11u;
    return SYM_WIN0 + t * 4 + g;
  }
  if (nprod <= p.hash_max_nprod) {
    uint32_t t = ceil_log2_u32(nprod < 64u ? 64u : (uint32_t)nprod) - 6u; // nprod <= 64 << t
    return SYM_HASH0 + t * 4 + g;
  }
  return SYM_GLOBAL0 + g;
}

// group bins: key = NUM_GRP0 + span_tier * 4 + q; q = 0: up to 128 outputs, q = 1: up to 512 (both:
// up to GRP_MAX rows of up to GRP_ALEN entries), q = 2: up to 512 outputs, wide group
// hash_num_max: most outputs of a row that is accumulated by the LDS hash kernel (<= 4096); rows
// beyond it go to the big-row bin
__host__ __device__ inline uint32_t num_key_of(uint32_t sym_key, uint32_t nnz, uint32_t hash_num_max,
                                               uint32_t group_rows = 1, bool wide = false) {
  if (nnz == 0) return KEY_NONE;
  uint32_t g = sym_key & 3u;
  if (sym_key < (uint32_t)SYM_HASH0) {
    uint32_t st = sym_key >> 2;
    if (nnz <= 8192u) {
      uint32_t ct = nnz <= 128 ? 0u : nnz <= 512 ? 1u : nnz <= 2048 ? 2u : 3u;
      if (group_rows > 1 && ct <= 1) return NUM_GRP0 + st * 4 + (wide ? 2u : ct);
      return NUM_WIN0 + (st * 4 + ct) * 4 + g;
    }
    return NUM_GLOBAL;
  }
  if (nnz <= hash_num_max) {
    uint32_t ct = ceil_log2_u32(nnz < 64u ? 64u : nnz) - 6u; // nnz <= 64 << ct
    return NUM_HASH0 + ct * 4 + g;
  }
  return NUM_GLOBAL;
}

// ---- small device helpers ------------------------------------------------------------------
__device__ __forceinline__ void lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  asm volatile("" ::: "memory");
}

// Data-parallel-primitive lane shifts (VALU only, no LDS traffic).  row_shr:n moves a value n
// lanes up inside each 16-lane row; lanes without a source get `fill`.
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ uint32_t dpp_move(uint32_t fill, uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)v, CTRL, ROW_MASK, 0xF, false);
}

// Inclusive add-scan inside groups of W consecutive lanes (W = 8, 16, 32 or 64).
template <int W> __device__ __forceinline__ uint32_t group_incl_scan(uint32_t v) {
  uint32_t t;
  if constexpr (W == 8) {
    const uint32_t l = threadIdx.x & 7u;
    t = dpp_move<0x111>(0u, v); if (l >= 1) v += t;
    t = dpp_move<0x112>(0u, v); if (l >= 2) v += t;
    t = dpp_move<0x114>(0u, v); if (l >= 4) v += t;
  } else {
    v += dpp_move<0x111>(0u, v);
    v += dpp_move<0x112>(0u, v);
    v += dpp_move<0x114>(0u, v);
    v += dpp_move<0x118>(0u, v);
    if constexpr (W >= 32) v += dpp_move<0x142, 0xA>(0u, v); // row_bcast:15 into rows 1 and 3
    if constexpr (W >= 64) v += dpp_move<0x143, 0xC>(0u, v); // row_bcast:31 into rows 2 and 3
  }
  return v;
}
// value of the last lane of the group
template <int W> __device__ __forceinline__ uint32_t group_last(uint32_t v) {
  if constexpr (W == 64) return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
  else return __shfl(v, W - 1, W);
}
template <int W> __device__ __forceinline__ uint32_t group_sum(uint32_t v) {
#pragma unroll
  for (int d = W / 2; d > 0; d >>= 1) v += __shfl_xor(v, d, W);
  return v;
}
template <int W> __device__ __forceinline__ uint64_t group_sum64(uint64_t v) {
#pragma unroll
  for (int d = W / 2; d > 0; d >>= 1) v += __shfl_xor(v, d, W);
  return v;
}
template <int W> __device__ __forceinline__ uint32_t group_min(uint32_t v) {
#pragma unroll
  for (int d = W / 2; d > 0; d >>= 1) v = min(v, (uint32_t)__shfl_xor(v, d, W));
  return v;
}
template <int W> __device__ __forceinline__ uint32_t group_max(uint32_t v) {
#pragma unroll
  for (int d = W / 2; d > 0; d >>= 1) v = max(v, (uint32_t)__shfl_xor(v, d, W));
  return v;
}

// Workgroups b and b+8 share an XCD (round-robin dispatch).  Map the hardware block id to a
// logical id so that every XCD works on one contiguous range of rows: neighbouring rows read
// the same B rows, which then stay in that XCD's L2.  Bijective for every nb.
__device__ __forceinline__ uint32_t xcd_remap(uint32_t b, uint32_t nb) {
  uint32_t q = nb >> 3, r = nb & 7u, x = b & 7u, j = b >> 3;
  uint32_t start = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  return start + j;
}

__device__ __forceinline__ uint32_t hash_slot(uint32_t c, uint32_t log2s) {
  return (c * 0x9E3779B1u) >> (32u - log2s);
}

// LDS fp64 accumulate: ds_add_f64 (plain IEEE add of an already rounded product).
__device__ __forceinline__ void lds_add_f64(double *p, double v) { unsafeAtomicAdd(p, v); }

// ---- B row info: {min col, max col, length, 1 if the row's columns are not strictly increasing}
// of every B row; info[rows].x counts the rows that are not strictly increasing (zero on entry).
// 8 lanes per row.
__global__ void __launch_bounds__(256) k_b_rowinfo(uint64_t rows, const uint64_t *__restrict__ b_ptr,
                                                   const uint32_t *__restrict__ b_col,
                                                   uint4 *__restrict__ info) {
  constexpr int G = 8;
  uint64_t row = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
  uint32_t lg = threadIdx.x & (G - 1);
  uint32_t mn = 0xFFFFFFFFu, mx = 0, unsorted = 0;
  uint64_t b0 = 0, b1 = 0;
  if (row < rows) {
    b0 = b_ptr[row];
    b1 = b_ptr[row + 1];
  }
  for (uint64_t j = b0 + lg; j < b1; j += G) {
    uint32_t c = b_col[j];
    if (j > b0 && b_col[j - 1] >= c) unsorted = 1;
    mn = min(mn, c);
    mx = max(mx, c);
  }
  mn = group_min<G>(mn);
  mx = group_max<G>(mx);
  unsorted = group_max<G>(unsorted);
  if (row < rows && lg == 0) {
    uint64_t len = b1 - b0;
    info[row] = make_uint4(mn, mx, len > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)len, unsorted);
    if (unsorted) atomicAdd(&info[rows].x, 1u);
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
};

__device__ __forceinline__ void analyze_finish(uint32_t row, uint64_t np, uint32_t mn, uint32_t mx,
                                               uint32_t len_a, const AnalyzeOut &o,
                                               const ClassParams &cp) {
  uint32_t span = np ? (mx - mn + 1u) : 0u;
  uint32_t key = sym_key_of(np, span, len_a, cp);
  uint64_t ub = np;
  if (key != KEY_NONE && (key < (uint32_t)SYM_HASH0 || key >= (uint32_t)SYM_GLOBAL0))
    ub = np < span ? np : span;
  o.r_cmin[row] = mn;
  o.r_cmax[row] = mx;
  o.r_ub[row] = (uint32_t)ub;
  o.r_key[row] = (uint8_t)key;
  if (o.r_nprod) o.r_nprod[row] = np;
}

template <int G>
__global__ void __launch_bounds__(256)
    k_analyze(uint32_t rows_a, const uint64_t *__restrict__ a_ptr, const uint32_t *__restrict__ a_col,
              const uint4 *__restrict__ b_info, AnalyzeOut o, ClassParams cp, uint32_t long_len) {
  uint32_t row = (blockIdx.x * blockDim.x + threadIdx.x) / G;
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
  if (!is_long) {
    // does the row repeat the previous row's column list?
    uint64_t p0 = 0;
    bool cmp = false;
    if (row > 0 && row < rows_a) {
      p0 = a_ptr[row - 1];
      cmp = (a0 - p0) == (a1 - a0) && a1 > a0;
    }
    differs = cmp ? 0u : 1u;
    for (uint64_t ka = a0 + lg; ka < a1; ka += G) {
      uint32_t kcol = a_col[ka];
      if (cmp && a_col[p0 + (ka - a0)] != kcol) differs = 1;
      uint4 bi = b_info[kcol];
      if (bi.z) {
        np += bi.z;
        mn = min(mn, bi.x);
        mx = max(mx, bi.y);
      }
    }
  }
  differs = group_max<G>(differs);
  np = group_sum64<G>(np);
  mn = group_min<G>(mn);
  mx = group_max<G>(mx);
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
                   const uint4 *__restrict__ b_info, AnalyzeOut o, ClassParams cp) {
  __shared__ uint64_t s_np[4];
  __shared__ uint32_t s_mn[4], s_mx[4];
  uint32_t nlong = *o.long_count;
  for (uint32_t idx = blockIdx.x; idx < nlong; idx += gridDim.x) {
    uint32_t row = o.long_rows[idx];
    uint64_t a0 = a_ptr[row], a1 = a_ptr[row + 1];
    uint64_t np = 0;
    uint32_t mn = 0xFFFFFFFFu, mx = 0;
    for (uint64_t ka = a0 + threadIdx.x; ka < a1; ka += blockDim.x) {
      uint4 bi = b_info[a_col[ka]];
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
            uint32_t *__restrict__ lead, uint32_t enable) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows) return;
  lead[i] = i;
  const uint32_t k = key[i];
  const uint64_t len = a_ptr[i + 1] - a_ptr[i];
  if (!enable || k >= (uint32_t)SYM_HASH0 || len > GRP_ALEN_WIDE) { // not a window row (or no products)
    gcnt[i] = 1;
    return;
  }
  uint32_t pos = 0; // rows before i that belong to the same run
  while (pos < 56u && pos < i && same[i - pos]) ++pos;
  if (pos >= 56u) {
    gcnt[i] = 1;
    return;
  }
  const uint32_t gmax = group_max_rows(len); // the rows of a run have the same length
  if (pos % gmax != 0) { // follower: handled by its leader
    gcnt[i] = 0;
    key[i] = (uint8_t)KEY_NONE;
    ub[i] = 0;
    lead[i] = i - pos % gmax;
    return;
  }
  uint32_t c = 1;
  while (c < gmax && i + c < rows && same[i + c]) ++c;
  gcnt[i] = (uint8_t)c;
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

// out[i] = exclusive prefix; out[n] = total.  Items are laid out thread-contiguous here so the
// per-thread prefix is a running sum.
__global__ void __launch_bounds__(SCAN_BLOCK)
    k_scan_final(const uint32_t *__restrict__ in, uint64_t n, const uint64_t *__restrict__ partial,
                 uint64_t *__restrict__ out, uint32_t nparts) {
  __shared__ uint64_t s_wave[4];
  __shared__ uint32_t s_items[SCAN_BLOCK * SCAN_ITEMS];
  uint64_t base = (uint64_t)blockIdx.x * SCAN_BLOCK * SCAN_ITEMS;
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
  uint64_t ex = block_excl_scan64(s, s_wave, nullptr) + partial[blockIdx.x];
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    uint64_t idx = base + (uint64_t)threadIdx.x * SCAN_ITEMS + i;
    if (idx < n) out[idx] = ex;
    ex += v[i];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = partial[nparts];
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
                   const uint64_t *__restrict__ blk_off, uint32_t *__restrict__ rows_out) {
  __shared__ uint32_t s_base[MAX_BINS];
  __shared__ uint32_t s_wcnt[4][MAX_BINS];
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

// bin_off[b] = start of bin b in rows_out (b = 0..nbins), read from the scanned histogram.
__global__ void k_part_binoff(const uint64_t *__restrict__ blk_off, uint32_t nbins, uint32_t nblocks,
                              uint64_t *__restrict__ bin_off) {
  uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b <= nbins) bin_off[b] = blk_off[(size_t)b * nblocks];
}

// ---- common per-row context ------------------------------------------------------------------
struct Csr {
  const uint64_t *__restrict__ ptr;
  const uint32_t *__restrict__ col;
  const double *__restrict__ val;
};

struct RowMeta {
  const uint32_t *__restrict__ cmin;
  const uint32_t *__restrict__ cmax;
  const uint8_t *__restrict__ sym_key;
  const uint64_t *__restrict__ t_off; // offset of the row's slot in temp_cols
  uint32_t *__restrict__ temp_cols;   // sorted structural columns per row (written by symbolic)
  uint32_t *__restrict__ nnz;         // structural nnz per row
  uint8_t *__restrict__ num_key;      // numeric bin key per row
  const uint8_t *__restrict__ gcnt;   // rows in the group led by this row (0: follower)
  const uint32_t *__restrict__ lead;  // leader of the row's group (the row itself when ungrouped)
  uint32_t hash_num_max;              // see num_key_of
};

// ---- walking the products of one row ----------------------------------------------------------
// Wave-uniform broadcast of a 64-bit lane value (index uniform across the wavefront).
__device__ __forceinline__ uint64_t readlane64(uint64_t v, uint32_t l) {
  uint32_t lo = __builtin_amdgcn_readlane((int)(uint32_t)v, (int)l);
  uint32_t hi = __builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), (int)l);
  return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ double readlane_f64(double v, uint32_t l) {
  return __longlong_as_double((long long)readlane64((uint64_t)__double_as_longlong(v), l));
}

#ifndef SG_PF
#define SG_PF 4
#endif
constexpr int PF = SG_PF; // B rows whose first 64 entries are fetched before any of them is consumed

// Wave-per-row walk with the A row handled on the scalar unit.  `row` must be wave-uniform
// (derived through readfirstlane): the A entries, their values and the bounds of the B rows
// are then fetched by scalar loads and live in SGPRs, the B row base is a scalar pointer and a
// lane addresses its entry as base + lane * size, so the vector ALU does no index arithmetic
// at all.  The first 64 entries of PF consecutive B rows are fetched back to back; each B row
// is consumed completely, in A order, before the next one (the order of the CPU loop).
// f(col, bval, aval): bval/aval are only loaded when WITH_VAL.
// Read-only kernel inputs seen through the constant address space: a load whose address is
// wave-uniform is then issued on the scalar unit (s_load) instead of the vector memory path.
template <class T> using const_as_ptr = const T __attribute__((address_space(4))) *;
template <class T> __device__ __forceinline__ const_as_ptr<T> scalar_view(const T *p) {
  return (const_as_ptr<T>)(uintptr_t)p;
}

// f(col, bval, aval, apos): apos = position of the A entry inside its row (wave-uniform).
//
// The walk is software-pipelined over groups of PF A entries: while group g is consumed from
// registers, the vector loads of group g+1 are in flight and the scalar loads of group g+2 (A
// entries, then B row bounds) have been issued, so neither latency is exposed.
template <bool WITH_AVAL> struct WalkInfo {
  uint64_t p0[PF];   // start of the B row in B.col / B.val
  uint32_t len[PF];  // entries of the B row (0 past the end of the A row)
  double av[PF];     // value of the A entry
};

template <bool WITH_AVAL>
__device__ __forceinline__ void walk_scalar_stage(const Csr &A, const Csr &B, uint64_t ka, uint64_t a1,
                                                  WalkInfo<WITH_AVAL> &s) {
  const auto a_col = scalar_view(A.col);
  const auto a_val = scalar_view(A.val);
  const auto b_ptr = scalar_view(B.ptr);
  if (ka >= a1) {
#pragma unroll
    for (int u = 0; u < PF; ++u) { s.len[u] = 0; s.p0[u] = 0; s.av[u] = 0.0; }
    return;
  }
  uint32_t k[PF];
  if (ka + PF <= a1) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      k[u] = a_col[ka + u];
      if constexpr (WITH_AVAL) s.av[u] = a_val[ka + u];
    }
  } else { // the last group re-reads the row's last entry for its missing slots
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const uint64_t kk = min(ka + u, a1 - 1);
      k[u] = a_col[kk];
      if constexpr (WITH_AVAL) s.av[u] = a_val[kk];
    }
  }
  uint64_t p1[PF];
#pragma unroll
  for (int u = 0; u < PF; ++u) {
    s.p0[u] = b_ptr[k[u]];
    p1[u] = b_ptr[k[u] + 1];
  }
#pragma unroll
  for (int u = 0; u < PF; ++u) s.len[u] = (ka + u < a1) ? (uint32_t)(p1[u] - s.p0[u]) : 0u;
}

template <bool WITH_VAL, bool WITH_AVAL>
__device__ __forceinline__ void walk_vector_stage(const Csr &B, const WalkInfo<WITH_AVAL> &s, uint32_t lane,
                                                  uint32_t (&c)[PF], double (&v)[PF]) {
#pragma unroll
  for (int u = 0; u < PF; ++u) {
    const bool ok = lane < s.len[u];
    c[u] = ok ? (B.col + s.p0[u])[lane] : EMPTY_KEY;
    if constexpr (WITH_VAL) v[u] = ok ? (B.val + s.p0[u])[lane] : 0.0;
  }
}

// begin() issues the first loads of the walk (two scalar stages and one vector stage) so that a
// kernel can overlap them with its LDS set-up; run() consumes the row.
template <bool WITH_VAL, bool WITH_AVAL = WITH_VAL> struct RowWalk {
  uint64_t a0, a1;
  WalkInfo<WITH_AVAL> s_cur, s_next;
  uint32_t c_cur[PF];
  double v_cur[PF];

  __device__ __forceinline__ void begin(const Csr &A, const Csr &B, uint32_t row, uint32_t lane) {
    const auto a_ptr = scalar_view(A.ptr);
    a0 = a_ptr[row];
    a1 = a_ptr[row + 1];
    walk_scalar_stage<WITH_AVAL>(A, B, a0, a1, s_cur);
    walk_vector_stage<WITH_VAL, WITH_AVAL>(B, s_cur, lane, c_cur, v_cur);
    walk_scalar_stage<WITH_AVAL>(A, B, a0 + PF, a1, s_next);
  }

  template <class F> __device__ __forceinline__ void run(const Csr &A, const Csr &B, uint32_t lane, F f) {
    WalkInfo<WITH_AVAL> s_after;
    uint32_t c_next[PF];
    double v_next[PF];
    for (uint64_t ka = a0; ka < a1; ka += PF) {
      walk_vector_stage<WITH_VAL, WITH_AVAL>(B, s_next, lane, c_next, v_next);
      walk_scalar_stage<WITH_AVAL>(A, B, ka + 2 * PF, a1, s_after);
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const uint32_t apos = (uint32_t)(ka - a0) + u;
        if (c_cur[u] != EMPTY_KEY) f(c_cur[u], WITH_VAL ? v_cur[u] : 0.0, WITH_AVAL ? s_cur.av[u] : 0.0, apos);
        if (s_cur.len[u] > 64) { // wave-uniform: the rest of a long B row, before the next A entry
          const uint32_t *pc = B.col + s_cur.p0[u];
          const double *pv = B.val + s_cur.p0[u];
          for (uint32_t j = 64 + lane; j < s_cur.len[u]; j += 64)
            f(pc[j], WITH_VAL ? pv[j] : 0.0, WITH_AVAL ? s_cur.av[u] : 0.0, apos);
        }
      }
      s_cur = s_next;
      s_next = s_after;
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        c_cur[u] = c_next[u];
        v_cur[u] = v_next[u];
      }
    }
  }
};

template <bool WITH_VAL, bool WITH_AVAL = WITH_VAL, class F>
__device__ __forceinline__ void wave_row_walk(const Csr &A, const Csr &B, uint32_t row, uint32_t lane, F f) {
  RowWalk<WITH_VAL, WITH_AVAL> w;
  w.begin(A, B, row, lane);
  w.run(A, B, lane, f);
}

// ---- flat walk ---------------------------------------------------------------------------------
// For rows whose B rows are short, one A entry per step leaves most lanes idle.  The flat walk
// lays the products of up to 64 A entries end to end (A order, then B row order) and hands 64
// consecutive products to the 64 lanes per step, whatever B rows they come from: lane l of a step
// holds product q = q0 + l and finds its A entry by a binary search over the exclusive prefix
// sums of the B row lengths (kept in LDS).  Products of one step are in CPU order by lane, and
// steps are sequential, so a consumer that resolves two lanes hitting one accumulator in lane
// order (claim_in_lane_order) still adds in the CPU order.
// LDS scratch per wavefront: st[64] u32 | p0s[64] u64 | avs[64] f64 = FLAT_BYTES.
// f(col, bval, aval, apos) is called by ALL lanes every step; col == EMPTY_KEY marks an idle lane.
constexpr uint32_t FLAT_BYTES = 1280;
#ifndef SG_FLAT_PF
#define SG_FLAT_PF 3
#endif
constexpr int FLAT_PF = SG_FLAT_PF; // steps of the flat walk whose loads are in flight ahead of the consumed one

template <bool WITH_VAL, class F>
__device__ __forceinline__ void flat_walk(const Csr &A, const Csr &B, uint32_t row, uint32_t lane,
                                          unsigned char *scratch, F f) {
  uint32_t *st = reinterpret_cast<uint32_t *>(scratch);
  uint64_t *p0s = reinterpret_cast<uint64_t *>(scratch + 256);
  double *avs = reinterpret_cast<double *>(scratch + 768);
  const uint64_t a0 = scalar_view(A.ptr)[row], a1 = scalar_view(A.ptr)[row + 1];
  for (uint64_t ab = a0; ab < a1; ab += 64) {
    const uint64_t ka = ab + lane;
    uint32_t len = 0;
    uint64_t p0 = 0;
    double av = 0.0;
    if (ka < a1) {
      const uint32_t k = A.col[ka];
      p0 = B.ptr[k];
      len = (uint32_t)(B.ptr[k + 1] - p0);
      if constexpr (WITH_VAL) av = A.val[ka];
    }
    const uint32_t incl = group_incl_scan<64>(len);
    const uint32_t total = group_last<64>(incl);
    lds_fence(); // the previous chunk has been consumed
    st[lane] = incl - len;
    p0s[lane] = p0;
    if constexpr (WITH_VAL) avs[lane] = av;
    lds_fence();
    auto fetch = [&](uint32_t q, uint32_t &t, uint32_t &c, double &bv) {
      const bool valid = q < total;
      const uint32_t qq = valid ? q : 0u;
      t = 0;
#pragma unroll
      for (uint32_t s = 32; s > 0; s >>= 1)
        if (st[t + s] <= qq) t += s; // largest t with st[t] <= q: the entry that holds product q
      const uint64_t p = p0s[t] + (qq - st[t]);
      c = valid ? B.col[p] : EMPTY_KEY;
      if constexpr (WITH_VAL) bv = valid ? B.val[p] : 0.0;
    };
    // FLAT_PF steps are in flight ahead of the one that is consumed
    uint32_t t_q[FLAT_PF], c_q[FLAT_PF];
    double bv_q[FLAT_PF];
#pragma unroll
    for (int d = 0; d < FLAT_PF; ++d) {
      t_q[d] = 0; c_q[d] = EMPTY_KEY; bv_q[d] = 0.0;
      if (64u * d < total) fetch(64u * d + lane, t_q[d], c_q[d], bv_q[d]);
    }
    for (uint32_t q0 = 0; q0 < total; q0 += 64) {
      const uint32_t t_cur = t_q[0], c_cur = c_q[0];
      const double bv_cur = bv_q[0];
#pragma unroll
      for (int d = 0; d + 1 < FLAT_PF; ++d) { t_q[d] = t_q[d + 1]; c_q[d] = c_q[d + 1]; bv_q[d] = bv_q[d + 1]; }
      t_q[FLAT_PF - 1] = 0; c_q[FLAT_PF - 1] = EMPTY_KEY; bv_q[FLAT_PF - 1] = 0.0;
      if (q0 + 64u * FLAT_PF < total) fetch(q0 + 64u * FLAT_PF + lane, t_q[FLAT_PF - 1], c_q[FLAT_PF - 1], bv_q[FLAT_PF - 1]);
      double av_cur = 0.0;
      if constexpr (WITH_VAL) av_cur = avs[t_cur];
      f(c_cur, bv_cur, av_cur, (uint32_t)(ab - a0) + t_cur);
    }
  }
}

// Add `prod` to vals[p] for every lane with pending == true so that lanes of this call that share
// one p add in lane order.  owner[] (one uint32 per accumulator, zero at kernel start) takes
// claims (stamp << 6 | 63 - lane) by atomic max: the lowest pending lane of the newest round wins
// its accumulator, adds, and the others try again in the next round.  stamp is wave-uniform.
__device__ __forceinline__ void claim_in_lane_order(double *vals, uint32_t *owner, uint32_t &stamp, uint32_t lane,
                                                    bool pending, uint32_t p, double prod) {
  for (;;) {
    ++stamp;
    const uint32_t tag = (stamp << 6) | (63u - lane);
    if (pending) atomicMax(&owner[p], tag);
    lds_fence();
    if (pending && owner[p] == tag) {
      lds_add_f64(&vals[p], prod);
      pending = false;
    }
    if (!__any(pending)) break;
    lds_fence();
  }
}

// r-th (0-based) set bit of a 32-bit word that has more than r bits set
__device__ __forceinline__ uint32_t select_bit(uint32_t bits, uint32_t r) {
  uint32_t base = 0, c;
  c = __popc(bits & 0xFFFFu); if (r >= c) { r -= c; bits >>= 16; base += 16; }
  c = __popc(bits & 0xFFu);   if (r >= c) { r -= c; bits >>= 8;  base += 8; }
  c = __popc(bits & 0xFu);    if (r >= c) { r -= c; bits >>= 4;  base += 4; }
  c = __popc(bits & 0x3u);    if (r >= c) { r -= c; bits >>= 2;  base += 2; }
  c = bits & 1u;              if (r >= c) { base += 1; }
  return base;
}

// ---- symbolic, window path: one wavefront per row ------------------------------------------------
// LDS per wavefront: `words` pairs {bitmap word, number of set bits before it} covering columns
// [cmin & ~31, cmax].
//
// Every product ORs its bit into the bitmap (lanes of one instruction that share a word are
// serialised by the LDS; combining them first in registers was measured and cost more than it
// saved).  The sorted column list is then written coalesced: output position p finds the word
// that holds the p-th set bit by a binary search over the per-word prefix counts.
// FLAT: flat walk (short B rows) instead of the scalar-unit walk; adds FLAT_BYTES of LDS per wavefront.
template <bool FLAT>
__global__ void __launch_bounds__(256)
    k_sym_window(const uint32_t *__restrict__ rows, uint32_t nrows, Csr A, Csr B, RowMeta m,
                 uint32_t words) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint32_t lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
  const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // wave-uniform on the scalar unit
  uint32_t idx = xcd_remap(blockIdx.x, gridDim.x) * wpb + w;
  if (idx >= nrows) return;
  const size_t region = (size_t)words * 8u + (FLAT ? FLAT_BYTES : 0u);
  uint32_t *bm = reinterpret_cast<uint32_t *>(smem + w * region);
  const uint32_t row = rows[idx];
  const uint32_t base = m.cmin[row] & ~31u;
  const uint32_t nw = ((m.cmax[row] - base) >> 5) + 1u;
  for (uint32_t i = lane; i < nw; i += 64) reinterpret_cast<uint64_t *>(bm)[i] = 0ull;
  lds_fence();
  auto mark = [&](uint32_t col, double, double, uint32_t) {
    if (col == EMPTY_KEY) return;
    uint32_t c = col - base;
    atomicOr(&bm[(c >> 5) * 2u], 1u << (c & 31u));
  };
  if constexpr (FLAT) flat_walk<false>(A, B, row, lane, smem + w * region + (size_t)words * 8u, mark);
  else wave_row_walk<false>(A, B, row, lane, mark);
  lds_fence();
  // per-word prefix of the set-bit counts
  uint32_t carry = 0;
  for (uint32_t r0 = 0; r0 < nw; r0 += 64) {
    uint32_t wi = r0 + lane;
    uint32_t pc = wi < nw ? __popc(bm[wi * 2u]) : 0u;
    uint32_t incl = group_incl_scan<64>(pc);
    if (wi < nw) bm[wi * 2u + 1u] = carry + incl - pc;
    carry += group_last<64>(incl);
  }
  lds_fence();
  // output position p -> word holding the p-th set bit (binary search on the prefixes) -> column
  const uint32_t n = carry;
  uint32_t *out = m.temp_cols + m.t_off[row];
  for (uint32_t p = lane; p < n; p += 64) {
    uint32_t lo = 0, hi = nw - 1u;
    while (lo < hi) {
      uint32_t mid = (lo + hi + 1u) >> 1;
      if (bm[mid * 2u + 1u] <= p) lo = mid; else hi = mid - 1u;
    }
    uint2 e = *reinterpret_cast<const uint2 *>(&bm[lo * 2u]);
    out[p] = base + (lo << 5) + select_bit(e.x, p - e.y);
  }
  // the group's rows share the pattern: same nnz; they are accumulated together when the group
  // kernel has a tier for them, else one by one (each reading the leader's column list)
  const uint32_t cnt = m.gcnt[row];
  if (lane < cnt) {
    m.nnz[row + lane] = n;
    const bool wide = A.ptr[row + 1] - A.ptr[row] > GRP_ALEN;
    const uint32_t gkey = num_key_of(m.sym_key[row], n, m.hash_num_max, cnt, wide);
    const bool grouped = gkey >= (uint32_t)NUM_GRP0 && gkey != KEY_NONE;
    m.num_key[row + lane] = (uint8_t)(grouped ? (lane == 0 ? gkey : KEY_NONE) : num_key_of(m.sym_key[row], n, m.hash_num_max));
  }
}

// ---- symbolic, hash path: one wavefront per row ----------------------------------------------
// LDS per wavefront: keys[S] | cnt[S/2] | tk[S/2]  (uint32 each) = 8*S bytes, S = 1 << log2s.
// S >= 2 * nprod, so the table is at most half full.  Sorting is a bucket rank sort: a monotone
// map of the column into S/2 buckets, counting sort into bucket order, then each key finds its
// rank among the (few) keys of its bucket.
template <bool FLAT>
__global__ void __launch_bounds__(256)
    k_sym_hash(const uint32_t *__restrict__ rows, uint32_t nrows, Csr A, Csr B, RowMeta m,
               uint32_t log2s) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint32_t lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
  const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  uint32_t idx = xcd_remap(blockIdx.x, gridDim.x) * wpb + w;
  if (idx >= nrows) return;
  const uint32_t S = 1u << log2s, NB = S >> 1, mask = S - 1u;
  const size_t region = (size_t)S * 8u + (FLAT ? FLAT_BYTES : 0u);
  uint32_t *keys = reinterpret_cast<uint32_t *>(smem + w * region);
  uint32_t *cnt = keys + S;
  uint32_t *tk = cnt + NB;
  const uint32_t row = rows[idx];
  for (uint32_t i = lane; i < S; i += 64) keys[i] = EMPTY_KEY;
  for (uint32_t i = lane; i < NB; i += 64) cnt[i] = 0u;
  lds_fence();
  uint32_t mine = 0;
  auto insert = [&](uint32_t c, double, double, uint32_t) {
    if (c == EMPTY_KEY) return;
    uint32_t h = hash_slot(c, log2s);
    for (uint32_t it = 0; it < S; ++it) { // bounded: the table is at most half full
      uint32_t old = atomicCAS(&keys[h], EMPTY_KEY, c);
      if (old == EMPTY_KEY) { ++mine; break; }
      if (old == c) break;
      h = (h + 1u) & mask;
    }
  };
  if constexpr (FLAT) flat_walk<false>(A, B, row, lane, smem + w * region + (size_t)S * 8u, insert);
  else wave_row_walk<false>(A, B, row, lane, insert);
  lds_fence();
  const uint32_t n = group_sum<64>(mine);
  const uint32_t cmin = m.cmin[row];
  const float scale = (float)NB / (float)(m.cmax[row] - cmin + 1u);
  // pass A: bucket counts
  for (uint32_t s = lane; s < S; s += 64) {
    uint32_t key = keys[s];
    if (key != EMPTY_KEY) {
      uint32_t b = min(NB - 1u, (uint32_t)((float)(key - cmin) * scale));
      atomicAdd(&cnt[b], 1u);
    }
  }
  lds_fence();
  // exclusive scan of cnt
  uint32_t carry = 0;
  for (uint32_t r0 = 0; r0 < NB; r0 += 64) {
    uint32_t v = cnt[r0 + lane]; // NB is a multiple of 64
    uint32_t incl = group_incl_scan<64>(v);
    cnt[r0 + lane] = carry + incl - v;
    carry += group_last<64>(incl);
  }
  lds_fence();
  // pass B: scatter into bucket order; afterwards cnt[b] = end of bucket b
  for (uint32_t s = lane; s < S; s += 64) {
    uint32_t key = keys[s];
    if (key != EMPTY_KEY) {
      uint32_t b = min(NB - 1u, (uint32_t)((float)(key - cmin) * scale));
      uint32_t p = atomicAdd(&cnt[b], 1u);
      tk[p] = key;
    }
  }
  lds_fence();
  // pass C: rank inside the bucket, write sorted
  uint32_t *out = m.temp_cols + m.t_off[row];
  for (uint32_t p = lane; p < n; p += 64) {
    uint32_t key = tk[p];
    uint32_t b = min(NB - 1u, (uint32_t)((float)(key - cmin) * scale));
    uint32_t start = b ? cnt[b - 1] : 0u, end = cnt[b];
    uint32_t rank = start;
    for (uint32_t q = start; q < end; ++q) rank += (tk[q] < key) ? 1u : 0u;
    out[rank] = key;
  }
  if (lane == 0) {
    m.nnz[row] = n;
    m.num_key[row] = (uint8_t)num_key_of(m.sym_key[row], n, m.hash_num_max);
  }
}

// ---- symbolic, hash path, large tables: one 256-thread workgroup per row ---------------------------
// The same algorithm as k_sym_hash with the four wavefronts of a workgroup sharing one table (the
// pattern of a row does not depend on the order in which its products are seen): a table of 8192
// slots takes 64 KiB of LDS, and one wavefront per table would leave two wavefronts on a CU.
// The wavefronts take the A entries round robin.  LDS: keys[S] | cnt[S/2] | tk[S/2] = 8*S bytes.
__global__ void __launch_bounds__(256)
    k_sym_hash_wg(const uint32_t *__restrict__ rows, uint32_t nrows, Csr A, Csr B, RowMeta m, uint32_t log2s) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ uint32_t s_n, s_wsum[4];
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
  const uint32_t S = 1u << log2s, NB = S >> 1, mask = S - 1u;
  uint32_t *keys = reinterpret_cast<uint32_t *>(smem);
  uint32_t *cnt = keys + S;
  uint32_t *tk = cnt + NB;
  const uint32_t row = scalar_view(rows)[xcd_remap(blockIdx.x, gridDim.x)];
  for (uint32_t i = tid; i < S; i += 256) keys[i] = EMPTY_KEY;
  for (uint32_t i = tid; i < NB; i += 256) cnt[i] = 0u;
  if (tid == 0) s_n = 0u;
  __syncthreads();
  const uint64_t a0 = scalar_view(A.ptr)[row], a1 = scalar_view(A.ptr)[row + 1];
  uint32_t mine = 0;
  for (uint64_t ka = a0 + w; ka < a1; ka += 4) {
    const uint32_t k = scalar_view(A.col)[ka];
    const uint64_t b0 = scalar_view(B.ptr)[k], b1 = scalar_view(B.ptr)[k + 1];
    for (uint64_t j = b0 + lane; j < b1; j += 64) {
      const uint32_t c = B.col[j];
      uint32_t h = hash_slot(c, log2s);
      for (uint32_t it = 0; it < S; ++it) { // bounded: the table is at most half full
        const uint32_t old = atomicCAS(&keys[h], EMPTY_KEY, c);
        if (old == EMPTY_KEY) { ++mine; break; }
        if (old == c) break;
        h = (h + 1u) & mask;
      }
    }
  }
  mine = group_last<64>(group_incl_scan<64>(mine));
  if (lane == 0 && mine) atomicAdd(&s_n, mine);
  __syncthreads();
  const uint32_t n = s_n;
  const uint32_t cmin = m.cmin[row];
  const float scale = (float)NB / (float)(m.cmax[row] - cmin + 1u);
  // pass A: bucket counts (buckets are monotone in the column)
  for (uint32_t s = tid; s < S; s += 256) {
    const uint32_t key = keys[s];
    if (key != EMPTY_KEY) atomicAdd(&cnt[min(NB - 1u, (uint32_t)((float)(key - cmin) * scale))], 1u);
  }
  __syncthreads();
  // exclusive scan of cnt: NB / 256 consecutive buckets per thread (NB is a multiple of 256 here)
  const uint32_t per = NB >> 8;
  uint32_t local = 0;
  for (uint32_t i = 0; i < per; ++i) local += cnt[tid * per + i];
  const uint32_t incl = group_incl_scan<64>(local);
  if (lane == 63) s_wsum[w] = incl;
  __syncthreads();
  uint32_t run = incl - local;
  for (uint32_t i = 0; i < w; ++i) run += s_wsum[i];
  for (uint32_t i = 0; i < per; ++i) {
    const uint32_t v = cnt[tid * per + i];
    cnt[tid * per + i] = run;
    run += v;
  }
  __syncthreads();
  // pass B: scatter into bucket order; afterwards cnt[b] = end of bucket b
  for (uint32_t s = tid; s < S; s += 256) {
    const uint32_t key = keys[s];
    if (key != EMPTY_KEY) tk[atomicAdd(&cnt[min(NB - 1u, (uint32_t)((float)(key - cmin) * scale))], 1u)] = key;
  }
  __syncthreads();
  // pass C: rank inside the bucket, write sorted
  uint32_t *out = m.temp_cols + m.t_off[row];
  for (uint32_t p = tid; p < n; p += 256) {
    const uint32_t key = tk[p];
    const uint32_t b = min(NB - 1u, (uint32_t)((float)(key - cmin) * scale));
    const uint32_t start = b ? cnt[b - 1] : 0u, end = cnt[b];
    uint32_t rank = start;
    for (uint32_t q = start; q < end; ++q) rank += (tk[q] < key) ? 1u : 0u;
    out[rank] = key;
  }
  if (tid == 0) {
    m.nnz[row] = n;
    m.num_key[row] = (uint8_t)num_key_of(m.sym_key[row], n, m.hash_num_max);
  }
}

// ---- symbolic, global path: one 256-thread workgroup per row, bitmap in global scratch ---------
// scratch: per workgroup `words_total` uint32 words, all zero on entry and on exit.
__global__ void __launch_bounds__(256)
    k_sym_global(const uint32_t *__restrict__ rows, uint32_t nrows, Csr A, Csr B, RowMeta m,
                 uint32_t *__restrict__ scratch, uint64_t words_total) {
  __shared__ uint64_t s_wave[4];
  __shared__ uint32_t s_carry;
  uint32_t *bm = scratch + (size_t)blockIdx.x * words_total;
  const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (uint32_t idx = blockIdx.x; idx < nrows; idx += gridDim.x) {
    const uint32_t row = rows[idx];
    const uint32_t w0 = m.cmin[row] >> 5, w1 = m.cmax[row] >> 5;
    const uint64_t a0 = A.ptr[row], a1 = A.ptr[row + 1];
    for (uint64_t ka = a0 + w; ka < a1; ka += 4) {
      uint32_t k = A.col[ka];
      uint64_t b0 = B.ptr[k], b1 = B.ptr[k + 1];
      for (uint64_t jb = b0 + lane; jb < b1; jb += 64) {
        uint32_t c = B.col[jb];
        atomicOr(&bm[c >> 5], 1u << (c & 31u));
      }
    }
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    uint32_t *out = m.temp_cols + m.t_off[row];
    for (uint32_t r0 = w0; r0 <= w1; r0 += 256) {
      uint32_t wi = r0 + threadIdx.x;
      uint32_t bits = 0;
      if (wi <= w1) bits = atomicExch(&bm[wi], 0u); // read through L2 and clear
      uint32_t pc = __popc(bits);
      uint64_t tot;
      uint32_t ex = (uint32_t)block_excl_scan64(pc, s_wave, &tot);
      uint32_t pos = s_carry + ex;
      uint32_t cbase = wi << 5;
      while (bits) {
        uint32_t b = __ffs(bits) - 1;
        out[pos++] = cbase + b;
        bits &= bits - 1u;
      }
      __syncthreads();
      if (threadIdx.x == 0) s_carry += (uint32_t)tot;
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      uint32_t n = s_carry;
      m.nnz[row] = n;
      m.num_key[row] = (uint8_t)num_key_of(m.sym_key[row], n, m.hash_num_max);
    }
    __syncthreads();
  }
}

// ---- symbolic, big rows: one 1024-thread workgroup per row, bitmap windows in LDS ----------------
// Needs B rows with strictly increasing columns.  The row's column hull is cut into windows of
// win_words * 32 columns.  Every A entry keeps {position, end} of what is left of its B row (in
// LDS for rows of up to BIG_CUR entries, else in g_state, two uint64 per entry of A); in a window
// the part of the B row that falls into it (a "piece": contiguous, the B row is sorted) is ORed
// into the LDS bitmap and the position moves on.  Pieces are classified by what is left of the B
// row: up to BIG_MID entries -> one wavefront takes the piece, four pieces at a time, all their
// loads in flight together; more -> all 16 wavefronts take 1024 entries of the piece per step,
// four pieces at a time, and a wavefront stops at its first block that leaves the window.
// The bitmap is then written out in blocks of 256 words (four per lane, the blocks dealt to the
// wavefronts round robin): block totals, a scan over the blocks, and every lane writes the
// columns of its own four words, at most 128.
// LDS: win_words u32 bitmap | BIG_CUR u64 pos | BIG_CUR u64 end | BIG_CUR u32 list of mid pieces |
//      BIG_CUR u32 list of long pieces | BIG_CUR u32 entries taken from long pieces | win_words / 64 u32 (block offsets; a quarter used).
constexpr uint32_t BIG_THREADS = 1024, BIG_WAVES = 16, BIG_CUR = 1024, BIG_MID = 256;
static_assert(BIG_CUR == BIG_THREADS, "a batch of A entries is one entry per thread, and the piece lists hold one batch");
constexpr uint32_t BIG_WIN_MAX = 32768;                   // words of the largest window
constexpr uint32_t BIG_BPW = BIG_WIN_MAX / 256 / BIG_WAVES; // blocks of 256 words per wavefront and window
__host__ __device__ inline size_t sym_big_lds(uint32_t win_words) {
  return (size_t)win_words * 4 + (size_t)BIG_CUR * (8 + 8 + 4 + 4 + 4) + (size_t)(win_words / 64) * 4;
}

__global__ void __launch_bounds__(BIG_THREADS)
    k_sym_big(const uint32_t *__restrict__ rows, uint32_t nrows, Csr A, Csr B, RowMeta m, uint32_t win_words,
              uint64_t *__restrict__ g_state, uint64_t nnz_a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint32_t *bits = reinterpret_cast<uint32_t *>(smem);
  uint64_t *s_pos = reinterpret_cast<uint64_t *>(smem + (size_t)win_words * 4);
  uint64_t *s_end = s_pos + BIG_CUR;
  uint32_t *s_mid = reinterpret_cast<uint32_t *>(s_end + BIG_CUR);
  uint32_t *s_long = s_mid + BIG_CUR;
  uint32_t *s_cnt = s_long + BIG_CUR;
  uint32_t *s_blk = s_cnt + BIG_CUR; // per block of 64 words: set bits, then offset of its first column
  __shared__ uint32_t s_n[2]; // mid pieces, long pieces of the current batch
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
  const uint32_t row = scalar_view(rows)[blockIdx.x];
  const uint64_t a0 = scalar_view(A.ptr)[row], a1 = scalar_view(A.ptr)[row + 1];
  const uint32_t len_a = (uint32_t)(a1 - a0);
  const uint32_t w0 = m.cmin[row] >> 5, w1 = m.cmax[row] >> 5;
  uint64_t *pos = len_a <= BIG_CUR ? s_pos : g_state + a0;
  uint64_t *end = len_a <= BIG_CUR ? s_end : g_state + nnz_a + a0;
  uint32_t *out = m.temp_cols + m.t_off[row];
  uint32_t carry = 0;
  if (tid < 2) s_n[tid] = 0u;
  __syncthreads();
  for (uint32_t wb = w0; wb <= w1; wb += win_words) {
    const uint32_t nw = min(win_words, w1 + 1u - wb);
    const uint32_t nblk = (nw + 255u) >> 8;
    const bool first = wb == w0, last = wb + nw > w1;
    const uint64_t chi = (uint64_t)(wb + nw) << 5;
    if (first) { // later windows find the bitmap cleared by the write-out of the one before
      for (uint32_t w = tid * 4u; w < win_words; w += BIG_THREADS * 4u)
        *reinterpret_cast<uint4 *>(&bits[w]) = make_uint4(0u, 0u, 0u, 0u);
    }
    for (uint32_t kb = 0; kb < len_a; kb += BIG_CUR) {
      // classify what is left of every B row of this batch of A entries
      const uint32_t idx = kb + tid;
      if (idx < len_a) {
        uint64_t p, e;
        if (first) {
          const uint32_t k = A.col[a0 + idx];
          p = B.ptr[k];
          e = B.ptr[k + 1];
          pos[idx] = p;
          end[idx] = e;
        } else {
          p = pos[idx];
          e = end[idx];
        }
        const uint64_t rem = e - p;
        if (rem > BIG_MID) {
          const uint32_t slot = atomicAdd(&s_n[1], 1u);
          s_long[slot] = idx;
          s_cnt[slot] = 0u;
        } else if (rem) {
          s_mid[atomicAdd(&s_n[0], 1u)] = idx;
        }
      }
      __syncthreads();
      const uint32_t nmid = s_n[0], nlong = s_n[1];
      // mid pieces: a wavefront takes four pieces and all of their (up to 4 x 64) entries at once
      for (uint32_t e0 = wv * 4u; e0 < nmid; e0 += BIG_WAVES * 4u) {
        uint32_t id[4], c[4][4];
        uint64_t p[4];
        uint32_t rem[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const bool on = e0 + u < nmid;
          id[u] = s_mid[on ? e0 + u : e0];
          p[u] = pos[id[u]];
          rem[u] = on ? (uint32_t)(end[id[u]] - p[u]) : 0u;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int j = 0; j < 4; ++j) c[u][j] = (64u * j + lane < rem[u]) ? B.col[p[u] + 64u * j + lane] : EMPTY_KEY;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          uint32_t took = 0;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const bool in = 64u * j + lane < rem[u] && (last || (uint64_t)c[u][j] < chi);
            if (in) atomicOr(&bits[(c[u][j] >> 5) - wb], 1u << (c[u][j] & 31u));
            took += (uint32_t)__popcll(__ballot(in));
          }
          if (lane == 0 && rem[u]) pos[id[u]] = p[u] + took;
        }
      }
      // long pieces: all wavefronts, 1024 entries of a piece per step, four pieces at a time
      for (uint32_t e0 = 0; e0 < nlong; e0 += 4u) {
        uint32_t c[4];
        uint64_t p[4], rem[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const bool on = e0 + u < nlong;
          const uint32_t id = s_long[on ? e0 + u : e0];
          p[u] = pos[id];
          rem[u] = on ? end[id] - p[u] : 0ull;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) c[u] = tid < rem[u] ? B.col[p[u] + tid] : EMPTY_KEY;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const bool in = tid < rem[u] && (last || (uint64_t)c[u] < chi);
          if (in) atomicOr(&bits[(c[u] >> 5) - wb], 1u << (c[u] & 31u));
          uint32_t n = (uint32_t)__popcll(__ballot(in));
          uint32_t took = n;
          for (uint64_t base = BIG_THREADS; n == 64u && base + wv * 64u < rem[u]; base += BIG_THREADS) {
            const uint64_t j = base + tid;
            const bool have = j < rem[u];
            const uint32_t c2 = have ? B.col[p[u] + j] : EMPTY_KEY;
            const bool in2 = have && (last || (uint64_t)c2 < chi);
            if (in2) atomicOr(&bits[(c2 >> 5) - wb], 1u << (c2 & 31u));
            n = (uint32_t)__popcll(__ballot(in2));
            took += n;
          }
          if (lane == 0 && took) atomicAdd(&s_cnt[e0 + u], took);
        }
      }
      __syncthreads();
      if (tid < nlong) pos[s_long[tid]] += s_cnt[tid];
      if (tid < 2) s_n[tid] = 0u;
      if (kb + BIG_CUR < len_a) __syncthreads(); // the lists are filled again by the next batch
    }
    // set bits per block of 256 words (four per lane); the words move to registers for the
    // write-out and the bitmap is cleared for the next window
    uint4 wd[BIG_BPW];
    uint32_t ex[BIG_BPW];
#pragma unroll
    for (uint32_t i = 0; i < BIG_BPW; ++i) {
      const uint32_t b = wv + BIG_WAVES * i;
      wd[i] = make_uint4(0u, 0u, 0u, 0u);
      ex[i] = 0u;
      if (b < nblk) { // wave-uniform
        uint4 *src = reinterpret_cast<uint4 *>(&bits[b * 256u + lane * 4u]);
        wd[i] = *src;
        const uint32_t pc = __popc(wd[i].x) + __popc(wd[i].y) + __popc(wd[i].z) + __popc(wd[i].w);
        uint32_t incl = 0;
        if (__ballot(pc != 0u)) {
          incl = group_incl_scan<64>(pc);
          *src = make_uint4(0u, 0u, 0u, 0u);
        }
        ex[i] = incl - pc;
        if (lane == 63) s_blk[b] = incl;
      }
    }
    __syncthreads();
    // every wavefront scans the block totals for itself (nblk <= 128: two per lane)
    const uint32_t t0 = 2u * lane < nblk ? s_blk[2u * lane] : 0u;
    const uint32_t t1 = 2u * lane + 1u < nblk ? s_blk[2u * lane + 1u] : 0u;
    const uint32_t pair_incl = group_incl_scan<64>(t0 + t1);
    const uint32_t tot = group_last<64>(pair_incl);
    const uint32_t pair_excl = pair_incl - (t0 + t1); // offset of block 2 * lane
#pragma unroll
    for (uint32_t i = 0; i < BIG_BPW; ++i) {
      const uint32_t b = wv + BIG_WAVES * i;
      if (b >= nblk) break;
      // block b is held by lane b / 2 (b is wave-uniform)
      const uint32_t boff = (uint32_t)__builtin_amdgcn_readlane((int)pair_excl, (int)(b >> 1)) +
                            ((b & 1u) ? (uint32_t)__builtin_amdgcn_readlane((int)t0, (int)(b >> 1)) : 0u);
      uint32_t at = carry + boff + ex[i];
      const uint32_t cbase = (wb + b * 256u + lane * 4u) << 5;
      const uint32_t w4[4] = {wd[i].x, wd[i].y, wd[i].z, wd[i].w};
#pragma unroll
      for (uint32_t j = 0; j < 4; ++j) {
        uint32_t word = w4[j];
        while (word) {
          out[at++] = cbase + 32u * j + (uint32_t)__ffs(word) - 1u;
          word &= word - 1u;
        }
      }
    }
    carry += tot;
    __syncthreads(); // s_blk and the lists are reused by the next window
  }
  if (tid == 0) {
    m.nnz[row] = carry;
    m.num_key[row] = (uint8_t)num_key_of(m.sym_key[row], carry, m.hash_num_max);
  }
}

// ---- numeric -----------------------------------------------------------------------------------
struct NumOut {
  const uint64_t *__restrict__ c_ptr;
  uint32_t *__restrict__ c_col;
  double *__restrict__ c_val;
  unsigned long long *zero_count; // entries whose accumulated value is exactly 0.0
};

// numeric, window path: one wavefront per row.
// LDS per wavefront: bmp[words] of {lo: bitmap word, hi: number of set bits before the word}
//                    | vals[cap] doubles | FLAT only: owner[cap] uint32 + flat-walk scratch.
template <bool FLAT>
__global__ void __launch_bounds__(256)
    k_num_window(const uint32_t *__restrict__ rows, uint32_t nrows, Csr A, Csr B, RowMeta m, NumOut o,
                 uint32_t words, uint32_t cap, uint32_t wpb) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // one row per wavefront: scalar row state
  uint32_t idx = xcd_remap(blockIdx.x, gridDim.x) * wpb + w;
  if (w >= wpb || idx >= nrows) return;
  const size_t region = (size_t)words * 8u + (size_t)cap * 8u + (FLAT ? (size_t)cap * 4u + FLAT_BYTES : 0u);
  uint64_t *bmp = reinterpret_cast<uint64_t *>(smem + w * region);
  uint32_t *bmp32 = reinterpret_cast<uint32_t *>(bmp);
  double *vals = reinterpret_cast<double *>(bmp + words);
  uint32_t *owner = reinterpret_cast<uint32_t *>(vals + cap);
  const uint32_t row = rows[idx];
  const uint32_t base = m.cmin[row] & ~31u;
  const uint32_t nw = ((m.cmax[row] - base) >> 5) + 1u;
  const uint32_t n = m.nnz[row];
  const uint64_t off = o.c_ptr[row];
  const uint32_t *tin = m.temp_cols + m.t_off[m.lead[row]];
  for (uint32_t i = lane; i < nw; i += 64) bmp[i] = 0ull;
  for (uint32_t i = lane; i < n; i += 64) {
    vals[i] = 0.0;
    if constexpr (FLAT) owner[i] = 0u;
  }
  lds_fence();
  for (uint32_t p = lane; p < n; p += 64) {
    uint32_t c = tin[p];
    o.c_col[off + p] = c;
    uint32_t cc = c - base;
    atomicOr(&bmp32[(cc >> 5) * 2u], 1u << (cc & 31u));
  }
  lds_fence();
  uint32_t carry = 0;
  for (uint32_t r0 = 0; r0 < nw; r0 += 64) {
    uint32_t wi = r0 + lane;
    uint32_t pc = wi < nw ? __popc(bmp32[wi * 2u]) : 0u;
    uint32_t incl = group_incl_scan<64>(pc);
    if (wi < nw) bmp32[wi * 2u + 1u] = carry + incl - pc;
    carry += group_last<64>(incl);
  }
  lds_fence();
  auto position = [&](uint32_t c) {
    uint32_t cc = c - base;
    uint64_t e = bmp[cc >> 5];
    return (uint32_t)(e >> 32) + __popc((uint32_t)e & ((1u << (cc & 31u)) - 1u));
  };
  if constexpr (FLAT) {
    uint32_t stamp = 0;
    flat_walk<true>(A, B, row, lane, reinterpret_cast<unsigned char *>(owner + cap),
                    [&](uint32_t c, double bv, double av, uint32_t) {
                      const bool live = c != EMPTY_KEY;
                      const uint32_t pos = live ? position(c) : 0u;
                      claim_in_lane_order(vals, owner, stamp, lane, live && pos < n, pos, __dmul_rn(av, bv));
                    });
  } else {
    wave_row_walk<true>(A, B, row, lane, [&](uint32_t c, double bv, double av, uint32_t) {
      const uint32_t pos = position(c);
      if (pos < n) lds_add_f64(&vals[pos], __dmul_rn(av, bv));
    });
  }
  lds_fence();
  uint32_t nz = 0;
  for (uint32_t p = lane; p < n; p += 64) {
    double v = vals[p];
    o.c_val[off + p] = v;
    nz += (v == 0.0) ? 1u : 0u;
  }
  nz = group_sum<64>(nz);
  if (lane == 0 && nz) atomicAdd(o.zero_count, (unsigned long long)nz);
}

// numeric, window path, row groups: one wavefront per group of up to GRP_MAX consecutive rows of
// A that share their column list.  Every B entry is loaded once and its output position is found
// once; it is then multiplied by each row's own A value (staged in LDS) and added to that row's
// accumulator.  Each row still receives its products in the order of its A row, as on the CPU.
// LDS per wavefront: bmp[words] {bits, prefix} | vals[gmax][cap] | avals[gmax][alen], with
// (gmax, alen) = (GRP_MAX, GRP_ALEN) or, for wide groups, (GRP_MAX_WIDE, GRP_ALEN_WIDE).
template <bool WIDE>
__global__ void __launch_bounds__(256)
    k_num_window_grp(const uint32_t *__restrict__ rows, uint32_t nrows, Csr A, Csr B, RowMeta m, NumOut o,
                     uint32_t words, uint32_t cap, uint32_t wpb) {
  constexpr uint32_t gmax = WIDE ? GRP_MAX_WIDE : GRP_MAX, alen = WIDE ? GRP_ALEN_WIDE : GRP_ALEN;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  uint32_t idx = xcd_remap(blockIdx.x, gridDim.x) * wpb + w;
  if (w >= wpb || idx >= nrows) return;
  const size_t region = (size_t)words * 8u + (size_t)gmax * cap * 8u + (size_t)gmax * alen * 8u;
  uint64_t *bmp = reinterpret_cast<uint64_t *>(smem + w * region);
  uint32_t *bmp32 = reinterpret_cast<uint32_t *>(bmp);
  double *vals = reinterpret_cast<double *>(bmp + words);
  double *avals = vals + (size_t)gmax * cap;
  // row state through the scalar unit (all of it was written by earlier kernels)
  const uint32_t row = scalar_view(rows)[idx];
  const uint32_t cnt = min((uint32_t)scalar_view(m.gcnt)[row], gmax);
  const uint32_t base = scalar_view(m.cmin)[row] & ~31u;
  const uint32_t nw = ((scalar_view(m.cmax)[row] - base) >> 5) + 1u;
  const uint32_t n = scalar_view((const uint32_t *)m.nnz)[row];
  const uint32_t *tin = m.temp_cols + scalar_view(m.t_off)[row];
  // first loads of the B walk, overlapped with the LDS set-up below
  RowWalk<true, false> walk;
  walk.begin(A, B, row, lane);
  const uint32_t len_a = (uint32_t)(walk.a1 - walk.a0);
  // lane j holds the output offset and the A offset of the group's j-th row
  uint64_t my_off = 0, my_a = 0;
  if (lane < cnt) {
    my_off = o.c_ptr[row + lane];
    my_a = A.ptr[row + lane];
  }
  for (uint32_t i = lane; i < nw; i += 64) bmp[i] = 0ull;
#pragma unroll
  for (uint32_t j = 0; j < GRP_MAX; ++j) {
    if (j < cnt) {
      for (uint32_t i = lane; i < n; i += 64) vals[j * cap + i] = 0.0;
      const double *av = A.val + readlane64(my_a, j);
      for (uint32_t t = lane; t < len_a && t < alen; t += 64) avals[j * alen + t] = av[t];
    }
  }
  lds_fence();
  for (uint32_t p = lane; p < n; p += 64) {
    uint32_t c = tin[p];
#pragma unroll
    for (uint32_t j = 0; j < GRP_MAX; ++j)
      if (j < cnt) o.c_col[readlane64(my_off, j) + p] = c;
    uint32_t cc = c - base;
    atomicOr(&bmp32[(cc >> 5) * 2u], 1u << (cc & 31u));
  }
  lds_fence();
  uint32_t carry = 0;
  for (uint32_t r0 = 0; r0 < nw; r0 += 64) {
    uint32_t wi = r0 + lane;
    uint32_t pc = wi < nw ? __popc(bmp32[wi * 2u]) : 0u;
    uint32_t incl = group_incl_scan<64>(pc);
    if (wi < nw) bmp32[wi * 2u + 1u] = carry + incl - pc;
    carry += group_last<64>(incl);
  }
  lds_fence();
  walk.run(A, B, lane, [&](uint32_t c, double bv, double, uint32_t apos) {
    uint32_t cc = c - base;
    uint64_t e = bmp[cc >> 5];
    uint32_t pos = (uint32_t)(e >> 32) + __popc((uint32_t)e & ((1u << (cc & 31u)) - 1u));
    for (uint32_t j = 0; j < cnt; ++j) lds_add_f64(&vals[j * cap + pos], __dmul_rn(avals[j * alen + apos], bv));
  });
  lds_fence();
  uint32_t nz = 0;
#pragma unroll
  for (uint32_t j = 0; j < GRP_MAX; ++j) {
    if (j < cnt) {
      const uint64_t off = readlane64(my_off, j);
      for (uint32_t p = lane; p < n; p += 64) {
        double v = vals[j * cap + p];
        o.c_val[off + p] = v;
        nz += (v == 0.0) ? 1u : 0u;
      }
    }
  }
  nz = group_sum<64>(nz);
  if (lane == 0 && nz) atomicAdd(o.zero_count, (unsigned long long)nz);
}

// numeric, hash path: one wavefront per row.
// LDS per wavefront: vals[S/2] double | keys[S] uint32 | pos[S] uint16 (= 10*S bytes; nnz <= S/2)
//                    | FLAT only: owner[S/2] uint32 + flat-walk scratch.
// The sorted structural columns are inserted first (slot -> output position), so the
// accumulate loop only probes (no compare-and-swap) and the output needs no sort.
template <bool FLAT>
__global__ void __launch_bounds__(256)
    k_num_hash(const uint32_t *__restrict__ rows, uint32_t nrows, Csr A, Csr B, RowMeta m, NumOut o,
               uint32_t log2s, uint32_t wpb) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  uint32_t idx = xcd_remap(blockIdx.x, gridDim.x) * wpb + w;
  if (w >= wpb || idx >= nrows) return;
  const uint32_t S = 1u << log2s, mask = S - 1u;
  const size_t region = (size_t)S * 10u + (FLAT ? (size_t)S * 2u + FLAT_BYTES : 0u);
  unsigned char *reg = smem + w * region;
  double *vals = reinterpret_cast<double *>(reg);                 // S/2 doubles = 4*S bytes
  uint32_t *keys = reinterpret_cast<uint32_t *>(reg + 4u * S);    // S uint32
  uint16_t *pos = reinterpret_cast<uint16_t *>(reg + 8u * S);     // S uint16
  uint32_t *owner = reinterpret_cast<uint32_t *>(reg + 10u * S);  // S/2 uint32 (FLAT)
  const uint32_t row = rows[idx];
  const uint32_t n = m.nnz[row];
  const uint64_t off = o.c_ptr[row];
  const uint32_t *tin = m.temp_cols + m.t_off[m.lead[row]];
  for (uint32_t i = lane; i < S; i += 64) keys[i] = EMPTY_KEY;
  for (uint32_t i = lane; i < n; i += 64) {
    vals[i] = 0.0;
    if constexpr (FLAT) owner[i] = 0u;
  }
  lds_fence();
  for (uint32_t p = lane; p < n; p += 64) {
    uint32_t c = tin[p];
    o.c_col[off + p] = c;
    uint32_t h = hash_slot(c, log2s);
    uint32_t it = 0;
    while (atomicCAS(&keys[h], EMPTY_KEY, c) != EMPTY_KEY && it < S) { h = (h + 1u) & mask; ++it; }
    if (it < S) pos[h] = (uint16_t)p;
  }
  lds_fence();
  // slot of column c (S when absent, which cannot happen for a consistent symbolic result)
  auto find = [&](uint32_t c) {
    uint32_t h = hash_slot(c, log2s);
    uint32_t it = 0;
    while (keys[h] != c && it < S) { h = (h + 1u) & mask; ++it; }
    return it < S ? h : S;
  };
  if constexpr (FLAT) {
    uint32_t stamp = 0;
    flat_walk<true>(A, B, row, lane, reg + 12u * S, [&](uint32_t c, double bv, double av, uint32_t) {
      const uint32_t h = c != EMPTY_KEY ? find(c) : S;
      const bool live = h < S;
      claim_in_lane_order(vals, owner, stamp, lane, live, live ? (uint32_t)pos[h] : 0u, __dmul_rn(av, bv));
    });
  } else {
    wave_row_walk<true>(A, B, row, lane, [&](uint32_t c, double bv, double av, uint32_t) {
      const uint32_t h = find(c);
      if (h < S) lds_add_f64(&vals[pos[h]], __dmul_rn(av, bv));
    });
  }
  lds_fence();
  uint32_t nz = 0;
  for (uint32_t p = lane; p < n; p += 64) {
    double v = vals[p];
    o.c_val[off + p] = v;
    nz += (v == 0.0) ? 1u : 0u;
  }
  nz = group_sum<64>(nz);
  if (lane == 0 && nz) atomicAdd(o.zero_count, (unsigned long long)nz);
}

// ---- numeric, big rows: slices of the output row, one wavefront per slice ---------------------------
// Needs B rows with strictly increasing columns.  The sorted structural columns of a big row are
// cut into chunks of SLICE_W outputs; a work item is a run of consecutive chunks of one row and is
// carried out by ONE wavefront, chunk after chunk.  For a chunk the wavefront keeps the chunk's
// columns and accumulators in LDS and walks the A row in storage order, taking from every B row
// only the entries whose columns fall into the chunk (a contiguous piece, since the B row is
// sorted): every output still receives its additions in the order of the CPU loop, and different
// chunks of a row run on different wavefronts at the same time.
//
// An item first builds the list of A entries whose B rows reach into its column range at all
// ({position in B, end position, A value, next column}, in A order; global scratch, `list_cap`
// entries per workgroup).  A chunk then looks only at the list's `next column` to find the entries
// it has to touch, and advances their positions for the chunk that follows.
#ifndef SG_SLICE_UG
#define SG_SLICE_UG 4
#endif
constexpr uint32_t SLICE_W = 1024;
constexpr uint32_t SLICE_LDS = SLICE_W * 16u + FLAT_BYTES; // cols u32 | vals f64 | owner u32 | flat-walk scratch

struct SliceItem {
  uint32_t row, c0, nc; // chunks [c0, c0 + nc) of the row
};

// chunks per item: long A rows get longer runs, so that building the list pays off
__host__ __device__ inline uint32_t slice_run(uint32_t len_a) {
  uint32_t g = len_a / 64u;
  return g < 4u ? 4u : g > 64u ? 64u : g;
}

__global__ void __launch_bounds__(256)
    k_slice_count(const uint32_t *__restrict__ rows, uint32_t nrows, const uint64_t *__restrict__ a_ptr,
                  const uint32_t *__restrict__ nnz, uint32_t *__restrict__ count) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nrows) return;
  const uint32_t row = rows[i];
  const uint32_t chunks = (nnz[row] + SLICE_W - 1u) / SLICE_W;
  const uint32_t g = slice_run((uint32_t)(a_ptr[row + 1] - a_ptr[row]));
  count[i] = (chunks + g - 1u) / g;
}

__global__ void __launch_bounds__(256)
    k_slice_fill(const uint32_t *__restrict__ rows, uint32_t nrows, const uint64_t *__restrict__ a_ptr,
                 const uint32_t *__restrict__ nnz, const uint64_t *__restrict__ offset,
                 SliceItem *__restrict__ items) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nrows) return;
  const uint32_t row = rows[i];
  const uint32_t chunks = (nnz[row] + SLICE_W - 1u) / SLICE_W;
  const uint32_t g = slice_run((uint32_t)(a_ptr[row + 1] - a_ptr[row]));
  uint64_t o = offset[i];
  for (uint32_t c = 0; c < chunks; c += g) items[o++] = SliceItem{row, c, min(g, chunks - c)};
}

// longest A row among the rows of the numeric phase's big-row bin (atomic max into out[0], zero on entry)
__global__ void __launch_bounds__(256)
    k_slice_maxlen(uint32_t nrows, const uint64_t *__restrict__ a_ptr, const uint8_t *__restrict__ num_key,
                   uint32_t *__restrict__ out) {
  uint32_t mx = 0;
  for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < nrows; r += gridDim.x * blockDim.x)
    if (num_key[r] == (uint8_t)NUM_GLOBAL) mx = max(mx, (uint32_t)min((uint64_t)0xFFFFFFFFu, a_ptr[r + 1] - a_ptr[r]));
  mx = group_max<64>(mx);
  if ((threadIdx.x & 63u) == 0 && mx) atomicMax(out, mx);
}

// first position in [lo, hi) of B.col whose column is >= key (per lane)
__device__ __forceinline__ uint64_t lower_bound_col(const uint32_t *__restrict__ col, uint64_t lo, uint64_t hi,
                                                    uint64_t key) {
  while (lo < hi) {
    const uint64_t mid = lo + ((hi - lo) >> 1);
    if ((uint64_t)col[mid] < key) lo = mid + 1; else hi = mid;
  }
  return lo;
}
// the same for N independent ranges per lane, their loads in flight together
template <int N>
__device__ __forceinline__ void lower_bound_col_n(const uint32_t *__restrict__ col, uint64_t (&lo)[N], uint64_t (&hi)[N],
                                                  uint64_t key) {
  for (;;) {
    bool open = false;
#pragma unroll
    for (int u = 0; u < N; ++u) open |= lo[u] < hi[u];
    if (!__any(open)) break;
    uint32_t c[N];
    uint64_t mid[N];
#pragma unroll
    for (int u = 0; u < N; ++u) {
      mid[u] = lo[u] + ((hi[u] - lo[u]) >> 1);
      c[u] = lo[u] < hi[u] ? col[mid[u]] : 0u;
    }
#pragma unroll
    for (int u = 0; u < N; ++u)
      if (lo[u] < hi[u]) {
        if ((uint64_t)c[u] < key) lo[u] = mid[u] + 1; else hi[u] = mid[u];
      }
  }
}

// `work`: [0] = next item to hand out (zero on entry), total = n_items[0].  o.zero_count[3] is set
// when a list entry read back from scratch is not a position inside B (never seen; it would mean
// that the list was damaged) - the entry is skipped and the host reports the multiply as failed.
// list_scratch: per workgroup, list_cap entries of {cur u64 | end u64 | av f64 | next u32 | queue u32}.
constexpr uint32_t SLICE_ENTRY_BYTES = 32;
__global__ void __launch_bounds__(64)
    k_num_slice(const SliceItem *__restrict__ items, const uint64_t *__restrict__ n_items, Csr A, Csr B, RowMeta m,
                NumOut o, unsigned long long *__restrict__ work, unsigned char *__restrict__ list_scratch,
                uint64_t list_cap, uint64_t nnz_b) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint32_t *scol = reinterpret_cast<uint32_t *>(smem);
  double *vals = reinterpret_cast<double *>(smem + SLICE_W * 4u);
  uint32_t *owner = reinterpret_cast<uint32_t *>(smem + SLICE_W * 12u);
  unsigned char *flat = smem + SLICE_W * 16u;
  uint32_t *st = reinterpret_cast<uint32_t *>(flat);
  uint64_t *p0s = reinterpret_cast<uint64_t *>(flat + 256);
  double *avs = reinterpret_cast<double *>(flat + 768);
  const uint32_t lane = threadIdx.x;
  uint64_t *l_cur = reinterpret_cast<uint64_t *>(list_scratch + (size_t)blockIdx.x * list_cap * SLICE_ENTRY_BYTES);
  uint64_t *l_end = l_cur + list_cap;
  double *l_av = reinterpret_cast<double *>(l_end + list_cap);
  uint32_t *l_next = reinterpret_cast<uint32_t *>(l_av + list_cap);
  uint32_t *l_q = l_next + list_cap;
  const uint64_t total_items = n_items[0];
  for (uint32_t i = lane; i < SLICE_W; i += 64) owner[i] = 0u;
  uint32_t stamp = 0;
  uint32_t nz = 0;
  constexpr int U = 4;  // blocks of list entries whose loads and searches are in flight together
  constexpr int UG = SG_SLICE_UG; // B row pieces whose loads and column searches are in flight together
#ifdef SG_SLICE_STAMPS
  // diagnostic build: cycles per phase, summed over the wavefronts into o.zero_count[8..]
  unsigned long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tlast = __builtin_amdgcn_s_memtime();
#define SG_STAMP(i) { const unsigned long long tn = __builtin_amdgcn_s_memtime(); tph[i] += tn - tlast; tlast = tn; }
#else
#define SG_STAMP(i)
#endif
  for (;;) {
    unsigned long long t = 0;
    if (lane == 0) t = atomicAdd(work, 1ull);
    t = readlane64(t, 0);
    if (t >= total_items) break;
    SG_STAMP(0)
    const SliceItem it = items[t];
    const uint32_t row = it.row;
    const uint32_t n = m.nnz[row];
    const uint64_t off = o.c_ptr[row];
    const uint32_t *tin = m.temp_cols + m.t_off[m.lead[row]];
    const uint64_t a0 = A.ptr[row], a1 = A.ptr[row + 1];
    const uint32_t chunks = (n + SLICE_W - 1u) / SLICE_W;
    const bool item_first = it.c0 == 0, item_last = it.c0 + it.nc == chunks;
    const uint64_t clo_item = item_first ? 0ull : (uint64_t)tin[(size_t)it.c0 * SLICE_W];
    const uint64_t chi_item = item_last ? (1ull << 32) : (uint64_t)tin[(size_t)(it.c0 + it.nc) * SLICE_W];
    // -- the list: A entries whose B rows have entries in [clo_item, chi_item), in A order.  A row of
    //    up to 64 entries keeps it in registers (one entry per lane).
    const bool in_regs = a1 - a0 <= 64;
    uint64_t cur = 0, end = 0;
    double av = 0.0;
    uint32_t nextc = EMPTY_KEY;
    uint32_t nlist = 0;
    if (in_regs) {
      const uint64_t ka = a0 + lane;
      uint64_t lo = 0, b1 = 0;
      if (ka < a1) {
        const uint32_t k = A.col[ka];
        av = A.val[ka];
        const uint64_t b0 = B.ptr[k];
        b1 = B.ptr[k + 1];
        lo = item_first ? b0 : lower_bound_col(B.col, b0, b1, clo_item);
      }
      nextc = lo < b1 ? B.col[lo] : EMPTY_KEY;
      const bool ne = lo < b1 && (uint64_t)nextc < chi_item;
      cur = ne ? lo : 0;
      end = ne ? b1 : 0;
      if (!ne) nextc = EMPTY_KEY;
    } else {
      for (uint64_t ab = a0; ab < a1; ab += 64 * U) { // U blocks of 64 entries, their searches together
        uint64_t lo[U], hi[U], b1[U];
        double a[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const uint64_t ka = ab + 64u * u + lane;
          lo[u] = hi[u] = b1[u] = 0;
          a[u] = 0.0;
          if (ka < a1) {
            const uint32_t k = A.col[ka];
            a[u] = A.val[ka];
            lo[u] = B.ptr[k];
            b1[u] = B.ptr[k + 1];
            hi[u] = item_first ? lo[u] : b1[u]; // empty range: nothing to search
          }
        }
        lower_bound_col_n<U>(B.col, lo, hi, clo_item);
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const uint32_t nc0 = lo[u] < b1[u] ? B.col[lo[u]] : EMPTY_KEY;
          const bool ne = lo[u] < b1[u] && (uint64_t)nc0 < chi_item;
          const uint64_t mask = __ballot(ne);
          if (ne) {
            const uint32_t e = nlist + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
            l_cur[e] = lo[u]; l_end[e] = b1[u]; l_av[e] = a[u]; l_next[e] = nc0;
          }
          nlist += (uint32_t)__popcll(mask);
        }
      }
      // the list is read back by other lanes: all its stores have to be done
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    SG_STAMP(1)
    // -- the chunks
    for (uint32_t c = it.c0; c < it.c0 + it.nc; ++c) {
      const uint32_t pbase = c * SLICE_W;
      const uint32_t ncols = min(SLICE_W, n - pbase);
      const bool chunk_last = c + 1 == it.c0 + it.nc; // of the item
      const uint64_t chi = chunk_last ? chi_item : (uint64_t)tin[(size_t)pbase + SLICE_W];
      const uint32_t chunks_left = chunks - c;
      lds_fence();
      {
        uint32_t tc[SLICE_W / 64]; // the chunk's columns: all loads in flight before the first LDS store
#pragma unroll
        for (uint32_t i = 0; i < SLICE_W / 64; ++i) {
          const uint32_t p = i * 64u + lane;
          tc[i] = p < ncols ? tin[(size_t)pbase + p] : EMPTY_KEY;
        }
#pragma unroll
        for (uint32_t i = 0; i < SLICE_W / 64; ++i) {
          scol[i * 64u + lane] = tc[i];
          vals[i * 64u + lane] = 0.0;
        }
      }
      if (stamp > (1u << 25)) { // claims compare tags: start over before the stamp field overflows
        for (uint32_t i = lane; i < SLICE_W; i += 64) owner[i] = 0u;
        stamp = 0;
      }
      lds_fence();
      SG_STAMP(2)
      // position of a column that is in the chunk: four-way search, three probes per level in
      // flight together (five dependent LDS round trips instead of ten)
      auto find = [&](uint32_t col) {
        uint32_t pos = 0;
#pragma unroll
        for (uint32_t len = SLICE_W / 4; len > 0; len >>= 2) {
          const uint32_t a = scol[pos + len], b = scol[pos + 2u * len], c = scol[pos + 3u * len];
          pos += len * ((a <= col ? 1u : 0u) + (b <= col ? 1u : 0u) + (c <= col ? 1u : 0u));
        }
        return pos;
      };
      // One block of up to 64 list entries (one per lane; `act` = has entries in this chunk): adds
      // their pieces in A order, moves cur / nextc on.
      auto block = [&](bool act, uint64_t &cur, uint64_t end, double av, uint32_t &nextc) {
        const uint64_t amask = __ballot(act);
        const uint32_t nact = (uint32_t)__popcll(amask);
        // expected length of a piece: what is left of the B row, spread over the chunks that follow
        const uint32_t est = act ? (uint32_t)min(end - cur, (uint64_t)SLICE_W) / chunks_left : 0u;
        SG_STAMP(3)
        if (group_sum<64>(est) >= 16u * nact) {
          // long pieces: one B row at a time, its entries over the lanes (distinct columns: no
          // conflicts); U pieces are loaded and searched together, then added in A order
          // group = UG pieces; the loads of the next group fly while this one is searched and added
          struct Group {
            uint32_t l[UG], rem[UG], col[UG];
            uint64_t p0[UG], pend[UG];
            double a[UG], v[UG];
            bool on[UG];
          };
          uint64_t mask = amask;
          auto fetch = [&](Group &g) {
#pragma unroll
            for (int u = 0; u < UG; ++u) {
              g.on[u] = mask != 0;
              g.l[u] = g.on[u] ? (uint32_t)__builtin_ctzll(mask) : 0u;
              if (g.on[u]) mask &= mask - 1ull;
              g.p0[u] = readlane64(cur, g.l[u]);
              g.pend[u] = readlane64(end, g.l[u]);
              g.a[u] = readlane_f64(av, g.l[u]);
              g.rem[u] = g.on[u] ? (uint32_t)min(g.pend[u] - g.p0[u], (uint64_t)64) : 0u;
            }
#pragma unroll
            for (int u = 0; u < UG; ++u) {
              const bool have = lane < g.rem[u];
              g.col[u] = have ? B.col[g.p0[u] + lane] : EMPTY_KEY;
              g.v[u] = have ? B.val[g.p0[u] + lane] : 0.0;
            }
          };
          Group g, gn;
          fetch(g);
          while (g.on[0]) {
            gn.on[0] = false;
            if (mask) fetch(gn); // other lanes' pieces: not touched by what follows
            uint32_t pos[UG];
            bool in[UG];
#pragma unroll
            for (int u = 0; u < UG; ++u) {
              in[u] = lane < g.rem[u] && (uint64_t)g.col[u] < chi;
              pos[u] = find(in[u] ? g.col[u] : 0u);
            }
#pragma unroll
            for (int u = 0; u < UG; ++u) {
              if (!g.on[u]) continue; // wave-uniform
              if (in[u]) lds_add_f64(&vals[pos[u]], __dmul_rn(g.a[u], g.v[u]));
              uint32_t took = (uint32_t)__popcll(__ballot(in[u]));
              uint64_t ncur = g.p0[u] + took;
              uint64_t notin = __ballot(lane < g.rem[u] && !in[u]);
              uint32_t nxt = EMPTY_KEY;
              if (notin) {
                nxt = (uint32_t)__builtin_amdgcn_readlane((int)g.col[u], (int)__builtin_ctzll(notin));
              } else {
                while (ncur < g.pend[u]) { // the whole block belonged to the chunk and the B row goes on
                  const uint32_t r2 = (uint32_t)min(g.pend[u] - ncur, (uint64_t)64);
                  const bool have = lane < r2;
                  const uint32_t c2 = have ? B.col[ncur + lane] : EMPTY_KEY;
                  const double v2 = have ? B.val[ncur + lane] : 0.0;
                  const bool in2 = have && (uint64_t)c2 < chi;
                  if (in2) lds_add_f64(&vals[find(c2)], __dmul_rn(g.a[u], v2));
                  took = (uint32_t)__popcll(__ballot(in2));
                  ncur += took;
                  notin = __ballot(have && !in2);
                  if (notin) {
                    nxt = (uint32_t)__builtin_amdgcn_readlane((int)c2, (int)__builtin_ctzll(notin));
                    break;
                  }
                }
              }
              if (lane == g.l[u]) { cur = ncur; nextc = nxt; }
            }
            g = gn;
          }
          SG_STAMP(4)
        } else {
          // short pieces: products laid end to end over the lanes, conflicts resolved in lane order.
          // A piece's entries are distinct columns of this chunk: at most ncols, at most chi - nextc;
          // the three entries after the first are looked at directly, a longer piece is searched.
          uint32_t cnt = 0;
          uint64_t slo = 0, shi = 0, lim = 0;
          uint32_t c1 = EMPTY_KEY, c2 = EMPTY_KEY, c3 = EMPTY_KEY;
          if (act) {
            lim = min(end, cur + min((uint64_t)ncols, chi - nextc));
            c1 = cur + 1 < end ? B.col[cur + 1] : EMPTY_KEY;
            c2 = cur + 2 < end ? B.col[cur + 2] : EMPTY_KEY;
            c3 = cur + 3 < end ? B.col[cur + 3] : EMPTY_KEY;
            const bool i1 = cur + 1 < lim && (uint64_t)c1 < chi;
            const bool i2 = i1 && cur + 2 < lim && (uint64_t)c2 < chi;
            const bool i3 = i2 && cur + 3 < lim && (uint64_t)c3 < chi;
            cnt = 1u + (i1 ? 1u : 0u) + (i2 ? 1u : 0u) + (i3 ? 1u : 0u);
            if (i3 && cur + 4 < lim) { slo = cur + 4; shi = lim; }
          }
          const bool searched = slo < shi;
          const uint64_t found = lower_bound_col(B.col, slo, shi, chi);
          if (searched) cnt = (uint32_t)(found - cur);
          const uint32_t incl = group_incl_scan<64>(cnt);
          const uint32_t total = group_last<64>(incl);
          lds_fence();
          st[lane] = incl - cnt;
          p0s[lane] = cur;
          avs[lane] = av;
          lds_fence();
          auto fetchq = [&](uint32_t q, uint32_t &tt, uint32_t &col, double &bv) {
            const bool valid = q < total;
            const uint32_t qq = valid ? q : 0u;
            tt = 0;
#pragma unroll
            for (uint32_t sft = 32; sft > 0; sft >>= 1)
              if (st[tt + sft] <= qq) tt += sft;
            const uint64_t p = p0s[tt] + (qq - st[tt]);
            col = valid ? B.col[p] : EMPTY_KEY;
            bv = valid ? B.val[p] : 0.0;
          };
          uint32_t t_c = 0, col_c = EMPTY_KEY;
          double bv_c = 0.0;
          if (total) fetchq(lane, t_c, col_c, bv_c);
          for (uint32_t q0 = 0; q0 < total; q0 += 64) {
            uint32_t t_n = 0, col_n = EMPTY_KEY;
            double bv_n = 0.0;
            if (q0 + 64 < total) fetchq(q0 + 64 + lane, t_n, col_n, bv_n); // flies during this step
            const bool valid = col_c != EMPTY_KEY;
            const uint32_t pos = valid ? find(col_c) : 0u;
            claim_in_lane_order(vals, owner, stamp, lane, valid, pos, __dmul_rn(avs[t_c], bv_c));
            t_c = t_n; col_c = col_n; bv_c = bv_n;
          }
          cur += cnt;
          if (act) { // the next column is one of the three looked at, unless the piece was longer
            const uint32_t seen = cnt == 1u ? c1 : cnt == 2u ? c2 : c3;
            nextc = cur >= end ? EMPTY_KEY : cnt <= 3u ? seen : B.col[cur];
          }
          SG_STAMP(5)
        }
      };
      if (in_regs) {
        const bool act = cur < end && (uint64_t)nextc < chi;
        if (__any(act)) block(act, cur, end, av, nextc);
      } else {
        // The list entries that have something in this chunk are gathered, in list order, in a
        // queue (global scratch, written here and read back by other lanes after a fence), and
        // taken from it 64 at a time.
        uint32_t nq = 0;
        for (uint32_t lb = 0; lb < nlist; lb += 64 * U) {
          uint32_t nx[U];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const uint32_t e = lb + 64u * u + lane;
            nx[u] = e < nlist ? l_next[e] : EMPTY_KEY;
          }
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const bool a = nx[u] != EMPTY_KEY && (uint64_t)nx[u] < chi; // chi is 2^32 in a row's last chunk
            const uint64_t mask = __ballot(a);
            if (a) l_q[nq + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = lb + 64u * u + lane;
            nq += (uint32_t)__popcll(mask);
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        for (uint32_t qb = 0; qb < nq; qb += 64) {
          bool have = qb + lane < nq;
          uint32_t e = have ? l_q[qb + lane] : 0u;
          if (e >= nlist) { e = 0u; if (have) o.zero_count[3] = 1ull; have = false; }
          uint64_t ecur = have ? l_cur[e] : 0ull;
          const uint64_t eend = have ? l_end[e] : 0ull;
          const double eav = have ? l_av[e] : 0.0;
          uint32_t enext = have ? l_next[e] : EMPTY_KEY;
          if (have && !(ecur < eend && eend <= nnz_b && (uint64_t)enext < chi)) {
            o.zero_count[3] = 1ull; // an entry that cannot be: reported by the host, never followed
            have = false;
          }
          block(have, ecur, eend, eav, enext);
          if (have) { l_cur[e] = ecur; l_next[e] = enext; }
        }
        // the list is read back by other lanes in the next chunk: all its stores have to be done
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      }
      lds_fence();
      SG_STAMP(3)
      for (uint32_t p = lane; p < ncols; p += 64) {
        const double v = vals[p];
        o.c_col[off + pbase + p] = scol[p];
        o.c_val[off + pbase + p] = v;
        nz += (v == 0.0) ? 1u : 0u;
      }
      SG_STAMP(6)
    }
  }
#ifdef SG_SLICE_STAMPS
  if (lane == 0)
    for (int i = 0; i < 8; ++i) atomicAdd(&o.zero_count[8 + i], tph[i]);
#endif
  nz = group_sum<64>(nz);
  if (lane == 0 && nz) atomicAdd(o.zero_count, (unsigned long long)nz);
}
#undef SG_STAMP

// numeric, global path: ONE wavefront per row (the k order must be kept), accumulating straight
// into the row's slice of c_val through L2 atomics.  scratch: per workgroup `words_total`
// {bitmap word, prefix} pairs, all zero on entry and on exit.
__global__ void __launch_bounds__(64)
    k_num_global(const uint32_t *__restrict__ rows, uint32_t nrows, Csr A, Csr B, RowMeta m, NumOut o,
                 unsigned long long *__restrict__ scratch, uint64_t words_total) {
  unsigned long long *bmp = scratch + (size_t)blockIdx.x * words_total;
  uint32_t *bmp32 = reinterpret_cast<uint32_t *>(bmp);
  const uint32_t lane = threadIdx.x;
  for (uint32_t idx = blockIdx.x; idx < nrows; idx += gridDim.x) {
    const uint32_t row = rows[idx];
    const uint32_t n = m.nnz[row];
    const uint64_t off = o.c_ptr[row];
    const uint32_t *tin = m.temp_cols + m.t_off[m.lead[row]];
    const uint32_t w0 = m.cmin[row] >> 5, w1 = m.cmax[row] >> 5;
    for (uint32_t p = lane; p < n; p += 64) {
      uint32_t c = tin[p];
      o.c_col[off + p] = c;
      __hip_atomic_store(&o.c_val[off + p], 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      atomicOr(&bmp32[(size_t)(c >> 5) * 2u], 1u << (c & 31u));
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
    uint32_t carry = 0;
    for (uint32_t r0 = w0; r0 <= w1; r0 += 64) {
      uint32_t wi = r0 + lane;
      uint32_t bits = 0;
      if (wi <= w1) bits = __hip_atomic_load(&bmp32[(size_t)wi * 2u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      uint32_t pc = __popc(bits);
      uint32_t incl = group_incl_scan<64>(pc);
      if (wi <= w1 && bits)
        __hip_atomic_store(&bmp32[(size_t)wi * 2u + 1u], carry + incl - pc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      carry += group_last<64>(incl);
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
    wave_row_walk<true>(A, B, row, lane, [&](uint32_t c, double bv, double av, uint32_t) {
      unsigned long long e = __hip_atomic_load(&bmp[c >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      uint32_t pos = (uint32_t)(e >> 32) + __popc((uint32_t)e & ((1u << (c & 31u)) - 1u));
      if (pos < n) unsafeAtomicAdd(&o.c_val[off + pos], __dmul_rn(av, bv));
    });
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
    uint32_t nz = 0;
    for (uint32_t p = lane; p < n; p += 64) {
      double v = __hip_atomic_load(&o.c_val[off + p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      nz += (v == 0.0) ? 1u : 0u;
      uint32_t c = tin[p];
      __hip_atomic_store(&bmp[c >> 5], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    nz = group_sum<64>(nz);
    if (lane == 0 && nz) atomicAdd(o.zero_count, (unsigned long long)nz);
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
  }
}

// ---- dropping exact zeros (rare) ----------------------------------------------------------------
__global__ void __launch_bounds__(256)
    k_count_nonzero(uint32_t rows, const uint64_t *__restrict__ c_ptr, const double *__restrict__ c_val,
                    uint32_t *__restrict__ keep) {
  constexpr int G = 8;
  uint32_t row = (blockIdx.x * blockDim.x + threadIdx.x) / G;
  uint32_t lg = threadIdx.x & (G - 1);
  uint32_t cnt = 0;
  if (row < rows)
    for (uint64_t j = c_ptr[row] + lg; j < c_ptr[row + 1]; j += G) cnt += (c_val[j] != 0.0) ? 1u : 0u;
  cnt = group_sum<G>(cnt);
  if (row < rows && lg == 0) keep[row] = cnt;
}

// one wavefront per row, order-preserving
__global__ void __launch_bounds__(256)
    k_compact_rows(uint32_t rows, const uint64_t *__restrict__ old_ptr, const uint32_t *__restrict__ old_col,
                   const double *__restrict__ old_val, const uint64_t *__restrict__ new_ptr,
                   uint32_t *__restrict__ new_col, double *__restrict__ new_val) {
  uint32_t row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  uint32_t lane = threadIdx.x & 63;
  if (row >= rows) return;
  uint64_t j0 = old_ptr[row], j1 = old_ptr[row + 1];
  uint64_t out = new_ptr[row];
  for (uint64_t jb = j0; jb < j1; jb += 64) {
    uint64_t j = jb + lane;
    double v = j < j1 ? old_val[j] : 0.0;
    bool keepit = j < j1 && v != 0.0;
    unsigned long long mball = __ballot(keepit);
    if (keepit) {
      uint64_t p = out + __popcll(mball & ((1ull << lane) - 1ull));
      new_col[p] = old_col[j];
      new_val[p] = v;
    }
    out += __popcll(mball);
  }
}

// ---- section bitmap (reference src/bitmap.cpp:12-40) ---------------------------------------------
constexpr uint32_t MAX_SECTIONS = 4096;
__global__ void __launch_bounds__(256)
    k_section_bitmap(uint64_t nnz, const uint32_t *__restrict__ a_col, uint32_t section_width,
                     uint32_t n_sections, uint32_t *__restrict__ words /* ceil(n/32), pre-zeroed */) {
  __shared__ uint32_t s_bits[MAX_SECTIONS / 32];
  const uint32_t nwords = (n_sections + 31u) >> 5;
  for (uint32_t i = threadIdx.x; i < nwords; i += blockDim.x) s_bits[i] = 0u;
  __syncthreads();
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nnz; i += (uint64_t)gridDim.x * blockDim.x) {
    uint32_t s = min(a_col[i] / section_width, n_sections - 1u);
    uint32_t bit = 1u << (s & 31u);
    if (!(s_bits[s >> 5] & bit)) atomicOr(&s_bits[s >> 5], bit);
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < nwords; i += blockDim.x)
    if (s_bits[i]) atomicOr(&words[i], s_bits[i]);
}

// ---- per-launch accounting (profile mode): sums over the rows of one bin --------------------------
__global__ void __launch_bounds__(256)
    k_bin_sums(const uint32_t *__restrict__ rows, uint32_t nrows, const uint64_t *__restrict__ a_ptr,
               const uint64_t *__restrict__ r_nprod, const uint32_t *__restrict__ r_nnz,
               const uint8_t *__restrict__ gcnt, uint32_t whole_groups,
               unsigned long long *__restrict__ out /* nnz_a, nprod, nnz_c */) {
  uint64_t sa = 0, sp = 0, sc = 0;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nrows; i += gridDim.x * blockDim.x) {
    uint32_t r = rows[i];
    uint64_t mult = whole_groups ? max((uint32_t)gcnt[r], 1u) : 1u; // the launch handled the whole group
    sa += mult * (a_ptr[r + 1] - a_ptr[r]);
    sp += mult * r_nprod[r];
    sc += mult * r_nnz[r];
  }
  sa = group_sum64<64>(sa);
  sp = group_sum64<64>(sp);
  sc = group_sum64<64>(sc);
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(&out[0], (unsigned long long)sa);
    atomicAdd(&out[1], (unsigned long long)sp);
    atomicAdd(&out[2], (unsigned long long)sc);
  }
}

// ---- validation ------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
    k_validate(uint64_t rows, uint64_t cols, uint64_t nnz, const uint64_t *__restrict__ ptr,
               const uint32_t *__restrict__ col, uint32_t *__restrict__ bad) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t r = i; r < rows; r += stride)
    if (ptr[r] > ptr[r + 1] || ptr[r + 1] > nnz) atomicOr(bad, 1u);
  if (i == 0 && (ptr[0] != 0 || ptr[rows] != nnz)) atomicOr(bad, 2u);
  for (uint64_t j = i; j < nnz; j += stride)
    if ((uint64_t)col[j] >= cols) atomicOr(bad, 4u);
}

// ---- operand selection (keep_rows / column relabel) ------------------------------------------------
__global__ void __launch_bounds__(256)
    k_select_count(uint64_t n_out, const uint32_t *__restrict__ keep_rows, const uint64_t *__restrict__ ptr,
                   const uint32_t *__restrict__ col, const uint32_t *__restrict__ col_new,
                   uint32_t *__restrict__ out_len) {
  constexpr int G = 8;
  uint64_t r = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
  uint32_t lg = threadIdx.x & (G - 1);
  uint32_t cnt = 0;
  if (r < n_out) {
    uint64_t src = keep_rows ? keep_rows[r] : r;
    uint64_t j0 = ptr[src], j1 = ptr[src + 1];
    if (col_new) {
      for (uint64_t j = j0 + lg; j < j1; j += G) cnt += (col_new[col[j]] != 0xFFFFFFFFu) ? 1u : 0u;
    } else if (lg == 0) {
      cnt = (uint32_t)(j1 - j0);
    }
  }
  cnt = group_sum<G>(cnt);
  if (r < n_out && lg == 0) out_len[r] = cnt;
}

// one wavefront per output row, order-preserving copy with relabel
__global__ void __launch_bounds__(256)
    k_select_copy(uint64_t n_out, const uint32_t *__restrict__ keep_rows, const uint64_t *__restrict__ ptr,
                  const uint32_t *__restrict__ col, const double *__restrict__ val,
                  const uint32_t *__restrict__ col_new, const uint64_t *__restrict__ out_ptr,
                  uint32_t *__restrict__ out_col, double *__restrict__ out_val) {
  uint64_t r = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  uint32_t lane = threadIdx.x & 63;
  if (r >= n_out) return;
  uint64_t src = keep_rows ? keep_rows[r] : r;
  uint64_t j0 = ptr[src], j1 = ptr[src + 1];
  uint64_t out = out_ptr[r];
  for (uint64_t jb = j0; jb < j1; jb += 64) {
    uint64_t j = jb + lane;
    uint32_t c = 0xFFFFFFFFu;
    if (j < j1) {
      c = col[j];
      if (col_new) c = col_new[c];
    }
    bool keepit = j < j1 && c != 0xFFFFFFFFu;
    unsigned long long mball = __ballot(keepit);
    if (keepit) {
      uint64_t p = out + __popcll(mball & ((1ull << lane) - 1ull));
      out_col[p] = c;
      out_val[p] = val[j];
    }
    out += __popcll(mball);
  }
}

// ---- B exchange helpers ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
    k_row_len(uint64_t rows, const uint64_t *__restrict__ ptr, uint32_t *__restrict__ len) {
  uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r < rows) len[r] = (uint32_t)(ptr[r + 1] - ptr[r]);
}

} // namespace sg
