// Bin keys, lane helpers (DPP scans, broadcasts), XCD-aware block mapping.
// Part of the gfx950 SpGEMM kernels; included through kernels.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sg {

constexpr uint32_t EMPTY_KEY = 0xFFFFFFFFu;
constexpr uint32_t KEY_NONE = 255u; // row needs no kernel (no products)

// ---- bin keys ------------------------------------------------------------------------------
// symbolic:  [0,24)  window, key = span_tier*4 + gsel, span <= 2048 << span_tier
//            [24,56) hash,   key = 24 + tier*4 + gsel, nprod <= 64 << tier (table of 128 << tier slots);
//                    the last tier (up to 8192 products, a table of 128 KiB) only for rows whose B rows
//                    are long (gsel 3): a whole workgroup shares the table and walks them
//            [56,60) global, key = 56 + gsel
// numeric:   [0,96)   window, key = (span_tier*4 + cap_tier)*4 + gsel, nnz <= 128 << (2*cap_tier)
//            [96,128) hash,   key = 96 + tier*4 + gsel,                 nnz <= 64 << tier (<= 4096)
//            128      global
//            [129,153) window, row groups: key = 129 + span_tier*4 + cap_tier (one wavefront per group
//                      of up to GRP_MAX consecutive rows of A with identical column lists)
constexpr int HASH_TIERS = 8; // table sizes 128 << tier
constexpr uint32_t HASH_WAVE_MAX_NPROD = 4096; // beyond it only the workgroup kernel (long B rows) takes a row
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
// the smaller of the two wide tiers: A rows of up to GRP_ALEN_MID entries and up to GRP_CAP_MID outputs
// (3 x 27 = 81 entries and 3 x 125 = 375 outputs: three unknowns per node of a 3-D grid)
constexpr uint32_t GRP_ALEN_MID = 96, GRP_CAP_MID = 384;
constexpr uint32_t GRP_CAP_TOP = 512; // most outputs a grouped row can have (the largest `cap` of the group kernel)
__host__ __device__ inline uint32_t group_max_rows(uint64_t len_a) { return len_a > GRP_ALEN ? GRP_MAX_WIDE : GRP_MAX; }

struct ClassParams {
  uint32_t win_max_span;   // window path allowed up to this column span (<= 65536)
  uint32_t win_ratio;      // ... and only if span <= win_ratio * nprod
  uint32_t hash_max_nprod; // LDS hash symbolic up to this many products (<= 8192)
  uint32_t g_thresh0, g_thresh1, g_thresh2; // mean B-row length thresholds for 8/16/32 lanes
  uint32_t short_max_nprod; // rows of the hash path with at most this many products take the short-row kernels (0: none)
};
// an entry of B as a 12-byte record {column, value}: as many bytes as the two arrays hold, and a short
// piece of a row lies in one 128-byte line (or two) instead of one of columns and one of values
struct __attribute__((packed, aligned(4))) Rec12 {
  uint32_t col, val_lo, val_hi;
};

// Short rows (kernels_short.h): at most SH_MAX_NPROD products and SH_MAX_ALEN entries in the A row.
// In the hash bins sub-bin 0 (key & 3 == 0) is theirs alone; the other rows use sub-bins 1..3.
constexpr uint32_t SH_MAX_NPROD = 512, SH_MAX_ALEN = 64;

// smallest l with 2^l >= x, at most 31 (every row of the analysis computes two of these and the class
// below: a counting loop and a 64-bit division here were a third of that kernel's vector instructions)
__host__ __device__ inline uint32_t ceil_log2_u32(uint32_t x) {
  if (x <= 1u) return 0u;
  const uint32_t l = 32u - (uint32_t)__builtin_clz(x - 1u);
  return l < 31u ? l : 31u;
}

// class of the mean length of the row's B rows, ceil(nprod / len_a), against the three thresholds
// (ceil(n / d) <= t is n <= t * d: no division)
__host__ __device__ inline uint32_t gsel_of(uint64_t nprod, uint32_t len_a, const ClassParams &p) {
  return nprod <= (uint64_t)p.g_thresh0 * len_a ? 0u : nprod <= (uint64_t)p.g_thresh1 * len_a ? 1u
         : nprod <= (uint64_t)p.g_thresh2 * len_a ? 2u : 3u;
}

__host__ __device__ inline uint32_t sym_key_of(uint64_t nprod, uint32_t span, uint32_t len_a,
                                               const ClassParams &p) {
  if (nprod == 0) return KEY_NONE;
  uint32_t g = gsel_of(nprod, len_a, p);
  if (span <= p.win_max_span && (uint64_t)span <= (uint64_t)p.win_ratio * nprod) {
    uint32_t t = ceil_log2_u32(span < 2048u ? 2048u : span) - 11u;
    return SYM_WIN0 + t * 4 + g;
  }
  if (nprod <= p.hash_max_nprod && (nprod <= HASH_WAVE_MAX_NPROD || g == 3u)) {
    uint32_t t = ceil_log2_u32(nprod < 64u ? 64u : (uint32_t)nprod) - 6u; // nprod <= 64 << t
    if (nprod <= p.short_max_nprod && nprod <= SH_MAX_NPROD && len_a <= SH_MAX_ALEN) return SYM_HASH0 + t * 4;
    return SYM_HASH0 + t * 4 + (g ? g : 1u);
  }
  return SYM_GLOBAL0 + g;
}

// group bins: key = NUM_GRP0 + span_tier * 4 + q; q = 0: up to 128 outputs, q = 1: up to 512 (both:
// up to GRP_MAX rows of up to GRP_ALEN entries), q = 2: wide group of up to GRP_CAP_MID outputs and
// GRP_ALEN_MID entries per A row, q = 3: wide group of up to 512 outputs, GRP_ALEN_WIDE entries
// hash_num_max: most outputs of a row that is accumulated by the LDS hash kernel (<= 4096); rows
// beyond it go to the big-row bin
__host__ __device__ inline uint32_t num_key_of(uint32_t sym_key, uint32_t nnz, uint32_t hash_num_max,
                                               uint32_t group_rows = 1, uint32_t len_a = 0) {
  if (nnz == 0) return KEY_NONE;
  uint32_t g = sym_key & 3u;
  if (sym_key >= (uint32_t)SYM_HASH0 && sym_key < (uint32_t)SYM_GLOBAL0 && g == 0u) // short row: its own numeric
    return NUM_HASH0 + (sym_key - (uint32_t)SYM_HASH0);                               // kernel, tiers by products
  if (sym_key < (uint32_t)SYM_HASH0) {
    uint32_t st = sym_key >> 2;
    if (nnz <= 8192u) {
      uint32_t ct = nnz <= 128 ? 0u : nnz <= 512 ? 1u : nnz <= 2048 ? 2u : 3u;
      if (group_rows > 1 && ct <= 1) {
        if (len_a <= GRP_ALEN) return NUM_GRP0 + st * 4 + ct;
        return NUM_GRP0 + st * 4 + ((len_a <= GRP_ALEN_MID && nnz <= GRP_CAP_MID) ? 2u : 3u);
      }
      return NUM_WIN0 + (st * 4 + ct) * 4 + g;
    }
    return NUM_GLOBAL;
  }
  if (nnz <= hash_num_max) {
    uint32_t ct = ceil_log2_u32(nnz < 64u ? 64u : nnz) - 6u; // nnz <= 64 << ct
    return NUM_HASH0 + ct * 4 + (g ? g : 1u);             // sub-bin 0 belongs to the short rows
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

} // namespace sg
