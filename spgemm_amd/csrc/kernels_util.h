// Section bitmap, per-launch accounting, validation, operand selection, B exchange helpers.
// Part of the gfx950 SpGEMM kernels; included through kernels.h.
#pragma once
#include "kernels_common.h"

namespace sg {

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

// Output invariants: out[0] += rows whose columns are not strictly increasing, out[1] += stored
// values that are exactly 0.0.  16 lanes per row.
__global__ void __launch_bounds__(256)
    k_check_rows(uint64_t rows, const uint64_t *__restrict__ ptr, const uint32_t *__restrict__ col,
                 const double *__restrict__ val, unsigned long long *__restrict__ out) {
  constexpr int G = 16;
  const uint64_t row = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
  const uint32_t lg = threadIdx.x & (G - 1);
  uint32_t bad = 0;
  uint64_t zeros = 0;
  if (row < rows) {
    const uint64_t j0 = ptr[row], j1 = ptr[row + 1];
    for (uint64_t j = j0 + lg; j < j1; j += G) {
      if (j > j0 && col[j - 1] >= col[j]) bad = 1;
      zeros += (val[j] == 0.0) ? 1u : 0u;
    }
  }
  bad = group_max<G>(bad);
  zeros = group_sum64<G>(zeros);
  if (row < rows && lg == 0) {
    if (bad) atomicAdd(&out[0], 1ull);
    if (zeros) atomicAdd(&out[1], (unsigned long long)zeros);
  }
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

// ---- small results for the host: up to six runs of 32-bit words copied into the context's pinned host
// buffer by ONE launch (a copy call per counter costs about 10 us each on the stream) -----------------
struct ReportSeg {
  const uint32_t *src;
  uint32_t *dst;
  uint32_t words;
};
struct ReportSegs {
  ReportSeg s[6];
};
__global__ void __launch_bounds__(256) k_report(ReportSegs r) {
#pragma unroll
  for (int i = 0; i < 6; ++i)
    for (uint32_t w = threadIdx.x; w < r.s[i].words; w += 256) r.s[i].dst[w] = r.s[i].src[w];
}

// ---- probe: order of the lanes of one LDS fp64 add that share an address -----------------------------
// The flat walks add up to 64 products with ONE ds_add_f64 instruction and need the lanes that hit
// the same accumulator to be applied in ascending lane order (the CPU's order of the products).
// Bit parity of those kernels with the CPU loop rests on this probe: the hardware's order is not
// documented.  Each trial lets the 64 lanes of every wavefront of the workgroup add doubles of
// very different magnitude to a few accumulators (1 .. 64 of them, some lanes idle) and compares
// every accumulator bit for bit with the sum taken lane by lane.  The patterns are those of the
// kernels that rely on the order: all wavefronts of the workgroup add at the same time into ONE
// table of PROBE_SLOTS doubles (just under a CU's 160 KiB), their accumulators interleaved (wave w
// owns the entries 4 * k + w), at strides of 1 .. 64 entries between a wave's accumulators, powers
// of two included: every accumulator of a trial then falls into the same few banks.  out[0] counts
// the differences.
constexpr uint32_t PROBE_SLOTS = 19456, PROBE_LDS_BYTES = PROBE_SLOTS * 8u, PROBE_BLOCKS = 256, PROBE_TRIALS = 64;
static_assert(63u * 64u * 4u + 3u < PROBE_SLOTS, "the largest stride keeps 64 accumulators of 4 wavefronts inside the table");
__global__ void __launch_bounds__(256) k_probe_lane_order(uint32_t trials, unsigned long long *__restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double *acc = reinterpret_cast<double *>(smem);
  __shared__ double xs[4][64];
  __shared__ uint32_t ps[4][64];
  const uint32_t w = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  auto mix = [](uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
  };
  unsigned long long bad = 0;
  for (uint32_t t = 0; t < trials; ++t) {
    const uint32_t cfg = blockIdx.x + t;
    const uint32_t nslots = 1u + (cfg * 7u) % 64u;
    // strides 1, 2, 4, ... 64 (every accumulator in the same banks) and the odd ones in between
    const uint32_t stride = (cfg & 1u) ? 1u << ((cfg >> 1) % 7u) : 1u + 2u * ((cfg >> 1) % 32u);
    const uint64_t h = mix(((uint64_t)blockIdx.x << 40) ^ ((uint64_t)w << 34) ^ ((uint64_t)t << 8) ^ lane);
    const uint32_t slot = (uint32_t)(h % nslots);
    double x = ldexp(1.0 + (double)((h >> 32) & 0xFFFFFu) / 1048576.0, (int)((h >> 20) % 60u) - 30);
    if (h & (1ull << 60)) x = -x;
    const bool active = (cfg & 2u) == 0 || (mix(h) % 3u) != 0;
    const uint32_t span = 64u * stride * 4u; // entries the four wavefronts' accumulators of this trial lie in
    for (uint32_t i = threadIdx.x; i < span; i += 256) acc[i] = 0.0;
    xs[w][lane] = x;
    ps[w][lane] = active ? slot : 0xFFFFFFFFu;
    __syncthreads();
    if (active) lds_add_f64(&acc[slot * stride * 4u + w], x);
    __syncthreads();
    if (lane < nslots) {
      double up = 0.0;
      for (int l = 0; l < 64; ++l)
        if (ps[w][l] == lane) up += xs[w][l];
      if (__double_as_longlong(acc[lane * stride * 4u + w]) != __double_as_longlong(up)) ++bad;
    }
    __syncthreads();
  }
  if (bad) atomicAdd(out, bad);
}

// ---- B as 16-byte records {column, 0, value} (read by the big-row numeric kernel) ---------------------
__global__ void __launch_bounds__(256)
    k_pack_records(uint64_t nnz, const uint32_t *__restrict__ col, const double *__restrict__ val, Rec12 *__restrict__ out) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nnz; i += (uint64_t)gridDim.x * blockDim.x) {
    const unsigned long long v = (unsigned long long)__double_as_longlong(val[i]);
    out[i] = Rec12{col[i], (uint32_t)v, (uint32_t)(v >> 32)};
  }
}

// ---- streaming copy: what this GPU's HBM delivers to a plain copy -----------------------------------------
// Every lane has four 16-byte loads in flight before its first store; a workgroup moves tiles of
// 16 KiB (four rows of 256 lanes x 16 bytes).
constexpr uint32_t COPY_UNROLL = 4;
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256)
    k_stream_copy(const uint4 *__restrict__ src4, uint4 *__restrict__ dst4, uint64_t n16) {
  const u32x4_t *__restrict__ src = reinterpret_cast<const u32x4_t *>(src4);
  u32x4_t *__restrict__ dst = reinterpret_cast<u32x4_t *>(dst4);
  const uint64_t tile = 256ull * COPY_UNROLL;
  const uint64_t ntiles = n16 / tile;
  for (uint64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const uint64_t i = t * tile + threadIdx.x;
    u32x4_t v[COPY_UNROLL];
#pragma unroll
    for (uint32_t u = 0; u < COPY_UNROLL; ++u) v[u] = __builtin_nontemporal_load(&src[i + 256ull * u]);
#pragma unroll
    for (uint32_t u = 0; u < COPY_UNROLL; ++u) __builtin_nontemporal_store(v[u], &dst[i + 256ull * u]);
  }
  for (uint64_t i = ntiles * tile + (uint64_t)blockIdx.x * 256ull + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * 256ull)
    dst[i] = src[i];
}

// ---- B exchange helpers ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
    k_row_len(uint64_t rows, const uint64_t *__restrict__ ptr, uint32_t *__restrict__ len) {
  uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r < rows) len[r] = (uint32_t)(ptr[r + 1] - ptr[r]);
}

} // namespace sg
