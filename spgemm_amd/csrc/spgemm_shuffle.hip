// Iterative locality shuffle on the device: row and column permutations that pull the entries of
// C towards the diagonal, so that a row block of A touches few row sections of B and the section
// bitmap can drop the rest.
//
// What it follows: reference src/partition.cpp:73-202 (iterative_shuffle and its two
// calculate_avg_indices_* helpers).  Every iteration gives each row (even iterations) or column
// (odd iterations) the mean position of the entries it holds in the other dimension and re-ranks
// the rows (columns) by that mean.  What differs, on purpose:
//   * the pattern of C comes from a device matrix (e.g. a product computed here), not from a
//     C_sparsity.mtx file (src/partition.cpp:129-131);
//   * the number of iterations is an argument; the reference runs until min(30 s, 0.00025 s * n)
//     of wall clock have passed (src/partition.cpp:142-146; its variance test never stops it),
//     which makes its result depend on the machine;
//   * means are exact: integer sum of positions, one division in double, and ties are broken by
//     the lower index (stable radix sort); the reference adds floats in storage order and sorts
//     with an unstable parallel sort, so its result is not reproducible either.  The CPU oracle
//     restates THIS rule and the GPU result equals it exactly.
#include <cstring>
#include <string>
#include <vector>

#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_segmented_radix_sort.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#include "internal.h"

namespace {

#define SH_HIP(expr)                                                                              \
  do {                                                                                            \
    hipError_t e_ = (expr);                                                                       \
    if (e_ != hipSuccess) {                                                                       \
      std::string m_ = std::string(#expr) + " failed: " + hipGetErrorString(e_);                  \
      return spgemm_internal_fail(SPGEMM_ERR_HIP, m_.c_str());                                    \
    }                                                                                             \
  } while (0)

struct DevBuf { // freed on every way out of the function
  void *p = nullptr;
  ~DevBuf() { if (p) (void)hipFree(p); }
  hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 8); }
  template <class T> T *as() const { return static_cast<T *>(p); }
};

__global__ void __launch_bounds__(256) k_iota(uint32_t n, uint32_t *out) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = i;
}

// entries per column
__global__ void __launch_bounds__(256) k_tr_count(uint64_t nnz, const uint32_t *__restrict__ col, uint32_t *__restrict__ cnt) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nnz) atomicAdd(&cnt[col[i]], 1u);
}

// rows of every column, in no particular order (only sums over them are taken)
__global__ void __launch_bounds__(256)
    k_tr_scatter(uint32_t rows, const uint64_t *__restrict__ ptr, const uint32_t *__restrict__ col,
                 unsigned long long *__restrict__ cursor, uint32_t *__restrict__ tr_row) {
  constexpr uint32_t G = 8;
  const uint64_t r64 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
  if (r64 >= rows) return;
  const uint32_t r = (uint32_t)r64, lg = threadIdx.x & (G - 1);
  for (uint64_t j = ptr[r] + lg; j < ptr[r + 1]; j += G) tr_row[atomicAdd(&cursor[col[j]], 1ull)] = r;
}

// key of segment s = bits of (double) mean of pos[idx[j]] over the segment; `empty` for an empty one
// (reference src/partition.cpp:83-90 and :108-115).  Non-negative doubles order like their bits.
__global__ void __launch_bounds__(256)
    k_seg_key(uint32_t n, const uint64_t *__restrict__ ptr, const uint32_t *__restrict__ idx,
              const uint32_t *__restrict__ pos, double empty, uint64_t *__restrict__ key) {
  constexpr uint32_t G = 8;
  const uint64_t s = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
  const uint32_t lg = threadIdx.x & (G - 1);
  uint64_t sum = 0, j0 = 0, j1 = 0;
  if (s < n) { j0 = ptr[s]; j1 = ptr[s + 1]; }
  for (uint64_t j = j0 + lg; j < j1; j += G) sum += pos[idx[j]];
#pragma unroll
  for (int d = G / 2; d > 0; d >>= 1) sum += __shfl_xor(sum, d, G);
  if (s < n && lg == 0) {
    const double mean = j1 > j0 ? (double)sum / (double)(j1 - j0) : empty;
    key[s] = (uint64_t)__double_as_longlong(mean);
  }
}

// the segment that was sorted to rank i gets position i (reference src/partition.cpp:157-160, :170-175)
__global__ void __launch_bounds__(256) k_rank(uint32_t n, const uint32_t *__restrict__ sorted_id, uint32_t *__restrict__ pos) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) pos[sorted_id[i]] = i;
}

// new column id of every entry (no entry may be dropped here: every id must be a column of the result)
__global__ void __launch_bounds__(256)
    k_relabel(uint64_t nnz, const uint32_t *__restrict__ col, const uint32_t *__restrict__ new_id, uint32_t *__restrict__ out) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nnz) out[i] = new_id[col[i]];
}

// row pointer relative to the first entry of a range of rows, as the 32-bit offsets the segmented sort takes
struct RelativeOffset {
  uint64_t base;
  __host__ __device__ unsigned int operator()(uint64_t v) const { return (unsigned int)(v - base); }
};

inline uint32_t blocks_for(uint64_t n, uint32_t per_block = 256) { return (uint32_t)((n + per_block - 1) / per_block); }

} // namespace

namespace {
struct Record12 {
  uint32_t a, b, c;
};
} // namespace

extern "C" int spgemm_internal_sort_records12(hipStream_t stream, const uint32_t *key_in, uint32_t *key_out,
                                              const void *rec_in, void *rec_out, size_t n, void *tmp, size_t *tmp_bytes) {
  hipError_t e = rocprim::radix_sort_pairs(tmp, *tmp_bytes, key_in, key_out, static_cast<const Record12 *>(rec_in),
                                           static_cast<Record12 *>(rec_out), n, 0, 32, stream);
  if (e != hipSuccess) return spgemm_internal_fail(SPGEMM_ERR_HIP, hipGetErrorString(e));
  return SPGEMM_OK;
}

extern "C" int spgemm_shuffle_iterative(spgemm_ctx *ctx, const spgemm_mat *C, uint32_t iterations,
                                        uint32_t *row_shuffle, uint32_t *col_shuffle, double *row_variance) {
  if (!ctx || !C || !row_shuffle || !col_shuffle) return spgemm_internal_fail(SPGEMM_ERR_INVALID, "NULL argument");
  hipStream_t st;
  int rc = spgemm_internal_stream(ctx, &st);
  if (rc != SPGEMM_OK) return rc;
  uint64_t rows64 = 0, cols64 = 0, nnz = 0;
  rc = spgemm_mat_info(C, &rows64, &cols64, &nnz);
  if (rc != SPGEMM_OK) return rc;
  if (rows64 > 0xFFFFFFFFull || cols64 > 0xFFFFFFFFull) return spgemm_internal_fail(SPGEMM_ERR_UNSUPPORTED, "more than 2^32 rows or columns");
  const uint32_t rows = (uint32_t)rows64, cols = (uint32_t)cols64;
  const uint64_t *c_ptr = nullptr;
  const uint32_t *c_col = nullptr;
  const double *c_val = nullptr;
  rc = spgemm_mat_device_ptrs(C, &c_ptr, &c_col, &c_val);
  if (rc != SPGEMM_OK) return rc;
  const uint32_t nmax = rows > cols ? rows : cols;

  DevBuf d_pos_r, d_pos_c, d_tcnt, d_tptr, d_trow, d_key, d_key2, d_id, d_id2, d_tmp;
  SH_HIP(d_pos_r.alloc((size_t)rows * 4));
  SH_HIP(d_pos_c.alloc((size_t)cols * 4));
  SH_HIP(d_tcnt.alloc(((size_t)cols + 1) * 4));
  SH_HIP(d_tptr.alloc(((size_t)cols + 1) * 8 * 2)); // pointer + scatter cursor
  SH_HIP(d_trow.alloc((size_t)nnz * 4));
  SH_HIP(d_key.alloc((size_t)nmax * 8));
  SH_HIP(d_key2.alloc((size_t)nmax * 8));
  SH_HIP(d_id.alloc((size_t)nmax * 4));
  SH_HIP(d_id2.alloc((size_t)nmax * 4));
  uint64_t *t_ptr = d_tptr.as<uint64_t>();
  unsigned long long *t_cur = d_tptr.as<unsigned long long>() + (size_t)cols + 1;

  // positions start as the identity (reference src/partition.cpp:126-127)
  if (rows) hipLaunchKernelGGL(k_iota, dim3(blocks_for(rows)), dim3(256), 0, st, rows, d_pos_r.as<uint32_t>());
  if (cols) hipLaunchKernelGGL(k_iota, dim3(blocks_for(cols)), dim3(256), 0, st, cols, d_pos_c.as<uint32_t>());
  if (nmax) hipLaunchKernelGGL(k_iota, dim3(blocks_for(nmax)), dim3(256), 0, st, nmax, d_id.as<uint32_t>());

  // pattern of C transposed (the reference loads the file a second time, transposed: :129-130)
  SH_HIP(hipMemsetAsync(d_tcnt.p, 0, ((size_t)cols + 1) * 4, st));
  if (nnz) hipLaunchKernelGGL(k_tr_count, dim3(blocks_for(nnz)), dim3(256), 0, st, nnz, c_col, d_tcnt.as<uint32_t>());
  {
    size_t bytes = 0;
    SH_HIP(rocprim::exclusive_scan(nullptr, bytes, d_tcnt.as<uint32_t>(), t_ptr, (uint64_t)0, (size_t)cols + 1,
                                   rocprim::plus<uint64_t>(), st));
    DevBuf scan_tmp;
    SH_HIP(scan_tmp.alloc(bytes));
    SH_HIP(rocprim::exclusive_scan(scan_tmp.p, bytes, d_tcnt.as<uint32_t>(), t_ptr, (uint64_t)0, (size_t)cols + 1,
                                   rocprim::plus<uint64_t>(), st));
    SH_HIP(hipMemcpyAsync(t_cur, t_ptr, ((size_t)cols + 1) * 8, hipMemcpyDeviceToDevice, st));
    if (rows && nnz)
      hipLaunchKernelGGL(k_tr_scatter, dim3(blocks_for((uint64_t)rows * 8)), dim3(256), 0, st, rows, c_ptr, c_col, t_cur,
                         d_trow.as<uint32_t>());
    SH_HIP(hipStreamSynchronize(st)); // scan_tmp goes out of scope
  }

  size_t sort_bytes = 0;
  SH_HIP(rocprim::radix_sort_pairs(nullptr, sort_bytes, d_key.as<uint64_t>(), d_key2.as<uint64_t>(), d_id.as<uint32_t>(),
                                   d_id2.as<uint32_t>(), (size_t)nmax, 0, 64, st));
  SH_HIP(d_tmp.alloc(sort_bytes));

  std::vector<uint64_t> h_key;
  for (uint32_t it = 0; it < iterations; ++it) {
    const bool by_col = it & 1u; // odd iterations move the columns (reference :149-163)
    const uint32_t n = by_col ? cols : rows;
    if (!n) continue;
    if (by_col)
      hipLaunchKernelGGL(k_seg_key, dim3(blocks_for((uint64_t)n * 8)), dim3(256), 0, st, n, t_ptr, d_trow.as<uint32_t>(),
                         d_pos_r.as<uint32_t>(), (double)cols, d_key.as<uint64_t>());
    else
      hipLaunchKernelGGL(k_seg_key, dim3(blocks_for((uint64_t)n * 8)), dim3(256), 0, st, n, c_ptr, c_col,
                         d_pos_c.as<uint32_t>(), (double)cols, d_key.as<uint64_t>());
    SH_HIP(rocprim::radix_sort_pairs(d_tmp.p, sort_bytes, d_key.as<uint64_t>(), d_key2.as<uint64_t>(), d_id.as<uint32_t>(),
                                     d_id2.as<uint32_t>(), (size_t)n, 0, 64, st));
    hipLaunchKernelGGL(k_rank, dim3(blocks_for(n)), dim3(256), 0, st, n, d_id2.as<uint32_t>(),
                       (by_col ? d_pos_c : d_pos_r).as<uint32_t>());
    SH_HIP(hipGetLastError());
    if (!by_col && row_variance) {
      // variance of (mean - rank) over the rows, as the reference logs it (:169-178)
      h_key.resize(n);
      SH_HIP(hipMemcpyAsync(h_key.data(), d_key2.p, (size_t)n * 8, hipMemcpyDeviceToHost, st));
      SH_HIP(hipStreamSynchronize(st));
      double sx = 0, sx2 = 0;
      for (uint32_t i = 0; i < n; ++i) {
        double mean;
        memcpy(&mean, &h_key[i], 8);
        const double x = mean - (double)i;
        sx += x / n;
        sx2 += x * x / n;
      }
      row_variance[it / 2] = n > 1 ? (sx2 - sx * sx) * ((double)n / (double)(n - 1)) : 0.0;
    }
  }
  // the maps handed out take a position to the row (column) that sits there (reference :187-199)
  std::vector<uint32_t> pr(rows), pc(cols);
  if (rows) SH_HIP(hipMemcpyAsync(pr.data(), d_pos_r.p, (size_t)rows * 4, hipMemcpyDeviceToHost, st));
  if (cols) SH_HIP(hipMemcpyAsync(pc.data(), d_pos_c.p, (size_t)cols * 4, hipMemcpyDeviceToHost, st));
  SH_HIP(hipStreamSynchronize(st));
  for (uint32_t r = 0; r < rows; ++r) row_shuffle[pr[r]] = r;
  for (uint32_t c = 0; c < cols; ++c) col_shuffle[pc[c]] = c;
  return SPGEMM_OK;
}

// Every row of a CSR structure sorted by 32-bit keys (one per entry, below `key_limit`), stable, the
// values moving with their keys: key_in / val_in -> key_out / val_out (different arrays).  Queued
// on the context's stream; returns after the stream has drained (the sort's scratch goes back to
// the pool).
extern "C" int spgemm_internal_sort_rows(spgemm_ctx *ctx, uint64_t rows, uint64_t nnz, uint64_t key_limit,
                                         const uint64_t *m_ptr, const uint32_t *key_in, uint32_t *key_out,
                                         const double *val_in, double *val_out) {
  hipStream_t st;
  int rc = spgemm_internal_stream(ctx, &st);
  if (rc != SPGEMM_OK) return rc;
  if (!nnz || !rows) return SPGEMM_OK;
  void *tmp = nullptr;
  size_t tmp_got = 0;
  auto fail_out = [&](int code) {
    (void)hipStreamSynchronize(st);
    if (tmp) spgemm_internal_pool_release(ctx, tmp, tmp_got);
    return code;
  };
#define RL_HIP(expr)                                                                              \
  do {                                                                                            \
    hipError_t e_ = (expr);                                                                       \
    if (e_ != hipSuccess) {                                                                       \
      std::string m_ = std::string(#expr) + " failed: " + hipGetErrorString(e_);                  \
      return fail_out(spgemm_internal_fail(SPGEMM_ERR_HIP, m_.c_str()));                          \
    }                                                                                             \
  } while (0)
  {
    unsigned int end_bit = 1;
    while (end_bit < 32 && (key_limit - 1) >> end_bit) ++end_bit;
    // the segmented sort counts entries in 32 bits: ranges of rows of fewer than 2^31 entries each
    std::vector<uint64_t> cuts{0, rows};
    if (nnz >= (1ull << 31)) {
      std::vector<uint64_t> h_ptr(rows + 1);
      RL_HIP(hipMemcpyAsync(h_ptr.data(), m_ptr, (rows + 1) * 8, hipMemcpyDeviceToHost, st));
      RL_HIP(hipStreamSynchronize(st));
      cuts.assign(1, 0);
      uint64_t r0 = 0;
      while (r0 < rows) {
        uint64_t r1 = r0 + 1; // a row of 2^31 entries or more cannot exist: fewer than 2^32 columns... but guard anyway
        while (r1 < rows && h_ptr[r1 + 1] - h_ptr[r0] < (1ull << 31)) ++r1;
        if (h_ptr[r1] - h_ptr[r0] >= (1ull << 31)) return fail_out(spgemm_internal_fail(SPGEMM_ERR_UNSUPPORTED, "a row of 2^31 entries or more"));
        cuts.push_back(r1);
        r0 = r1;
      }
      // the bases of the ranges are needed below
      std::vector<uint64_t> bases;
      for (size_t i = 0; i + 1 < cuts.size(); ++i) bases.push_back(h_ptr[cuts[i]]);
      bases.push_back(nnz);
      for (size_t i = 0; i + 1 < cuts.size(); ++i) {
        const uint64_t base = bases[i], cnt = bases[i + 1] - base, segs = cuts[i + 1] - cuts[i];
        if (!cnt) continue;
        auto offs = rocprim::make_transform_iterator(m_ptr + cuts[i], RelativeOffset{base});
        size_t need = 0;
        RL_HIP(rocprim::segmented_radix_sort_pairs(nullptr, need, key_in + base, key_out + base, val_in + base,
                                                   val_out + base, (unsigned int)cnt, (unsigned int)segs, offs, offs + 1, 0, end_bit, st));
        if (need > tmp_got) {
          RL_HIP(hipStreamSynchronize(st));
          if (tmp) spgemm_internal_pool_release(ctx, tmp, tmp_got);
          tmp = nullptr;
          rc = spgemm_internal_pool_alloc(ctx, need, &tmp, &tmp_got);
          if (rc != SPGEMM_OK) return fail_out(rc);
        }
        RL_HIP(rocprim::segmented_radix_sort_pairs(tmp, need, key_in + base, key_out + base, val_in + base,
                                                   val_out + base, (unsigned int)cnt, (unsigned int)segs, offs, offs + 1, 0, end_bit, st));
      }
    } else {
      auto offs = rocprim::make_transform_iterator(m_ptr, RelativeOffset{0});
      size_t need = 0;
      RL_HIP(rocprim::segmented_radix_sort_pairs(nullptr, need, key_in, key_out, val_in, val_out, (unsigned int)nnz,
                                                 (unsigned int)rows, offs, offs + 1, 0, end_bit, st));
      rc = spgemm_internal_pool_alloc(ctx, need ? need : 8, &tmp, &tmp_got);
      if (rc != SPGEMM_OK) return fail_out(rc);
      RL_HIP(rocprim::segmented_radix_sort_pairs(tmp, need, key_in, key_out, val_in, val_out, (unsigned int)nnz,
                                                 (unsigned int)rows, offs, offs + 1, 0, end_bit, st));
    }
  }
  RL_HIP(hipStreamSynchronize(st));
#undef RL_HIP
  if (tmp) spgemm_internal_pool_release(ctx, tmp, tmp_got);
  return SPGEMM_OK;
}

// Columns of M renamed by new_id (a device array of M.cols entries, a bijection onto [0, new_cols))
// and every row sorted by its new column ids, values moving with their columns.  A permutation
// of the columns commutes with the multiply: (A * B) with the columns of the RESULT renamed equals
// A * (B with its columns renamed), entry for entry and bit for bit, because every output entry
// still receives the same products in the same order.  Renaming the result leaves B, and with it
// the locality of the multiply, untouched.
extern "C" int spgemm_mat_relabel_columns(spgemm_ctx *ctx, const spgemm_mat *M, const uint32_t *d_new_id,
                                          uint64_t new_cols, spgemm_mat **out) {
  if (!ctx || !M || !d_new_id || !out) return spgemm_internal_fail(SPGEMM_ERR_INVALID, "NULL argument");
  *out = nullptr;
  hipStream_t st;
  int rc = spgemm_internal_stream(ctx, &st);
  if (rc != SPGEMM_OK) return rc;
  uint64_t rows = 0, cols = 0, nnz = 0;
  rc = spgemm_mat_info(M, &rows, &cols, &nnz);
  if (rc != SPGEMM_OK) return rc;
  if (new_cols == 0 || new_cols > 0xFFFFFFFFull) return spgemm_internal_fail(SPGEMM_ERR_INVALID, "new_cols out of range");
  const uint64_t *m_ptr = nullptr;
  const uint32_t *m_col = nullptr;
  const double *m_val = nullptr;
  rc = spgemm_mat_device_ptrs(M, &m_ptr, &m_col, &m_val);
  if (rc != SPGEMM_OK) return rc;
  spgemm_mat *R = nullptr;
  rc = spgemm_internal_alloc_mat(ctx, rows, new_cols, nnz, &R);
  if (rc != SPGEMM_OK) return rc;
  const uint64_t *cr_ptr = nullptr;
  const uint32_t *cr_col = nullptr;
  const double *cr_val = nullptr;
  rc = spgemm_mat_device_ptrs(R, &cr_ptr, &cr_col, &cr_val);
  if (rc != SPGEMM_OK) { spgemm_mat_free(ctx, R); return rc; }
  uint64_t *r_ptr = const_cast<uint64_t *>(cr_ptr);
  uint32_t *r_col = const_cast<uint32_t *>(cr_col);
  double *r_val = const_cast<double *>(cr_val);
  void *keys = nullptr;
  size_t keys_got = 0;
  auto fail_out = [&](int code) {
    (void)hipStreamSynchronize(st);
    if (keys) spgemm_internal_pool_release(ctx, keys, keys_got);
    spgemm_mat_free(ctx, R);
    return code;
  };
#define RL_HIP(expr)                                                                              \
  do {                                                                                            \
    hipError_t e_ = (expr);                                                                       \
    if (e_ != hipSuccess) {                                                                       \
      std::string m_ = std::string(#expr) + " failed: " + hipGetErrorString(e_);                  \
      return fail_out(spgemm_internal_fail(SPGEMM_ERR_HIP, m_.c_str()));                          \
    }                                                                                             \
  } while (0)
  RL_HIP(hipMemcpyAsync(r_ptr, m_ptr, (rows + 1) * 8, hipMemcpyDeviceToDevice, st));
  if (nnz) {
    rc = spgemm_internal_pool_alloc(ctx, nnz * 4, &keys, &keys_got);
    if (rc != SPGEMM_OK) return fail_out(rc);
    hipLaunchKernelGGL(k_relabel, dim3(blocks_for(nnz)), dim3(256), 0, st, nnz, m_col, d_new_id, static_cast<uint32_t *>(keys));
    RL_HIP(hipGetLastError());
    // every row sorted by its new ids, values moving with their columns
    rc = spgemm_internal_sort_rows(ctx, rows, nnz, new_cols, m_ptr, static_cast<const uint32_t *>(keys), r_col, m_val, r_val);
    if (rc != SPGEMM_OK) return fail_out(rc);
  }
  RL_HIP(hipStreamSynchronize(st));
#undef RL_HIP
  if (keys) spgemm_internal_pool_release(ctx, keys, keys_got);
  *out = R;
  return SPGEMM_OK;
}
