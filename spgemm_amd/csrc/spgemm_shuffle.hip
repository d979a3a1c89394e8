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
  const uint32_t r = (blockIdx.x * blockDim.x + threadIdx.x) / G, lg = threadIdx.x & (G - 1);
  if (r >= rows) return;
  for (uint64_t j = ptr[r] + lg; j < ptr[r + 1]; j += G) tr_row[atomicAdd(&cursor[col[j]], 1ull)] = r;
}

// key of segment s = bits of (double) mean of pos[idx[j]] over the segment; `empty` for an empty one
// (reference src/partition.cpp:83-90 and :108-115).  Non-negative doubles order like their bits.
__global__ void __launch_bounds__(256)
    k_seg_key(uint32_t n, const uint64_t *__restrict__ ptr, const uint32_t *__restrict__ idx,
              const uint32_t *__restrict__ pos, double empty, uint64_t *__restrict__ key) {
  constexpr uint32_t G = 8;
  const uint32_t s = (blockIdx.x * blockDim.x + threadIdx.x) / G, lg = threadIdx.x & (G - 1);
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

inline uint32_t blocks_for(uint64_t n, uint32_t per_block = 256) { return (uint32_t)((n + per_block - 1) / per_block); }

} // namespace

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
