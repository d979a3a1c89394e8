// Partitions and shuffles of the host driver (interface of the reference's include/partition.hpp
// and include/parts.hpp; the integer work is done by the C ABI).
#pragma once

#include "../../include/spgemm_amd.h"
#include "matrix.hpp"

#include <cstring>
#include <fstream>
#include <numeric>
#include <stdexcept>
#include <string>
#include <vector>

namespace partition {

using Partition = spgemm_partition; // {start_row, end_row, start_col, end_col}, reference include/partition.hpp:10-15
typedef std::vector<Partition> Partitions;
typedef std::vector<midx_t> Shuffle;

// Random permutation (reference src/partition.cpp:15-37), seeded so that runs can be repeated.
inline Shuffle shuffle(size_t size, uint64_t seed) {
  Shuffle s(size);
  if (spgemm_shuffle_random((uint32_t)size, seed, s.data()) != SPGEMM_OK) throw std::runtime_error(spgemm_last_error());
  return s;
}

// Iterative locality shuffle (reference src/partition.cpp:123-202).  The reference reads the pattern
// of C from C_sparsity.mtx; here C = A * B is computed on device `device` first.  `iterations`
// replaces the reference's wall-clock limit.
inline void iterative_shuffle(const matrix::CSRMatrix<double> &A, const matrix::CSRMatrix<double> &B, int device,
                              uint32_t iterations, Shuffle *shuffled_rows, Shuffle *shuffled_cols) {
  auto check = [](int rc) { if (rc != SPGEMM_OK) throw std::runtime_error(spgemm_last_error()); };
  auto upload = [&](spgemm_ctx *ctx, const matrix::CSRMatrix<double> &m, spgemm_mat **out) {
    std::vector<uint64_t> ptr(m.height + 1);
    for (midx_t i = 0; i <= m.height; ++i) ptr[i] = m.row_ptr[i];
    std::vector<double> val(m.non_zeros); // the blob's values may be 4-byte aligned only
    std::memcpy(val.data(), m.values, (size_t)m.non_zeros * sizeof(double));
    check(spgemm_mat_from_host(ctx, m.height, m.width, m.non_zeros, ptr.data(), m.col_idx, val.data(), out));
  };
  spgemm_ctx *ctx = nullptr;
  check(spgemm_create(device, &ctx));
  spgemm_mat *dA = nullptr, *dB = nullptr, *dC = nullptr;
  try {
    upload(ctx, A, &dA);
    upload(ctx, B, &dB);
    check(spgemm_multiply(ctx, dA, dB, &dC));
    shuffled_rows->assign(A.height, 0);
    shuffled_cols->assign(B.width, 0);
    check(spgemm_shuffle_iterative(ctx, dC, iterations, shuffled_rows->data(), shuffled_cols->data(), nullptr));
  } catch (...) {
    if (dC) spgemm_mat_free(ctx, dC);
    if (dB) spgemm_mat_free(ctx, dB);
    if (dA) spgemm_mat_free(ctx, dA);
    spgemm_destroy(ctx);
    throw;
  }
  spgemm_mat_free(ctx, dC);
  spgemm_mat_free(ctx, dB);
  spgemm_mat_free(ctx, dA);
  spgemm_destroy(ctx);
}

inline Shuffle identity(size_t size) {
  Shuffle s(size);
  std::iota(s.begin(), s.end(), 0u);
  return s;
}

// reference src/partition.cpp:204-217
inline void save_partitions(const Partitions &p, const std::string &file) {
  std::ofstream csv(file);
  if (csv.fail()) throw std::runtime_error("cannot write " + file);
  csv << "rank,start_row,end_row,start_col,end_col" << std::endl;
  for (size_t i = 0; i < p.size(); ++i)
    csv << i << "," << p[i].start_row << "," << p[i].end_row << "," << p[i].start_col << "," << p[i].end_col << std::endl;
}

// one integer per line (reference src/partition.cpp:219-252)
inline void save_shuffle(const Shuffle &s, const std::string &file) {
  std::ofstream out(file);
  if (out.fail()) throw std::runtime_error("cannot write " + file);
  for (auto i : s) out << i << "\n";
}

inline bool load_shuffle(const std::string &file, Shuffle &s) {
  std::ifstream in(file);
  if (!in.is_open()) return false;
  size_t idx = 0;
  unsigned long long v;
  while (in >> v) {
    if (idx >= s.size()) return false;
    s[idx++] = (midx_t)v;
  }
  return idx == s.size();
}

} // namespace partition

namespace parts {
namespace baseline {

// Equal rows and columns, remainder spread over the first parts (reference src/parts/baseline.cpp:10-42).
inline partition::Partitions partition(const matrix::Fields &f, int n_parts) {
  partition::Partitions p(n_parts);
  if (spgemm_partition_naive(f.height, f.width, n_parts, p.data()) != SPGEMM_OK) throw std::runtime_error(spgemm_last_error());
  return p;
}

// Rows cut where the running weight exceeds total / n_parts (reference src/parts/baseline.cpp:44-86
// weighs rows by nnz of C_sparsity; any weight works, the driver passes per-row product counts).
inline partition::Partitions balanced_partition(const std::vector<uint64_t> &row_weight, midx_t width, int n_parts) {
  partition::Partitions p(n_parts);
  if (spgemm_partition_balanced(row_weight.data(), (uint32_t)row_weight.size(), width, n_parts, p.data()) != SPGEMM_OK)
    throw std::runtime_error(spgemm_last_error());
  return p;
}

} // namespace baseline
} // namespace parts
