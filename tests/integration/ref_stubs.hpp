// Stand-ins for the parts of the reference's headers that the device-resident adapter of the
// repository's integration guide touches, with the reference's names, members and blob layouts
// (include/matrix.hpp:9-31, 63-68; include/matrix_impl.hpp:228-363; src/matrix.cpp:66-68;
// include/partition.hpp:10-17; include/mults.hpp:17-33).  They exist so that the documented adapter
// is compiled, and run, by the tests; nothing here is part of the product.
#pragma once
#include <bitset>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <iostream>
#include <memory>
#include <stdexcept>
#include <string>
#include <tuple>
#include <utility>
#include <vector>

#ifndef N_SECTIONS
#define N_SECTIONS 8 // the reference sets it at configure time (CMakeLists.txt:6)
#endif
#define ROUND8(x) (((x) + 7) & ~7)
typedef uint32_t midx_t;

namespace matrix {
typedef struct alignas(8) Fields {
  bool transposed;
  midx_t height;
  midx_t width;
  midx_t non_zeros;
} Fields;
typedef struct alignas(8) BlockedFields {
  size_t height;
  size_t n_sections;
  size_t section_offst[N_SECTIONS];
  size_t section_start_row[N_SECTIONS];
} BlockedFields;
using Data = std::pair<std::byte *, size_t>;
namespace utils {
inline BlockedFields *get_blocked_fields(std::byte *serialized_data) { return (BlockedFields *)((void *)serialized_data); }
} // namespace utils

// view over a CSR blob: Fields | row_ptr[height + 1] | col_idx[nnz] | values[nnz], padded to 8 bytes
template <typename T = double> class CSRMatrix {
public:
  midx_t height, width, non_zeros;
  midx_t *row_ptr = nullptr, *col_idx = nullptr;
  T *values = nullptr;
  static size_t expected_data_size(size_t h, size_t nnz) {
    return ROUND8(sizeof(Fields) + (h + 1) * sizeof(midx_t) + nnz * sizeof(midx_t) + nnz * sizeof(T));
  }
  CSRMatrix(const Data &d) {
    const Fields *f = (const Fields *)d.first;
    height = f->height; width = f->width; non_zeros = f->non_zeros;
    if (d.second != expected_data_size(height, non_zeros)) throw std::runtime_error("CSR blob has the wrong size");
    row_ptr = (midx_t *)(d.first + sizeof(Fields));
    col_idx = row_ptr + height + 1;
    values = (T *)((std::byte *)col_idx + (size_t)non_zeros * sizeof(midx_t));
  }
};

// What BlockedCSRMatrix::filter (include/matrix_impl.hpp:507-534) returns for a block given as plain
// CSR arrays: header + the blobs of the row sections whose bit is set, in ascending order.
inline std::vector<std::byte> filtered_blob(midx_t height, midx_t width, const std::vector<midx_t> &row_ptr,
                                            const std::vector<midx_t> &col_idx, const std::vector<double> &values,
                                            const std::bitset<N_SECTIONS> &bitmap) {
  std::vector<std::byte> out(sizeof(BlockedFields));
  BlockedFields bf{};
  bf.height = height;
  bf.n_sections = bitmap.count();
  const size_t sh = height / N_SECTIONS;
  size_t j = 0;
  for (size_t s = 0; s < N_SECTIONS; ++s) {
    if (!bitmap[s]) continue;
    const size_t r0 = sh * s, r1 = s + 1 == N_SECTIONS ? height : sh * (s + 1);
    const size_t z0 = row_ptr[r0], z1 = row_ptr[r1];
    const size_t sz = CSRMatrix<>::expected_data_size(r1 - r0, z1 - z0), at = out.size();
    out.resize(at + sz, std::byte(0));
    Fields f{false, (midx_t)(r1 - r0), width, (midx_t)(z1 - z0)};
    std::memcpy(out.data() + at, &f, sizeof(f));
    midx_t *rp = (midx_t *)(out.data() + at + sizeof(Fields));
    for (size_t r = r0; r <= r1; ++r) rp[r - r0] = (midx_t)(row_ptr[r] - z0);
    std::memcpy(rp + (r1 - r0 + 1), col_idx.data() + z0, (z1 - z0) * sizeof(midx_t));
    std::memcpy((std::byte *)(rp + (r1 - r0 + 1)) + (z1 - z0) * sizeof(midx_t), values.data() + z0, (z1 - z0) * sizeof(double));
    bf.section_offst[j] = at;
    bf.section_start_row[j] = r0;
    ++j;
  }
  std::memcpy(out.data(), &bf, sizeof(bf));
  return out;
}
} // namespace matrix

namespace partition {
typedef struct Partition {
  midx_t start_row, end_row, start_col, end_col;
} Partition;
typedef std::vector<Partition> Partitions;
} // namespace partition

namespace mults {
class MatrixMultiplication {
protected:
  int rank;
  int n_nodes;
  partition::Partitions partitions;

public:
  virtual void gemm(std::vector<size_t>, size_t) {}
  virtual size_t get_B_serialization_size() { return 0; }
  virtual void save_result(std::string) {}
  virtual void reset() {}
  MatrixMultiplication(int rank, int n_nodes, partition::Partitions partitions)
      : rank(rank), n_nodes(n_nodes), partitions(std::move(partitions)) {}
  virtual ~MatrixMultiplication() = default;
};
} // namespace mults
