// Multiply operators of the host driver: the reference's mults interface
// (include/mults.hpp:17-33: gemm / get_B_serialization_size / save_result / reset) implemented on
// the GPU through the C ABI of include/spgemm_amd.h.
//
// One operator per rank.  The ranks of this driver live in ONE process (rank r drives GPU
// r % n_gpus); communication::World stands in for MPI_COMM_WORLD and lets an operator reach the
// B row sections of its peers, which it pulls device to device (spgemm_pack_sections +
// spgemm_buffer_copy, peer-to-peer over xGMI between different GPUs).
#pragma once

#include "../../include/spgemm_amd.h"
#include "matrix.hpp"
#include "measure.hpp"
#include "partition.hpp"

#include <barrier>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#define HOST_TRY(expr)                                                                          \
  do {                                                                                          \
    if ((expr) != SPGEMM_OK) throw std::runtime_error(std::string(#expr) + ": " + spgemm_last_error()); \
  } while (0)

namespace bitmap {
// Section bitmap of an A block, computed on the device (reference src/bitmap.cpp:12-40).
inline std::vector<bool> compute_bitmap(spgemm_ctx *ctx, const spgemm_mat *part_A, uint32_t n_sections) {
  std::vector<uint64_t> words((n_sections + 63) / 64, 0);
  HOST_TRY(spgemm_section_bitmap(ctx, part_A, n_sections, words.data()));
  std::vector<bool> bits(n_sections);
  size_t set = 0;
  for (uint32_t s = 0; s < n_sections; ++s) {
    bits[s] = (words[s >> 6] >> (s & 63)) & 1ull;
    set += bits[s];
  }
  printf("Bitmap drop percentage: %g\n", (n_sections - set) / (double)n_sections); // parsed by scripts/drop_average.py:14
  return bits;
}
inline std::vector<uint64_t> to_words(const std::vector<bool> &bits) {
  std::vector<uint64_t> w((bits.size() + 63) / 64, 0);
  for (size_t s = 0; s < bits.size(); ++s)
    if (bits[s]) w[s >> 6] |= 1ull << (s & 63);
  return w;
}
} // namespace bitmap

namespace mults {
class HipSpgemm;
}

namespace communication {
// Stand-in for MPI_COMM_WORLD inside one process: the ranks are host threads (one per rank, as
// the reference's ranks are processes, src/main.cpp:35-45), barrier() is MPI_Barrier.
struct World {
  int n_nodes = 0;
  std::vector<spgemm_ctx *> ctx;          // one per rank (ranks may share a GPU)
  std::vector<mults::HipSpgemm *> ops;    // registered by the operators
  std::unique_ptr<std::barrier<>> bar;
  explicit World(int n, int n_gpus) : n_nodes(n), ctx(n, nullptr), ops(n, nullptr), bar(new std::barrier<>(n)) {
    for (int r = 0; r < n; ++r) HOST_TRY(spgemm_create(r % n_gpus, &ctx[r]));
  }
  void barrier() { bar->arrive_and_wait(); }
  ~World() {
    for (auto c : ctx) spgemm_destroy(c);
  }
};
} // namespace communication

namespace mults {

class MatrixMultiplication { // reference include/mults.hpp:17-33
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

struct HipOptions {
  bool drop = true;          // exchange only the B sections the bitmap marks
  uint32_t n_sections = 512; // the reference's N_SECTIONS (compile-time there, CMakeLists.txt:6)
  // where a column shuffle is applied: to the result of every multiply (B keeps its column
  // locality) or to B before the exchange; see product_rows_are_narrow
  bool relabel_result = true;
};

// True when most rows of A * B span at most max_span columns (column hull of the B rows a row of A
// selects): B then has column locality that relabelling it would destroy, and a column shuffle is
// better applied to the result.  For a B without it (random graphs) sorting the result costs more
// than it saves.  Reads the whole operands, so every rank comes to the same answer.
inline bool product_rows_are_narrow(const matrix::Triplets<double> &tA, const matrix::Triplets<double> &tB,
                                    uint64_t max_span = 65536) {
  const int64_t none = -1;
  std::vector<int64_t> bmin(tB.nrows, INT64_MAX), bmax(tB.nrows, none);
  for (size_t i = 0; i < tB.rows.size(); ++i) {
    bmin[tB.rows[i]] = std::min<int64_t>(bmin[tB.rows[i]], tB.cols[i]);
    bmax[tB.rows[i]] = std::max<int64_t>(bmax[tB.rows[i]], tB.cols[i]);
  }
  std::vector<int64_t> lo(tA.nrows, INT64_MAX), hi(tA.nrows, none);
  for (size_t i = 0; i < tA.rows.size(); ++i) {
    const midx_t k = tA.cols[i];
    if (bmax[k] == none) continue;
    lo[tA.rows[i]] = std::min(lo[tA.rows[i]], bmin[k]);
    hi[tA.rows[i]] = std::max(hi[tA.rows[i]], bmax[k]);
  }
  uint64_t live = 0, narrow = 0;
  for (midx_t r = 0; r < tA.nrows; ++r)
    if (hi[r] != none) { ++live; narrow += (uint64_t)(hi[r] - lo[r] + 1) <= max_span; }
  return live > 0 && 2 * narrow >= live;
}

// Row-wise Gustavson on the GPU for one rank.
//
//  * part_A: rows keep_rows of A (the reference loads them the same way, src/main.cpp:213-217);
//  * B: this rank owns the row sections [rank*N/P, (rank+1)*N/P) of B with ALL columns; the column
//    order col_order (the whole B_shuffle: column p of the result stands for original column
//    col_order[p]; the reference passes only its slice, keep_cols, because it cuts B by columns) is
//    applied to the result;
//  * gemm(): pulls the sections the bitmap marks from every rank, in ascending order, straight into
//    the col_idx/val arrays of a full-height local B, then one multiply gives the rank's rows of C
//    (local row ids, shuffled-global column ids: exactly what the reference's C_<rank>.mtx holds).
class HipSpgemm : public MatrixMultiplication {
  communication::World &world;
  spgemm_ctx *ctx;
  HipOptions opt;
  spgemm_mat *part_A = nullptr;
  spgemm_mat *block_B = nullptr; // my row sections
  spgemm_mat *C = nullptr;
  void *d_col_new_id = nullptr; // device: new column id of every original column (null: identity)
  uint64_t b_first_row = 0, b_rows_total = 0, b_cols = 0;
  std::vector<uint64_t> sec_nnz;       // entries of each of MY sections (index: global section id)
  std::vector<uint64_t> sec_blob_size; // reference blob size of each of my sections
  midx_t a_height = 0;
  // exchange buffers, allocated once by prepare() (the reference allocates its receive buffers inside
  // every gemm, src/mults/drop_parallel.cpp:65-68): what I receive (= the arrays of my local B) and,
  // per destination, what I send
  struct Triple { void *len = nullptr, *col = nullptr, *val = nullptr; uint64_t rows = 0, nnz = 0; };
  Triple recv;
  void *recv_ptr = nullptr;
  std::vector<Triple> send;            // index: destination rank (mine stays empty)
  std::vector<uint64_t> rows_from, nnz_from;
  bool prepared = false;

  static void upload(spgemm_ctx *ctx, const matrix::CSRMatrix<> &m, spgemm_mat **out) {
    std::vector<uint64_t> ptr(m.height + 1);
    for (midx_t i = 0; i <= m.height; ++i) ptr[i] = m.row_ptr[i];
    std::vector<double> val(m.non_zeros); // the blob's values may be 4-byte aligned only
    std::memcpy(val.data(), m.values, (size_t)m.non_zeros * sizeof(double));
    HOST_TRY(spgemm_mat_from_host(ctx, m.height, m.width, m.non_zeros, ptr.data(), m.col_idx, val.data(), out));
  }

  uint32_t sec_lo(int r) const { return (uint32_t)((uint64_t)r * opt.n_sections / n_nodes); }
  void section_rows(uint32_t s, uint64_t *r0, uint64_t *r1) const {
    uint64_t sh = b_rows_total / opt.n_sections;
    *r0 = sh * s;
    *r1 = (s + 1 == opt.n_sections) ? b_rows_total : sh * (s + 1);
  }

public:
  std::vector<bool> bitmap;               // reference include/mults.hpp:129 (std::bitset<N_SECTIONS>)
  std::vector<std::vector<bool>> bitmaps; // all ranks', filled by the driver (src/main.cpp:253-258)

  HipSpgemm(int rank, int n_nodes, partition::Partitions partitions, const matrix::Triplets<double> &tA,
            const std::vector<midx_t> *keep_rows, const matrix::Triplets<double> &tB,
            const std::vector<midx_t> *col_order, communication::World &world, HipOptions opt)
      : MatrixMultiplication(rank, n_nodes, std::move(partitions)), world(world), ctx(world.ctx.at(rank)), opt(opt) {
    world.ops.at(rank) = this;
    matrix::ManagedCSRMatrix<> hA(tA, false, keep_rows, nullptr);
    a_height = hA.height;
    upload(ctx, hA, &part_A);
    b_rows_total = tB.nrows;
    b_cols = tB.ncols;
    if (b_rows_total / opt.n_sections == 0) throw std::runtime_error("fewer B rows than sections");
    uint64_t r0, r1, tmp;
    section_rows(sec_lo(rank), &r0, &tmp);
    if (sec_lo(rank + 1) > sec_lo(rank)) section_rows(sec_lo(rank + 1) - 1, &tmp, &r1); else r1 = r0;
    b_first_row = r0;
    std::vector<midx_t> rows_keep;
    for (uint64_t r = r0; r < r1; ++r) rows_keep.push_back((midx_t)r);
    // The column order is applied to the RESULT of every multiply (spgemm_mat_relabel_columns), not
    // to B: a permutation of the columns commutes with the product bit for bit, and B keeps the
    // column locality the kernels live on.  new id of original column c = position of c in col_order
    // (the keep_cols semantics of the loader).
    matrix::ManagedCSRMatrix<> hB(tB, false, &rows_keep, opt.relabel_result ? nullptr : col_order);
    if (!opt.relabel_result) sort_rows(hB); // the loader relabels in place: back to ascending columns
    upload(ctx, hB, &block_B);
    if (col_order && opt.relabel_result) {
      bool identity = col_order->size() == b_cols;
      for (size_t p = 0; identity && p < col_order->size(); ++p) identity = (*col_order)[p] == p;
      if (!identity) {
        if (col_order->size() != b_cols) throw std::runtime_error("the column order must hold every column of B once");
        std::vector<uint32_t> new_id(b_cols);
        for (size_t p = 0; p < col_order->size(); ++p) new_id[(*col_order)[p]] = (uint32_t)p;
        HOST_TRY(spgemm_buffer_alloc(ctx, (size_t)b_cols * 4, &d_col_new_id));
        HOST_TRY(spgemm_buffer_upload(ctx, d_col_new_id, new_id.data(), (size_t)b_cols * 4));
      }
    }
    sec_nnz.assign(opt.n_sections, 0);
    sec_blob_size.assign(opt.n_sections, 0);
    for (uint32_t s = sec_lo(rank); s < sec_lo(rank + 1); ++s) {
      uint64_t a, b;
      section_rows(s, &a, &b);
      sec_nnz[s] = hB.row_ptr[b - r0] - hB.row_ptr[a - r0];
      sec_blob_size[s] = matrix::CSRMatrix<>::expected_data_size(b - a, sec_nnz[s]);
    }
    measure_point(rank, measure::bitmaps, measure::MeasurementEvent::START);
    bitmap = opt.drop ? bitmap::compute_bitmap(ctx, part_A, opt.n_sections) : std::vector<bool>(opt.n_sections, true);
    measure_point(rank, measure::bitmaps, measure::MeasurementEvent::END);
  }

  // After relabelling, rows are put back in ascending column order.
  static void sort_rows(matrix::ManagedCSRMatrix<> &m) {
    std::vector<std::pair<midx_t, double>> tmp;
    for (midx_t r = 0; r < m.height; ++r) {
      midx_t a = m.row_ptr[r], b = m.row_ptr[r + 1];
      if (std::is_sorted(m.col_idx + a, m.col_idx + b)) continue;
      tmp.clear();
      for (midx_t j = a; j < b; ++j) {
        double v;
        std::memcpy(&v, reinterpret_cast<std::byte *>(m.values) + (size_t)j * 8, 8);
        tmp.push_back({m.col_idx[j], v});
      }
      std::stable_sort(tmp.begin(), tmp.end(), [](auto &x, auto &y) { return x.first < y.first; });
      for (midx_t j = a; j < b; ++j) {
        m.col_idx[j] = tmp[j - a].first;
        std::memcpy(reinterpret_cast<std::byte *>(m.values) + (size_t)j * 8, &tmp[j - a].second, 8);
      }
    }
  }

  ~HipSpgemm() override {
    if (d_col_new_id) spgemm_buffer_free(ctx, d_col_new_id);
    for (void *p : {recv.len, recv.col, recv.val, recv_ptr}) spgemm_buffer_free(ctx, p);
    for (auto &t : send)
      for (void *p : {t.len, t.col, t.val}) spgemm_buffer_free(ctx, p);
    spgemm_mat_free(ctx, C);
    spgemm_mat_free(ctx, part_A);
    spgemm_mat_free(ctx, block_B);
  }

  // Size of my B block in the reference's blocked serialization (header + section blobs).
  size_t get_B_serialization_size() override {
    size_t sz = matrix::blocked_header_size(opt.n_sections);
    for (auto b : sec_blob_size) sz += b;
    return sz;
  }
  // ... and filtered for every destination (reference src/mults/drop_parallel.cpp:31-36).
  std::vector<size_t> get_B_serialization_sizes() {
    std::vector<size_t> out(n_nodes);
    for (int d = 0; d < n_nodes; ++d) {
      size_t sz = matrix::blocked_header_size(opt.n_sections);
      for (uint32_t s = sec_lo(rank); s < sec_lo(rank + 1); ++s)
        if (bitmaps.at(d)[s]) sz += sec_blob_size[s];
      out[d] = sz;
    }
    return out;
  }

  void reset() override {
    spgemm_mat_free(ctx, C);
    C = nullptr;
  }

  // rows / entries this rank provides to `dest` in one gemm
  void provided(int dest, uint64_t *rows, uint64_t *nnz) const {
    *rows = *nnz = 0;
    for (uint32_t s = sec_lo(rank); s < sec_lo(rank + 1); ++s)
      if (bitmaps.at(dest)[s]) {
        uint64_t a, b;
        section_rows(s, &a, &b);
        *rows += b - a;
        *nnz += sec_nnz[s];
      }
  }

  // Allocates the exchange buffers; call once, after `bitmaps` holds every rank's bitmap.
  void prepare() {
    if (prepared) return;
    if (bitmaps.size() != (size_t)n_nodes) throw std::runtime_error("bitmaps were not shared before prepare");
    rows_from.assign(n_nodes, 0);
    nnz_from.assign(n_nodes, 0);
    for (int s = 0; s < n_nodes; ++s) {
      world.ops.at(s)->provided(rank, &rows_from[s], &nnz_from[s]);
      recv.rows += rows_from[s];
      recv.nnz += nnz_from[s];
    }
    HOST_TRY(spgemm_buffer_alloc(ctx, recv.rows * 4 + 4, &recv.len));
    HOST_TRY(spgemm_buffer_alloc(ctx, recv.nnz * 4 + 4, &recv.col));
    HOST_TRY(spgemm_buffer_alloc(ctx, recv.nnz * 8 + 8, &recv.val));
    HOST_TRY(spgemm_buffer_alloc(ctx, (b_rows_total + 1) * 8, &recv_ptr));
    send.assign(n_nodes, Triple());
    for (int d = 0; d < n_nodes; ++d) {
      if (d == rank) continue;
      provided(d, &send[d].rows, &send[d].nnz);
      if (!send[d].rows) continue;
      HOST_TRY(spgemm_buffer_alloc(ctx, send[d].rows * 4 + 4, &send[d].len));
      HOST_TRY(spgemm_buffer_alloc(ctx, send[d].nnz * 4 + 4, &send[d].col));
      HOST_TRY(spgemm_buffer_alloc(ctx, send[d].nnz * 8 + 8, &send[d].val));
    }
    prepared = true;
  }

  // Called by every rank's thread at the same time (the ranks meet in world.barrier()):
  //  1. filter: my sections that each destination's bitmap marks are packed into my send buffer for
  //     it (BlockedCSRMatrix::filter, reference include/matrix_impl.hpp:507-534); my own go straight
  //     into my receive buffer;
  //  2. barrier: every rank's send buffers are complete;
  //  3. the sections I need are copied from the owners' send buffers, ascending (peer to peer between
  //     GPUs), the row pointer of the local full-height B is rebuilt, one multiply follows.
  // The send buffers are written again only in the next gemm, which starts behind the driver's
  // barrier, i.e. after every rank has finished this one.
  void gemm(std::vector<size_t> /*serialized_sizes_B_bytes*/, size_t /*max_size_B_bytes*/) override {
    auto *ms = measure::Measure::get_instance(rank);
    prepare();
    measure_point(rank, measure::filter, measure::MeasurementEvent::START);
    uint64_t my_ro = 0, my_zo = 0; // where my own sections go in the receive buffer
    for (int s = 0; s < rank; ++s) { my_ro += rows_from[s]; my_zo += nnz_from[s]; }
    for (int d = 0; d < n_nodes; ++d) {
      auto words = bitmap::to_words(bitmaps[d]);
      uint64_t r = 0, z = 0;
      if (d == rank) {
        if (!rows_from[rank]) continue;
        HOST_TRY(spgemm_pack_sections(ctx, block_B, opt.n_sections, b_first_row, b_rows_total, words.data(),
                                      (uint32_t *)recv.len + my_ro, (uint32_t *)recv.col + my_zo, (double *)recv.val + my_zo,
                                      &r, &z));
        if (r != rows_from[rank] || z != nnz_from[rank]) throw std::runtime_error("section exchange: size mismatch");
      } else if (send[d].rows) {
        HOST_TRY(spgemm_pack_sections(ctx, block_B, opt.n_sections, b_first_row, b_rows_total, words.data(),
                                      (uint32_t *)send[d].len, (uint32_t *)send[d].col, (double *)send[d].val, &r, &z));
        if (r != send[d].rows || z != send[d].nnz) throw std::runtime_error("section exchange: size mismatch");
        ms->track_bytes(r * 4 + z * 12); // bytes this rank sends (counted at the sender, like communication::send)
      }
    }
    HOST_TRY(spgemm_synchronize(ctx));
    measure_point(rank, measure::filter, measure::MeasurementEvent::END);
    measure_point(rank, measure::wait_all, measure::MeasurementEvent::START);
    world.barrier();
    uint64_t ro = 0, zo = 0;
    for (int s = 0; s < n_nodes; ++s) {
      if (s != rank && rows_from[s]) {
        const Triple &t = world.ops[s]->send[rank];
        if (t.rows != rows_from[s] || t.nnz != nnz_from[s]) throw std::runtime_error("section exchange: size mismatch");
        HOST_TRY(spgemm_buffer_copy(ctx, (uint32_t *)recv.len + ro, world.ops[s]->ctx, t.len, t.rows * 4));
        HOST_TRY(spgemm_buffer_copy(ctx, (uint32_t *)recv.col + zo, world.ops[s]->ctx, t.col, t.nnz * 4));
        HOST_TRY(spgemm_buffer_copy(ctx, (double *)recv.val + zo, world.ops[s]->ctx, t.val, t.nnz * 8));
      }
      ro += rows_from[s];
      zo += nnz_from[s];
    }
    measure_point(rank, measure::wait_all, measure::MeasurementEvent::END);
    measure_point(rank, measure::deserialize, measure::MeasurementEvent::START);
    auto words = bitmap::to_words(bitmaps[rank]);
    HOST_TRY(spgemm_assemble_row_ptr(ctx, b_rows_total, opt.n_sections, words.data(), (const uint32_t *)recv.len, recv.rows,
                                     (uint64_t *)recv_ptr, nullptr));
    spgemm_mat *B_local = nullptr;
    HOST_TRY(spgemm_mat_wrap_device(ctx, b_rows_total, b_cols, recv.nnz, (const uint64_t *)recv_ptr, (const uint32_t *)recv.col,
                                    (const double *)recv.val, 0, &B_local));
    measure_point(rank, measure::deserialize, measure::MeasurementEvent::END);
    measure_point(rank, measure::mult, measure::MeasurementEvent::START);
    HOST_TRY(spgemm_multiply(ctx, part_A, B_local, &C));
    if (d_col_new_id) {
      spgemm_mat *shuffled = nullptr;
      HOST_TRY(spgemm_mat_relabel_columns(ctx, C, (const uint32_t *)d_col_new_id, b_cols, &shuffled));
      spgemm_mat_free(ctx, C);
      C = shuffled;
    }
    measure_point(rank, measure::mult, measure::MeasurementEvent::END);
    spgemm_stats st;
    HOST_TRY(spgemm_get_stats(ctx, &st));
    last_stats = st;
    ms->add_interval_ns(measure::analyze, st.ms_analyze * 1e6);
    ms->add_interval_ns(measure::symbolic, st.ms_symbolic * 1e6);
    ms->add_interval_ns(measure::numeric, st.ms_numeric * 1e6);
    spgemm_mat_free(ctx, B_local);
  }

  spgemm_stats last_stats{};

  static bool same_device(spgemm_ctx *a, spgemm_ctx *b) {
    int da = -1, db = -2;
    spgemm_ctx_device(a, &da);
    spgemm_ctx_device(b, &db);
    return da == db;
  }

  // C_<rank>.mtx: local row ids, shuffled-global column ids (reference src/mults/shared_methods.cpp:17-20).
  void save_result(std::string path) override {
    if (!C) throw std::runtime_error("save_result before gemm");
    uint64_t rows, cols, nnz;
    HOST_TRY(spgemm_mat_info(C, &rows, &cols, &nnz));
    std::vector<uint64_t> ptr(rows + 1);
    std::vector<uint32_t> col(nnz);
    std::vector<double> val(nnz);
    HOST_TRY(spgemm_mat_to_host(ctx, C, ptr.data(), col.data(), val.data()));
    std::vector<matrix::Cell<double>> lines(nnz);
    for (uint64_t r = 0; r < rows; ++r)
      for (uint64_t j = ptr[r]; j < ptr[r + 1]; ++j) lines[j] = {{(midx_t)r, col[j]}, val[j]};
    matrix::utils::write_matrix_market<double>(path, (midx_t)rows, (midx_t)cols, lines);
  }
};

} // namespace mults
