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

#include <condition_variable>
#include <future>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#define HOST_TRY(expr)                                                                          \
  do {                                                                                          \
    if ((expr) != SPGEMM_OK) throw std::runtime_error(std::string(#expr) + ": " + spgemm_last_error()); \
  } while (0)

namespace bitmap {
// Section bitmap of an A block, computed on the device (reference src/bitmap.cpp:12-40).
inline void report_drop(const std::vector<bool> &bits) {
  size_t set = 0;
  for (bool b : bits) set += b;
  printf("Bitmap drop percentage: %g\n", (bits.size() - set) / (double)bits.size()); // parsed by scripts/drop_average.py:14
}
inline std::vector<bool> compute_bitmap(spgemm_ctx *ctx, const spgemm_mat *part_A, uint32_t n_sections, bool report = true) {
  std::vector<uint64_t> words((n_sections + 63) / 64, 0);
  HOST_TRY(spgemm_section_bitmap(ctx, part_A, n_sections, words.data()));
  std::vector<bool> bits(n_sections);
  for (uint32_t s = 0; s < n_sections; ++s) bits[s] = (words[s >> 6] >> (s & 63)) & 1ull;
  if (report) report_drop(bits);
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
// A meeting point of n threads that can be called off: once break_all() has been called nobody
// waits any more and every arrive_and_wait() returns false (a rank that failed must not leave the
// others waiting for it, which is what MPI_Abort sees to in the reference).
class Rendezvous {
  std::mutex m;
  std::condition_variable cv;
  int n, waiting = 0;
  uint64_t phase = 0;
  bool broken = false;

public:
  explicit Rendezvous(int n) : n(n) {}
  bool arrive_and_wait() {
    std::unique_lock<std::mutex> lk(m);
    if (broken) return false;
    if (++waiting == n) {
      waiting = 0;
      ++phase;
      cv.notify_all();
      return true;
    }
    const uint64_t p = phase;
    cv.wait(lk, [&] { return phase != p || broken; });
    return phase != p;
  }
  void break_all() {
    std::lock_guard<std::mutex> lk(m);
    broken = true;
    cv.notify_all();
  }
};

// Stand-in for MPI_COMM_WORLD inside one process: the ranks are host threads (one per rank, as
// the reference's ranks are processes, src/main.cpp:35-45), barrier() is MPI_Barrier.
struct World {
  int n_nodes = 0;
  std::vector<spgemm_ctx *> ctx;          // one per rank (ranks may share a GPU): the multiply
  std::vector<spgemm_ctx *> xctx;         // one more per rank with a stream of its own: the exchange of the staged schedule (null: not staged)
  std::vector<mults::HipSpgemm *> ops;    // registered by the operators
  Rendezvous run_start;                   // of the rank threads, before every run
  Rendezvous exchange;                    // of the threads that carry the exchange: every rank's send buffers are packed
  explicit World(int n, int n_gpus, bool staged = false)
      : n_nodes(n), ctx(n, nullptr), xctx(n, nullptr), ops(n, nullptr), run_start(n), exchange(n) {
    for (int r = 0; r < n; ++r) {
      HOST_TRY(spgemm_create(r % n_gpus, &ctx[r]));
      if (staged) HOST_TRY(spgemm_create(r % n_gpus, &xctx[r]));
    }
  }
  void barrier() {
    if (!run_start.arrive_and_wait()) throw std::runtime_error("another rank failed");
  }
  void exchange_barrier() {
    if (!exchange.arrive_and_wait()) throw std::runtime_error("another rank failed");
  }
  // called by a rank that gives up: nobody waits for it any more
  void abandon() {
    run_start.break_all();
    exchange.break_all();
  }
  ~World() {
    for (auto c : xctx) spgemm_destroy(c);
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
  // > 1: the pipelined schedule.  The rank's rows are cut into `stages` blocks, each with a section
  // bitmap and an exchange of its own; the exchange of block s + 1 runs on a second context (its own
  // stream, carried by a helper thread) while block s is multiplied: the overlap the reference's
  // ring gets from Isend / Irecv around mult_step (src/mults/drop_parallel.cpp:79-114).  Every row
  // of C is still computed by one full-width multiply, so the result is the same, bit for bit.
  uint32_t stages = 1;
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
//  * part_A: rows keep_rows of A (the reference loads them the same way, src/main.cpp:213-217), cut
//    into `stages` blocks of consecutive rows;
//  * B: this rank owns the row sections [rank*N/P, (rank+1)*N/P) of B with ALL columns; the column
//    order col_order (the whole B_shuffle: column p of the result stands for original column
//    col_order[p]; the reference passes only its slice, keep_cols, because it cuts B by columns) is
//    applied to the result;
//  * gemm(): per block of rows, pulls the sections the block's bitmap marks from every rank, in
//    ascending order, straight into the col_idx/val arrays of a full-height local B, then one
//    multiply gives the block's rows of C (local row ids, shuffled-global column ids: exactly what
//    the reference's C_<rank>.mtx holds).
class HipSpgemm : public MatrixMultiplication {
  communication::World &world;
  spgemm_ctx *ctx;  // multiplies
  spgemm_ctx *xctx; // packs, copies and assembles: `ctx` itself unless the schedule is staged
  HipOptions opt;
  spgemm_mat *block_B = nullptr; // my row sections
  void *d_col_new_id = nullptr;  // device: new column id of every original column (null: identity)
  uint64_t b_first_row = 0, b_rows_total = 0, b_cols = 0;
  std::vector<uint64_t> sec_nnz;       // entries of each of MY sections (index: global section id)
  std::vector<uint64_t> sec_blob_size; // reference blob size of each of my sections
  midx_t a_height = 0;
  // exchange buffers, allocated once by prepare() (the reference allocates its receive buffers inside
  // every gemm, src/mults/drop_parallel.cpp:65-68): what I receive (= the arrays of my local B) and,
  // per destination, what I send
  struct Triple { void *len = nullptr, *col = nullptr, *val = nullptr; uint64_t rows = 0, nnz = 0; };
  struct Stage {
    spgemm_mat *A = nullptr;                // the block's rows of part_A
    uint64_t rows = 0;
    std::vector<bool> bitmap;               // sections the block needs
    std::vector<std::vector<bool>> bitmaps; // ... and what the same block of every rank needs
    Triple recv;
    void *recv_ptr = nullptr;
    std::vector<Triple> send;               // index: destination rank (mine stays empty)
    std::vector<uint64_t> rows_from, nnz_from;
    spgemm_mat *C = nullptr;
  };
  std::vector<Stage> stage;
  bool prepared = false;
  struct ExchangeCost { double filter_ns = 0, wait_ns = 0, deserialize_ns = 0; uint64_t bytes = 0; };

  static void upload(spgemm_ctx *ctx, const matrix::CSRMatrix<> &m, midx_t r0, midx_t r1, spgemm_mat **out) {
    std::vector<uint64_t> ptr(r1 - r0 + 1);
    for (midx_t i = r0; i <= r1; ++i) ptr[i - r0] = m.row_ptr[i] - m.row_ptr[r0];
    const size_t j0 = m.row_ptr[r0], nz = m.row_ptr[r1] - m.row_ptr[r0];
    std::vector<double> val(nz); // the blob's values may be 4-byte aligned only
    std::memcpy(val.data(), reinterpret_cast<const std::byte *>(m.values) + j0 * sizeof(double), nz * sizeof(double));
    HOST_TRY(spgemm_mat_from_host(ctx, r1 - r0, m.width, nz, ptr.data(), m.col_idx + j0, val.data(), out));
  }

  uint32_t sec_lo(int r) const { return (uint32_t)((uint64_t)r * opt.n_sections / n_nodes); }
  void section_rows(uint32_t s, uint64_t *r0, uint64_t *r1) const {
    uint64_t sh = b_rows_total / opt.n_sections;
    *r0 = sh * s;
    *r1 = (s + 1 == opt.n_sections) ? b_rows_total : sh * (s + 1);
  }
  static double ns_since(std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count();
  }

public:
  std::vector<bool> bitmap;               // reference include/mults.hpp:129 (std::bitset<N_SECTIONS>): sections any of my blocks needs
  std::vector<std::vector<bool>> bitmaps; // all ranks', filled by share_bitmaps (src/main.cpp:253-258)

  HipSpgemm(int rank, int n_nodes, partition::Partitions partitions, const matrix::Triplets<double> &tA,
            const std::vector<midx_t> *keep_rows, const matrix::Triplets<double> &tB,
            const std::vector<midx_t> *col_order, communication::World &world, HipOptions opt)
      : MatrixMultiplication(rank, n_nodes, std::move(partitions)), world(world), ctx(world.ctx.at(rank)),
        xctx(world.xctx.at(rank) ? world.xctx.at(rank) : world.ctx.at(rank)), opt(opt) {
    if (opt.stages < 1) throw std::runtime_error("stages must be at least 1");
    if (opt.stages > 1 && !world.xctx.at(rank)) throw std::runtime_error("a staged schedule needs a World made for it");
    world.ops.at(rank) = this;
    matrix::ManagedCSRMatrix<> hA(tA, false, keep_rows, nullptr);
    a_height = hA.height;
    stage.resize(opt.stages);
    for (uint32_t s = 0; s < opt.stages; ++s) { // equal shares of the rows
      const midx_t r0 = (midx_t)((uint64_t)hA.height * s / opt.stages), r1 = (midx_t)((uint64_t)hA.height * (s + 1) / opt.stages);
      stage[s].rows = r1 - r0;
      upload(ctx, hA, r0, r1, &stage[s].A);
    }
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
    upload(ctx, hB, 0, hB.height, &block_B);
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
    bitmap.assign(opt.n_sections, !opt.drop);
    for (auto &st : stage) {
      st.bitmap = opt.drop ? bitmap::compute_bitmap(ctx, st.A, opt.n_sections, false) : std::vector<bool>(opt.n_sections, true);
      for (uint32_t s = 0; s < opt.n_sections; ++s) bitmap[s] = bitmap[s] || st.bitmap[s];
    }
    if (opt.drop) bitmap::report_drop(bitmap);
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
    for (auto &st : stage) {
      for (void *p : {st.recv.len, st.recv.col, st.recv.val, st.recv_ptr}) spgemm_buffer_free(ctx, p);
      for (auto &t : st.send)
        for (void *p : {t.len, t.col, t.val}) spgemm_buffer_free(ctx, p);
      spgemm_mat_free(ctx, st.C);
      spgemm_mat_free(ctx, st.A);
    }
    spgemm_mat_free(ctx, block_B);
  }

  // "Allgather" of the bitmaps (src/main.cpp:253-258): every block's, of every rank.
  void share_bitmaps(const std::vector<std::unique_ptr<HipSpgemm>> &all) {
    bitmaps.assign(n_nodes, {});
    for (int r = 0; r < n_nodes; ++r) {
      if (all.at(r)->stage.size() != stage.size()) throw std::runtime_error("the ranks cut their rows into different numbers of blocks");
      bitmaps[r] = all[r]->bitmap;
    }
    for (size_t s = 0; s < stage.size(); ++s) {
      stage[s].bitmaps.assign(n_nodes, {});
      for (int r = 0; r < n_nodes; ++r) stage[s].bitmaps[r] = all[r]->stage[s].bitmap;
    }
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
    for (auto &st : stage) {
      spgemm_mat_free(ctx, st.C);
      st.C = nullptr;
    }
  }

  // rows / entries this rank provides to block `sg` of `dest` in one gemm
  void provided(int dest, size_t sg, uint64_t *rows, uint64_t *nnz) const {
    *rows = *nnz = 0;
    for (uint32_t s = sec_lo(rank); s < sec_lo(rank + 1); ++s)
      if (stage.at(sg).bitmaps.at(dest)[s]) {
        uint64_t a, b;
        section_rows(s, &a, &b);
        *rows += b - a;
        *nnz += sec_nnz[s];
      }
  }

  // Allocates the exchange buffers; call once, after share_bitmaps.
  void prepare() {
    if (prepared) return;
    if (bitmaps.size() != (size_t)n_nodes) throw std::runtime_error("bitmaps were not shared before prepare");
    for (size_t sg = 0; sg < stage.size(); ++sg) {
      Stage &st = stage[sg];
      st.rows_from.assign(n_nodes, 0);
      st.nnz_from.assign(n_nodes, 0);
      for (int s = 0; s < n_nodes; ++s) {
        world.ops.at(s)->provided(rank, sg, &st.rows_from[s], &st.nnz_from[s]);
        st.recv.rows += st.rows_from[s];
        st.recv.nnz += st.nnz_from[s];
      }
      HOST_TRY(spgemm_buffer_alloc(ctx, st.recv.rows * 4 + 4, &st.recv.len));
      HOST_TRY(spgemm_buffer_alloc(ctx, st.recv.nnz * 4 + 4, &st.recv.col));
      HOST_TRY(spgemm_buffer_alloc(ctx, st.recv.nnz * 8 + 8, &st.recv.val));
      HOST_TRY(spgemm_buffer_alloc(ctx, (b_rows_total + 1) * 8, &st.recv_ptr));
      st.send.assign(n_nodes, Triple());
      for (int d = 0; d < n_nodes; ++d) {
        if (d == rank) continue;
        provided(d, sg, &st.send[d].rows, &st.send[d].nnz);
        if (!st.send[d].rows) continue;
        HOST_TRY(spgemm_buffer_alloc(ctx, st.send[d].rows * 4 + 4, &st.send[d].len));
        HOST_TRY(spgemm_buffer_alloc(ctx, st.send[d].nnz * 4 + 4, &st.send[d].col));
        HOST_TRY(spgemm_buffer_alloc(ctx, st.send[d].nnz * 8 + 8, &st.send[d].val));
      }
    }
    // The first pack of a matrix builds the host copy of its row pointer and its row lengths on the
    // handle; done here, on the matrix's own context, so that the exchange context only ever queues copies.
    {
      std::vector<uint64_t> none((opt.n_sections + 63) / 64, 0);
      uint64_t r = 0, z = 0;
      HOST_TRY(spgemm_pack_sections(ctx, block_B, opt.n_sections, b_first_row, b_rows_total, none.data(), nullptr, nullptr, nullptr, &r, &z));
      HOST_TRY(spgemm_synchronize(ctx));
    }
    prepared = true;
  }

  // The exchange of block sg; called by every rank at the same time (the callers meet in
  // world.exchange_barrier()), on xctx:
  //  1. filter: my sections that each destination's bitmap marks are packed into my send buffer for
  //     it (BlockedCSRMatrix::filter, reference include/matrix_impl.hpp:507-534); my own go straight
  //     into my receive buffer;
  //  2. barrier: every rank's send buffers are complete;
  //  3. the sections I need are copied from the owners' send buffers, ascending (peer to peer between
  //     GPUs), and the row pointer of the local full-height B is rebuilt.
  // Returns with the block's local B complete in st.recv / st.recv_ptr.  The send buffers of a block
  // are written again only in the next gemm, which starts behind the driver's barrier, i.e. after
  // every rank has finished this one.
  ExchangeCost exchange(size_t sg) {
    Stage &st = stage.at(sg);
    ExchangeCost cost;
    auto t0 = std::chrono::steady_clock::now();
    uint64_t my_ro = 0, my_zo = 0; // where my own sections go in the receive buffer
    for (int s = 0; s < rank; ++s) { my_ro += st.rows_from[s]; my_zo += st.nnz_from[s]; }
    for (int d = 0; d < n_nodes; ++d) {
      auto words = bitmap::to_words(st.bitmaps[d]);
      uint64_t r = 0, z = 0;
      if (d == rank) {
        if (!st.rows_from[rank]) continue;
        HOST_TRY(spgemm_pack_sections(xctx, block_B, opt.n_sections, b_first_row, b_rows_total, words.data(),
                                      (uint32_t *)st.recv.len + my_ro, (uint32_t *)st.recv.col + my_zo,
                                      (double *)st.recv.val + my_zo, &r, &z));
        if (r != st.rows_from[rank] || z != st.nnz_from[rank]) throw std::runtime_error("section exchange: size mismatch");
      } else if (st.send[d].rows) {
        HOST_TRY(spgemm_pack_sections(xctx, block_B, opt.n_sections, b_first_row, b_rows_total, words.data(),
                                      (uint32_t *)st.send[d].len, (uint32_t *)st.send[d].col, (double *)st.send[d].val, &r, &z));
        if (r != st.send[d].rows || z != st.send[d].nnz) throw std::runtime_error("section exchange: size mismatch");
        cost.bytes += r * 4 + z * 12; // bytes this rank sends (counted at the sender, like communication::send)
      }
    }
    HOST_TRY(spgemm_synchronize(xctx));
    cost.filter_ns = ns_since(t0);
    t0 = std::chrono::steady_clock::now();
    world.exchange_barrier();
    uint64_t ro = 0, zo = 0;
    for (int s = 0; s < n_nodes; ++s) {
      if (s != rank && st.rows_from[s]) {
        HipSpgemm *peer = world.ops[s];
        const Triple &t = peer->stage.at(sg).send[rank];
        if (t.rows != st.rows_from[s] || t.nnz != st.nnz_from[s]) throw std::runtime_error("section exchange: size mismatch");
        HOST_TRY(spgemm_buffer_copy(xctx, (uint32_t *)st.recv.len + ro, peer->xctx, t.len, t.rows * 4));
        HOST_TRY(spgemm_buffer_copy(xctx, (uint32_t *)st.recv.col + zo, peer->xctx, t.col, t.nnz * 4));
        HOST_TRY(spgemm_buffer_copy(xctx, (double *)st.recv.val + zo, peer->xctx, t.val, t.nnz * 8));
      }
      ro += st.rows_from[s];
      zo += st.nnz_from[s];
    }
    cost.wait_ns = ns_since(t0);
    t0 = std::chrono::steady_clock::now();
    auto words = bitmap::to_words(st.bitmaps[rank]);
    HOST_TRY(spgemm_assemble_row_ptr(xctx, b_rows_total, opt.n_sections, words.data(), (const uint32_t *)st.recv.len,
                                     st.recv.rows, (uint64_t *)st.recv_ptr, nullptr));
    HOST_TRY(spgemm_synchronize(xctx)); // the multiply runs on another stream
    cost.deserialize_ns = ns_since(t0);
    return cost;
  }

  // The multiply of block sg over the local B its exchange left behind.
  void multiply_stage(size_t sg) {
    Stage &st = stage.at(sg);
    spgemm_mat *B_local = nullptr;
    HOST_TRY(spgemm_mat_wrap_device(ctx, b_rows_total, b_cols, st.recv.nnz, (const uint64_t *)st.recv_ptr,
                                    (const uint32_t *)st.recv.col, (const double *)st.recv.val, 0, &B_local));
    const int rc = spgemm_multiply(ctx, st.A, B_local, &st.C);
    if (rc == SPGEMM_OK && d_col_new_id) {
      spgemm_mat *shuffled = nullptr;
      const int rc2 = spgemm_mat_relabel_columns(ctx, st.C, (const uint32_t *)d_col_new_id, b_cols, &shuffled);
      if (rc2 != SPGEMM_OK) { spgemm_mat_free(ctx, B_local); HOST_TRY(rc2); }
      spgemm_mat_free(ctx, st.C);
      st.C = shuffled;
    }
    spgemm_mat_free(ctx, B_local);
    HOST_TRY(rc);
    spgemm_stats s;
    HOST_TRY(spgemm_get_stats(ctx, &s));
    if (sg == 0) last_stats = s;
    else {
      last_stats.nprod += s.nprod;
      last_stats.nnz_c += s.nnz_c;
      last_stats.ms_analyze += s.ms_analyze;
      last_stats.ms_symbolic += s.ms_symbolic;
      last_stats.ms_scan += s.ms_scan;
      last_stats.ms_numeric += s.ms_numeric;
      last_stats.ms_compact += s.ms_compact;
      last_stats.ms_total += s.ms_total;
    }
  }

  // One run.  stages == 1: exchange, then multiply, on one context.  Staged: the exchange of block
  // s + 1 is carried by a helper thread on the exchange context while this thread multiplies block s.
  void gemm(std::vector<size_t> /*serialized_sizes_B_bytes*/, size_t /*max_size_B_bytes*/) override {
    auto *ms = measure::Measure::get_instance(rank);
    prepare();
    if (const char *f = std::getenv("DPHPC_AMD_FAIL_RANK")) // test hook: this rank gives up inside a run
      if (std::atoi(f) == rank) throw std::runtime_error("failure requested through DPHPC_AMD_FAIL_RANK");
    auto account = [&](const ExchangeCost &c) {
      ms->add_interval_ns(measure::filter, c.filter_ns);
      ms->add_interval_ns(measure::wait_all, c.wait_ns);
      ms->add_interval_ns(measure::deserialize, c.deserialize_ns);
      ms->track_bytes(c.bytes);
    };
    std::future<ExchangeCost> next;
    try {
      if (stage.size() > 1) next = std::async(std::launch::async, [this] { return exchange(0); });
      for (size_t sg = 0; sg < stage.size(); ++sg) {
        account(stage.size() > 1 ? next.get() : exchange(sg));
        if (sg + 1 < stage.size()) next = std::async(std::launch::async, [this, sg] { return exchange(sg + 1); });
        measure_point(rank, measure::mult, measure::MeasurementEvent::START);
        multiply_stage(sg);
        measure_point(rank, measure::mult, measure::MeasurementEvent::END);
      }
    } catch (...) {
      // a helper that is still on its way must not outlive this call: nobody waits for this rank any
      // more, so it comes back (with an error of its own, which is dropped)
      world.abandon();
      if (next.valid()) {
        try { next.get(); } catch (...) {}
      }
      throw;
    }
    ms->add_interval_ns(measure::analyze, last_stats.ms_analyze * 1e6);
    ms->add_interval_ns(measure::symbolic, last_stats.ms_symbolic * 1e6);
    ms->add_interval_ns(measure::numeric, last_stats.ms_numeric * 1e6);
  }

  spgemm_stats last_stats{};

  // C_<rank>.mtx: local row ids, shuffled-global column ids (reference src/mults/shared_methods.cpp:17-20).
  void save_result(std::string path) override {
    std::vector<matrix::Cell<double>> lines;
    uint64_t row0 = 0, cols = b_cols;
    for (auto &st : stage) {
      if (!st.C) throw std::runtime_error("save_result before gemm");
      uint64_t rows, c, nnz;
      HOST_TRY(spgemm_mat_info(st.C, &rows, &c, &nnz));
      cols = c;
      std::vector<uint64_t> ptr(rows + 1);
      std::vector<uint32_t> col(nnz);
      std::vector<double> val(nnz);
      HOST_TRY(spgemm_mat_to_host(ctx, st.C, ptr.data(), col.data(), val.data()));
      for (uint64_t r = 0; r < rows; ++r)
        for (uint64_t j = ptr[r]; j < ptr[r + 1]; ++j) lines.push_back({{(midx_t)(row0 + r), col[j]}, val[j]});
      row0 += rows;
    }
    matrix::utils::write_matrix_market<double>(path, (midx_t)row0, (midx_t)cols, lines);
  }
};

} // namespace mults
