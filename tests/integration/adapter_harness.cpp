// Runs the adapter that the repository's integration guide documents (extracted into adapter_extracted.hpp by
// tools/extract_adapter.py) on the GPU: a rank's rows of A against two column blocks of B that
// arrive as filtered blocked blobs, two runs with different section bitmaps; the entries it emits
// must be the bits of the plain triple loop of the reference's mult_step
// (src/mults/drop_parallel.cpp:43-59).  Prints OK and returns 0 on success.
#include "ref_stubs.hpp"

#include "adapter_extracted.hpp"

#include <algorithm>
#include <map>

struct Csr {
  midx_t h = 0, w = 0;
  std::vector<midx_t> ptr, col;
  std::vector<double> val;
};

static uint64_t rng_state = 88172645463325252ull;
static uint64_t rng() {
  rng_state ^= rng_state << 13;
  rng_state ^= rng_state >> 7;
  rng_state ^= rng_state << 17;
  return rng_state;
}

static Csr random_csr(midx_t h, midx_t w, unsigned per_row, midx_t col_lo, midx_t col_hi) {
  Csr m;
  m.h = h; m.w = w;
  m.ptr.push_back(0);
  for (midx_t r = 0; r < h; ++r) {
    std::vector<midx_t> cs;
    const unsigned n = (unsigned)(rng() % (per_row + 1));
    for (unsigned i = 0; i < n; ++i) cs.push_back(col_lo + (midx_t)(rng() % (col_hi - col_lo)));
    std::sort(cs.begin(), cs.end());
    cs.erase(std::unique(cs.begin(), cs.end()), cs.end());
    for (midx_t c : cs) {
      m.col.push_back(c);
      m.val.push_back((double)((int64_t)(rng() % 2001) - 1000) / 7.0);
    }
    m.ptr.push_back((midx_t)m.col.size());
  }
  return m;
}

// blob of a plain CSR matrix, as ManagedCSRMatrix holds it
static std::vector<std::byte> csr_blob(const Csr &m) {
  std::vector<std::byte> out(matrix::CSRMatrix<>::expected_data_size(m.h, m.col.size()), std::byte(0));
  matrix::Fields f{false, m.h, m.w, (midx_t)m.col.size()};
  std::memcpy(out.data(), &f, sizeof(f));
  midx_t *rp = (midx_t *)(out.data() + sizeof(f));
  std::memcpy(rp, m.ptr.data(), m.ptr.size() * sizeof(midx_t));
  std::memcpy(rp + m.ptr.size(), m.col.data(), m.col.size() * sizeof(midx_t));
  std::memcpy((std::byte *)(rp + m.ptr.size()) + m.col.size() * sizeof(midx_t), m.val.data(), m.val.size() * sizeof(double));
  return out;
}

int main() {
  const midx_t H = 300, K = 1003, W0 = 700, W1 = 541; // K rows of B: sections of 125 rows, the last one 128
  const partition::Partitions parts = {{0, H, 0, W0}, {0, H, W0, W0 + W1}};
  int failures = 0;
  for (int run = 0; run < 2; ++run) {
    // run 0: A reaches into sections 1..5 of B; run 1: into sections 0..2
    rng_state = 88172645463325252ull + 977ull * run;
    Csr A = random_csr(H, K, 9, run == 0 ? 125 : 0, run == 0 ? 750 : 375);
    Csr B0 = random_csr(K, W0, 12, 0, W0), B1 = random_csr(K, W1, 5, 0, W1);
    std::bitset<N_SECTIONS> bitmap;
    for (midx_t c : A.col) bitmap[std::min<size_t>(c / (K / N_SECTIONS), N_SECTIONS - 1)] = true;
    auto blobA = csr_blob(A);
    matrix::CSRMatrix<> part_A(matrix::Data{blobA.data(), blobA.size()});
    mults::HipBlocks op(0, 2, parts, part_A, 1);
    // three gemm() calls of one operator (reset() + the ring's two steps each):
    // rep 0: first use, the device buffers are allocated and wrapped;
    // rep 1: other values in the same pattern: buffers and handle are reused (spgemm_mat_invalidate);
    // rep 2: another pattern with more entries: the buffers are allocated anew
    for (int rep = 0; rep < 3; ++rep) {
      if (rep == 1) {
        for (auto &v : B0.val) v = -v * 3.0;
        for (auto &v : B1.val) v *= 0.5;
      }
      if (rep == 2) B0 = random_csr(K, W0, 20, 0, W0);
      op.reset();
      auto blob0 = matrix::filtered_blob(K, W0, B0.ptr, B0.col, B0.val, bitmap);
      auto blob1 = matrix::filtered_blob(K, W1, B1.ptr, B1.col, B1.val, bitmap);
      op.mult_step(blob0.data(), blob0.size(), 0);
      op.mult_step(blob1.data(), blob1.size(), 1);
      std::map<std::pair<midx_t, midx_t>, double> got, want;
      op.for_each_entry([&](midx_t r, midx_t c, double v) { got[{r, c}] = v; });
      for (int blk = 0; blk < 2; ++blk) { // the reference's loop
        const Csr &B = blk ? B1 : B0;
        for (midx_t r = 0; r < H; ++r)
          for (midx_t ja = A.ptr[r]; ja < A.ptr[r + 1]; ++ja)
            for (midx_t jb = B.ptr[A.col[ja]]; jb < B.ptr[A.col[ja] + 1]; ++jb) {
              const double res = A.val[ja] * B.val[jb];
              auto key = std::make_pair(r, parts[blk].start_col + B.col[jb]);
              auto it = want.find(key);
              if (it == want.end()) want[key] = res; else it->second += res;
            }
      }
      for (auto it = want.begin(); it != want.end();) // the library's result holds no exact zeros (spgemm_stats::nnz_c)
        it = it->second == 0.0 ? want.erase(it) : std::next(it);
      bool same = got.size() == want.size();
      if (!same) std::cout << "  entries: got " << got.size() << ", expected " << want.size() << std::endl;
      for (auto &kv : want) {
        auto it = got.find(kv.first);
        if (it == got.end() || std::memcmp(&it->second, &kv.second, 8) != 0) {
          std::cout << "  first difference at (" << kv.first.first << ", " << kv.first.second << "): expected " << kv.second;
          if (it == got.end()) std::cout << ", missing" << std::endl; else std::cout << ", got " << it->second << std::endl;
          same = false;
          break;
        }
      }
      std::cout << "run " << run << " rep " << rep << ": sections " << bitmap.count() << " of " << N_SECTIONS << ", entries "
                << want.size() << (same ? ": same bits" : ": DIFFERENT") << std::endl;
      failures += !same;
    }
  }
  if (failures) return 1;
  std::cout << "OK" << std::endl;
  return 0;
}
