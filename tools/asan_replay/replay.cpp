// Replays the operands and options of the GPU fuzz test (tools/asan_replay/dump_inputs.py) through
// the C ABI of include/spgemm_amd.h without Python and without the CPU checker: create a context,
// set the options, upload A and B, multiply twice, bring each result to the host, free everything,
// destroy the context.  Meant to be linked against a build of the library whose HOST code is
// instrumented with AddressSanitizer (tools/asan_replay/run.sh): any write of the library's host
// side past a heap or stack buffer ends the run with ASan's report.  Results are not checked here
// (tests/test_gpu_fuzz.py does that); a checksum is printed so that the run cannot be optimised away.
#include "../../include/spgemm_amd.h"

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

struct Csr {
  uint64_t rows = 0, cols = 0, nnz = 0;
  std::vector<uint64_t> ptr;
  std::vector<uint32_t> col;
  std::vector<double> val;
};

static bool get(FILE *f, void *p, size_t bytes) { return bytes == 0 || fread(p, 1, bytes, f) == bytes; }
static bool read_csr(FILE *f, Csr &m) {
  uint64_t h[3];
  if (!get(f, h, sizeof(h))) return false;
  m.rows = h[0]; m.cols = h[1]; m.nnz = h[2];
  m.ptr.resize(m.rows + 1); m.col.resize(m.nnz); m.val.resize(m.nnz);
  return get(f, m.ptr.data(), m.ptr.size() * 8) && get(f, m.col.data(), m.nnz * 4) && get(f, m.val.data(), m.nnz * 8);
}

#define CHECK(expr)                                                                       \
  do {                                                                                    \
    if ((expr) != SPGEMM_OK) {                                                            \
      fprintf(stderr, "case %d: %s failed: %s\n", ncase, #expr, spgemm_last_error());    \
      return 1;                                                                           \
    }                                                                                     \
  } while (0)

int main(int argc, char **argv) {
  if (argc < 2) { fprintf(stderr, "usage: replay <inputs.bin>\n"); return 2; }
  FILE *f = fopen(argv[1], "rb");
  if (!f) { perror(argv[1]); return 2; }
  int ncase = 0;
  unsigned long long sum = 0;
  for (;; ++ncase) {
    uint64_t nopt = 0;
    if (!get(f, &nopt, 8)) break;
    std::vector<std::pair<std::string, int64_t>> opts;
    for (uint64_t i = 0; i < nopt; ++i) {
      char name[33] = {0};
      int64_t v = 0;
      if (!get(f, name, 32) || !get(f, &v, 8)) { fprintf(stderr, "truncated input\n"); return 2; }
      opts.push_back({name, v});
    }
    Csr A, B;
    if (!read_csr(f, A) || !read_csr(f, B)) { fprintf(stderr, "truncated input\n"); return 2; }
    spgemm_ctx *ctx = nullptr;
    CHECK(spgemm_create(0, &ctx));
    for (auto &o : opts) CHECK(spgemm_set_option(ctx, o.first.c_str(), o.second));
    spgemm_mat *dA = nullptr, *dB = nullptr;
    CHECK(spgemm_mat_from_host(ctx, A.rows, A.cols, A.nnz, A.ptr.data(), A.col.data(), A.val.data(), &dA));
    CHECK(spgemm_mat_from_host(ctx, B.rows, B.cols, B.nnz, B.ptr.data(), B.col.data(), B.val.data(), &dB));
    for (int rep = 0; rep < 2; ++rep) {
      spgemm_mat *dC = nullptr;
      CHECK(spgemm_multiply(ctx, dA, dB, &dC));
      uint64_t rows, cols, nnz;
      CHECK(spgemm_mat_info(dC, &rows, &cols, &nnz));
      std::vector<uint64_t> ptr(rows + 1);
      std::vector<uint32_t> col(nnz);
      std::vector<double> val(nnz);
      CHECK(spgemm_mat_to_host(ctx, dC, ptr.data(), col.data(), val.data()));
      spgemm_stats st;
      CHECK(spgemm_get_stats(ctx, &st));
      sum += nnz + ptr[rows] + (nnz ? col[nnz - 1] : 0) + st.nprod;
      CHECK(spgemm_mat_free(ctx, dC));
    }
    if (ncase % 2) { CHECK(spgemm_mat_free(ctx, dA)); CHECK(spgemm_mat_free(ctx, dB)); } // the other half is left to spgemm_destroy
    CHECK(spgemm_destroy(ctx));
  }
  printf("%d cases replayed, checksum %llu\n", ncase, sum);
  return 0;
}
