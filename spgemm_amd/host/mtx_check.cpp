// Test helper (plain g++): reads a Matrix Market file with the host layer's reader and prints
// "nrows ncols entries checksum", the checksum being order-sensitive (entry i contributes
// (i + 1) * (row * 1000003 + col) and the value's bits rotated by i): the Python tests compute the
// same from their own parse.  Exit status 3 with the message on stderr when the reader throws.
#include "matrix.hpp"

#include <cstdio>

int main(int argc, char **argv) {
  if (argc < 2) return 2;
  try {
    auto t = matrix::utils::read_matrix_market<double>(argv[1]);
    unsigned long long h = 0;
    for (size_t i = 0; i < t.vals.size(); ++i) {
      unsigned long long bits;
      std::memcpy(&bits, &t.vals[i], 8);
      const unsigned s = (unsigned)(i % 63);
      h += (unsigned long long)(i + 1) * ((unsigned long long)t.rows[i] * 1000003ull + t.cols[i]);
      h ^= (bits << s) | (bits >> ((64 - s) % 64));
    }
    printf("%u %u %zu %llu\n", t.nrows, t.ncols, t.vals.size(), h);
    return 0;
  } catch (const std::exception &e) {
    fprintf(stderr, "%s\n", e.what());
    return 3;
  }
}
