// Host-only check of the blocked B mirror (matrix::BlockedCSRMatrix): cuts the column block
// [col0, col1) of a Matrix Market file into row sections, filters it with a bitmap, views the
// filtered bytes again and compares every row with the unfiltered matrix.
//   blocked_check <file.mtx> <n_sections> <col0> <col1> <bitmap of 0/1>
// prints "<blob bytes> <filtered bytes> <rows checked>"; used by tests/test_host_blocked.py.
#include "matrix.hpp"

#include <iostream>

int main(int argc, char **argv) {
  if (argc != 6) { std::cerr << "usage: blocked_check file n_sections col0 col1 bitmap\n"; return 2; }
  try {
    const size_t n_sections = std::stoul(argv[2]);
    const midx_t col0 = (midx_t)std::stoul(argv[3]), col1 = (midx_t)std::stoul(argv[4]);
    const std::string bits = argv[5];
    if (bits.size() != n_sections) { std::cerr << "bitmap length != n_sections\n"; return 2; }
    std::vector<midx_t> keep_cols;
    for (midx_t c = col0; c < col1; ++c) keep_cols.push_back(c);
    matrix::ManagedCSRMatrix<> full(argv[1], false, nullptr, &keep_cols);
    matrix::BlockedCSRMatrix<> blocked(full, n_sections);
    std::vector<bool> bitmap(n_sections);
    for (size_t s = 0; s < n_sections; ++s) bitmap[s] = bits[s] == '1';
    std::vector<std::byte> sent = blocked.filter(bitmap);
    const size_t sent_size = sent.size();
    matrix::BlockedCSRMatrix<> recv(std::move(sent), n_sections); // what the receiver rebuilds
    size_t checked = 0;
    const size_t sh = full.height / n_sections;
    for (midx_t i = 0; i < full.height; ++i) {
      const size_t s = std::min<size_t>(i / sh, n_sections - 1);
      if (!bitmap[s]) {
        bool threw = false;
        try { (void)recv.row(i); } catch (const std::out_of_range &) { threw = true; }
        if (!threw) { std::cerr << "row " << i << " of a dropped section is present\n"; return 1; }
        continue;
      }
      auto a = full.row(i), b = recv.row(i), c = blocked.row(i);
      if (a.len != b.len || a.len != c.len) { std::cerr << "row " << i << ": length differs\n"; return 1; }
      for (midx_t j = 0; j < a.len; ++j) {
        double va, vb;
        std::memcpy(&va, reinterpret_cast<const std::byte *>(a.data) + (size_t)j * 8, 8);
        std::memcpy(&vb, reinterpret_cast<const std::byte *>(b.data) + (size_t)j * 8, 8);
        if (a.pos[j] != b.pos[j] || std::memcmp(&va, &vb, 8) != 0) { std::cerr << "row " << i << ": entry differs\n"; return 1; }
      }
      ++checked;
    }
    auto whole = blocked.serialize();
    std::cout << whole.second << " " << sent_size << " " << checked << std::endl;
    return 0;
  } catch (const std::exception &e) {
    std::cerr << "error: " << e.what() << std::endl;
    return 1;
  }
}
