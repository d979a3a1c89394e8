// Host-side Matrix/CSR abstraction of the SpGEMM driver.
//
// It keeps the interface of the reference's matrix layer (include/matrix.hpp,
// include/matrix_impl.hpp): Fields, SmallVec, CSRMatrix (a view over one packed blob
// [Fields][row_ptr][col_idx][values], 8-byte rounded, so it can be shipped as bytes),
// ManagedCSRMatrix (owns the blob, loads a Matrix Market file with keep_rows / keep_cols),
// BlockedCSRMatrix (the same matrix cut into row sections, with filter(bitmap)).
// The arithmetic does not live here: operators hand these matrices to the GPU through the C ABI
// of include/spgemm_amd.h.
#pragma once

#include <algorithm>
#include <charconv>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <unordered_map>
#include <utility>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#define ROUND8(x) (((x) + 7) & ~(size_t)7)

typedef uint32_t midx_t; // reference include/matrix.hpp:9

namespace matrix {

template <typename T = double> struct SmallVec { // reference include/matrix.hpp:13-17
  const T *data;
  const midx_t *pos;
  const midx_t len;
};

using CellPos = std::pair<midx_t, midx_t>;
template <typename T = double> using Cell = std::pair<CellPos, T>;
using Data = std::pair<std::byte *, size_t>;

struct alignas(8) Fields { // reference include/matrix.hpp:26-31 (16 bytes)
  bool transposed;
  midx_t height;
  midx_t width;
  midx_t non_zeros;
};
static_assert(sizeof(Fields) == 16, "Fields must keep the reference's 16-byte layout");

template <typename T> struct Triplets {
  midx_t nrows = 0, ncols = 0;
  std::vector<midx_t> rows, cols;
  std::vector<T> vals;
};

namespace utils {

// Matrix Market `coordinate real|integer|pattern general|symmetric` reader (the reference maps the
// file and hands it to a multi-threaded parser, include/matrix_impl.hpp:158-226).  The file is mapped,
// the entry lines are cut into one piece per thread at line ends, every thread parses its piece with
// std::from_chars (correctly rounded, no locale, no terminating NUL needed) and the pieces are joined in
// file order, so rows keep the order in which the file lists their entries.  Symmetric input is
// expanded.  Threads: SPGEMM_IO_THREADS, else OMP_NUM_THREADS, else the hardware's, at most 64.
namespace detail {
struct MappedFile {
  const char *p = nullptr;
  size_t n = 0;
  int fd = -1;
  explicit MappedFile(const std::string &path) {
    fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) throw std::runtime_error("could not open file (reading): " + path);
    struct stat st;
    if (fstat(fd, &st) != 0) { ::close(fd); throw std::runtime_error("could not stat " + path); }
    n = (size_t)st.st_size;
    if (n) {
      void *m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
      if (m == MAP_FAILED) { ::close(fd); throw std::runtime_error("could not map " + path); }
      p = static_cast<const char *>(m);
    }
  }
  ~MappedFile() {
    if (p) munmap(const_cast<char *>(p), n);
    if (fd >= 0) ::close(fd);
  }
  MappedFile(const MappedFile &) = delete;
  MappedFile &operator=(const MappedFile &) = delete;
};
inline unsigned io_threads() {
  for (const char *name : {"SPGEMM_IO_THREADS", "OMP_NUM_THREADS"})
    if (const char *e = std::getenv(name)) {
      const long v = std::strtol(e, nullptr, 10);
      if (v >= 1) return (unsigned)std::min<long>(v, 64);
    }
  return std::min(64u, std::max(1u, std::thread::hardware_concurrency()));
}
inline const char *skip_blank(const char *p, const char *end) {
  while (p < end && (*p == ' ' || *p == '\t' || *p == '\r')) ++p;
  return p;
}
// the entry lines of [p, end) (whole lines); false on a malformed line
template <typename T>
bool parse_entries(const char *p, const char *end, bool pattern, bool symmetric, std::vector<midx_t> &rows,
                   std::vector<midx_t> &cols, std::vector<T> &vals, size_t *entries) {
  size_t n = 0;
  while (p < end) {
    p = skip_blank(p, end);
    if (p < end && *p != '\n' && *p != '%') {
      unsigned long long r = 0, c = 0;
      auto a = std::from_chars(p, end, r);
      if (a.ec != std::errc()) return false;
      p = skip_blank(a.ptr, end);
      a = std::from_chars(p, end, c);
      if (a.ec != std::errc() || r == 0 || c == 0) return false;
      p = a.ptr;
      double v = 1.0;
      if (!pattern) {
        p = skip_blank(p, end);
        if (p < end && *p == '+') ++p; // (from_chars takes no leading plus sign)
        auto d = std::from_chars(p, end, v);
        if (d.ec == std::errc::invalid_argument) return false;
        p = d.ptr; // (out of range: the nearest representable value was stored, as strtod does)
      }
      rows.push_back((midx_t)(r - 1)); cols.push_back((midx_t)(c - 1)); vals.push_back((T)v);
      if (symmetric && r != c) { rows.push_back((midx_t)(c - 1)); cols.push_back((midx_t)(r - 1)); vals.push_back((T)v); }
      ++n;
    }
    while (p < end && *p != '\n') ++p; // rest of the line (a complex or extra field is ignored)
    if (p < end) ++p;
  }
  *entries = n;
  return true;
}
} // namespace detail

template <typename T> Triplets<T> read_matrix_market(const std::string &path) {
  detail::MappedFile file(path);
  const char *p = file.p, *end = file.p + file.n;
  auto line_end = [&](const char *q) { while (q < end && *q != '\n') ++q; return q; };
  const char *le = line_end(p);
  std::string header(p, le);
  for (auto &ch : header) ch = (char)tolower(ch);
  if (header.rfind("%%matrixmarket matrix coordinate", 0) != 0)
    throw std::runtime_error("unsupported Matrix Market header in " + path);
  const bool symmetric = header.find("symmetric") != std::string::npos;
  const bool pattern = header.find("pattern") != std::string::npos;
  p = le < end ? le + 1 : end;
  while (p < end && (*p == '%' || *p == '\n' || *p == '\r')) p = std::min(end, line_end(p) + 1);
  // the size line
  unsigned long long dims[3] = {0, 0, 0};
  for (auto &d : dims) {
    p = detail::skip_blank(p, end);
    auto a = std::from_chars(p, end, d);
    if (a.ec != std::errc()) throw std::runtime_error("bad size line in " + path);
    p = a.ptr;
  }
  p = std::min(end, line_end(p) + 1);
  Triplets<T> t;
  t.nrows = (midx_t)dims[0];
  t.ncols = (midx_t)dims[1];
  const size_t nnz = dims[2];
  // one piece per thread, cut at line ends (a piece of at least 1 MiB: small files take one thread)
  const size_t body = (size_t)(end - p);
  const unsigned nt = (unsigned)std::max<size_t>(1, std::min<size_t>(detail::io_threads(), body >> 20));
  std::vector<const char *> cut(nt + 1, end);
  cut[0] = p;
  for (unsigned i = 1; i < nt; ++i) {
    const char *q = std::max(p + body * i / nt, cut[i - 1]);
    const char *e = line_end(q);
    cut[i] = e < end ? e + 1 : end;
  }
  struct Piece { std::vector<midx_t> rows, cols; std::vector<T> vals; size_t entries = 0; bool ok = true; };
  std::vector<Piece> piece(nt);
  auto work = [&](unsigned i) {
    Piece &pc = piece[i];
    const size_t guess = (size_t)((double)nnz * (double)(cut[i + 1] - cut[i]) / (double)std::max<size_t>(body, 1) * 1.05) + 16;
    pc.rows.reserve(symmetric ? 2 * guess : guess);
    pc.cols.reserve(symmetric ? 2 * guess : guess);
    pc.vals.reserve(symmetric ? 2 * guess : guess);
    pc.ok = detail::parse_entries<T>(cut[i], cut[i + 1], pattern, symmetric, pc.rows, pc.cols, pc.vals, &pc.entries);
  };
  {
    std::vector<std::thread> th;
    for (unsigned i = 1; i < nt; ++i) th.emplace_back(work, i);
    work(0);
    for (auto &x : th) x.join();
  }
  size_t entries = 0, total = 0;
  for (auto &pc : piece) {
    if (!pc.ok) throw std::runtime_error("malformed entry line in Matrix Market file " + path);
    entries += pc.entries;
    total += pc.vals.size();
  }
  if (entries < nnz) throw std::runtime_error("truncated Matrix Market file " + path);
  if (entries > nnz) throw std::runtime_error("more entry lines than the size line announces in " + path);
  t.rows.resize(total); t.cols.resize(total); t.vals.resize(total);
  size_t at = 0;
  for (auto &pc : piece) { // (in file order)
    std::copy(pc.rows.begin(), pc.rows.end(), t.rows.begin() + at);
    std::copy(pc.cols.begin(), pc.cols.end(), t.cols.begin() + at);
    std::copy(pc.vals.begin(), pc.vals.end(), t.vals.begin() + at);
    at += pc.vals.size();
  }
  return t;
}

// Header-only read (reference src/matrix.cpp:16-44).
inline Fields read_fields(const std::string &path, bool transposed, const std::vector<midx_t> *keep_rows,
                          const std::vector<midx_t> *keep_cols) {
  std::ifstream f(path);
  if (!f.is_open()) throw std::runtime_error("could not open file (reading): " + path);
  std::string line;
  std::getline(f, line);
  while (std::getline(f, line)) if (!line.empty() && line[0] != '%') break;
  unsigned long long r = 0, c = 0, n = 0;
  if (sscanf(line.c_str(), "%llu %llu %llu", &r, &c, &n) != 3) throw std::runtime_error("bad size line in " + path);
  if (transposed) std::swap(r, c);
  if (keep_rows) r = keep_rows->size();
  if (keep_cols) c = keep_cols->size();
  return Fields{transposed, (midx_t)r, (midx_t)c, (midx_t)n};
}

// Sorted by (row, col), one-based, like the reference (src/matrix.cpp:46-64); values are written
// with 17 significant digits (the reference's default stream precision keeps 6).
template <typename T>
void write_matrix_market(const std::string &path, midx_t height, midx_t width, std::vector<Cell<T>> &lines) {
  FILE *f = fopen(path.c_str(), "w");
  if (!f) throw std::runtime_error("could not open file (writing): " + path);
  fprintf(f, "%%%%MatrixMarket matrix coordinate real general\n%u %u %zu\n", height, width, lines.size());
  std::sort(lines.begin(), lines.end());
  for (auto &[pos, val] : lines) fprintf(f, "%u %u %.17g\n", pos.first + 1, pos.second + 1, (double)val);
  fclose(f);
}

} // namespace utils

// CSR view over a packed blob (reference include/matrix_impl.hpp:230-363).
template <typename T = double> class CSRMatrix {
protected:
  std::byte *data = nullptr;
  size_t size = 0;

  void bind() {
    Fields *fields = reinterpret_cast<Fields *>(data);
    height = fields->height; width = fields->width; non_zeros = fields->non_zeros; transposed = fields->transposed;
    if (size != expected_data_size(height, non_zeros)) throw std::runtime_error("CSR blob has the wrong size");
    row_ptr = reinterpret_cast<midx_t *>(data + sizeof(Fields));
    col_idx = row_ptr + (height + 1);
    values = reinterpret_cast<T *>(reinterpret_cast<std::byte *>(col_idx) + (size_t)non_zeros * sizeof(midx_t));
  }

public:
  midx_t height = 0, width = 0, non_zeros = 0;
  bool transposed = false;
  midx_t *row_ptr = nullptr;
  midx_t *col_idx = nullptr;
  T *values = nullptr; // may be only 4-byte aligned, as in the reference: copy out before wide loads

  static size_t expected_data_size(size_t h, size_t nnz) {
    return ROUND8(sizeof(Fields) + (h + 1) * sizeof(midx_t) + nnz * sizeof(midx_t) + nnz * sizeof(T));
  }

  CSRMatrix() = default;
  explicit CSRMatrix(const Data &d) : data(d.first), size(d.second) { bind(); }

  SmallVec<T> row(midx_t i) const {
    if (transposed || i >= height) throw std::out_of_range("CSRMatrix::row");
    auto o = row_ptr[i];
    return {values + o, col_idx + o, row_ptr[i + 1] - o};
  }
  SmallVec<T> col(midx_t j) const {
    if (!transposed || j >= height) throw std::out_of_range("CSRMatrix::col");
    auto o = row_ptr[j];
    return {values + o, col_idx + o, row_ptr[j + 1] - o};
  }

  void save(const std::string &path) const {
    std::vector<Cell<T>> lines(non_zeros);
    size_t l = 0;
    for (midx_t r = 0; r < height; ++r)
      for (midx_t j = row_ptr[r]; j < row_ptr[r + 1]; ++j, ++l) {
        T v;
        std::memcpy(&v, reinterpret_cast<const std::byte *>(values) + (size_t)j * sizeof(T), sizeof(T));
        lines[l] = transposed ? Cell<T>{{col_idx[j], r}, v} : Cell<T>{{r, col_idx[j]}, v};
      }
    utils::write_matrix_market<T>(path, height, width, lines);
  }

  Data serialize() const { return {data, size}; } // borrowed: do not write
};

// Owns its blob (reference include/matrix_impl.hpp:365-384).  Rows keep the order in which the
// file lists their entries, the order the reference's single-threaded loader inserts in (:198-221).
template <typename T = double> class ManagedCSRMatrix : public CSRMatrix<T> {
  std::unique_ptr<std::byte[]> storage;

  void build(midx_t h, midx_t w, bool tr, const std::vector<midx_t> &r, const std::vector<midx_t> &c,
             const std::vector<T> &v) {
    size_t nnz = v.size();
    size_t sz = CSRMatrix<T>::expected_data_size(h, nnz);
    storage.reset(new std::byte[sz]());
    this->data = storage.get();
    this->size = sz;
    Fields fs{tr, h, w, (midx_t)nnz};
    std::memcpy(this->data, &fs, sizeof(fs));
    this->bind();
    std::vector<midx_t> cnt(h + 1, 0);
    for (auto x : r) cnt[x + 1]++;
    for (midx_t i = 0; i < h; ++i) cnt[i + 1] += cnt[i];
    std::memcpy(this->row_ptr, cnt.data(), (h + 1) * sizeof(midx_t));
    std::vector<midx_t> cur(cnt.begin(), cnt.end() - 1);
    for (size_t i = 0; i < nnz; ++i) {
      midx_t p = cur[r[i]]++;
      this->col_idx[p] = c[i];
      std::memcpy(reinterpret_cast<std::byte *>(this->values) + (size_t)p * sizeof(T), &v[i], sizeof(T));
    }
  }

public:
  ManagedCSRMatrix() = default;
  // keep_rows / keep_cols: a kept row (column) gets its index in the list as its new id
  // (reference include/matrix_impl.hpp:163-217).
  ManagedCSRMatrix(const std::string &file_path, bool tr = false, const std::vector<midx_t> *keep_rows = nullptr,
                   const std::vector<midx_t> *keep_cols = nullptr) {
    auto t = utils::read_matrix_market<T>(file_path);
    from_triplets(t, tr, keep_rows, keep_cols);
  }
  ManagedCSRMatrix(const Triplets<T> &t, bool tr, const std::vector<midx_t> *keep_rows,
                   const std::vector<midx_t> *keep_cols) {
    from_triplets(t, tr, keep_rows, keep_cols);
  }
  ManagedCSRMatrix(midx_t h, midx_t w, const std::vector<midx_t> &r, const std::vector<midx_t> &c,
                   const std::vector<T> &v, bool tr = false) {
    build(h, w, tr, r, c, v);
  }

  void from_triplets(const Triplets<T> &t, bool tr, const std::vector<midx_t> *keep_rows,
                     const std::vector<midx_t> *keep_cols) {
    midx_t h = tr ? t.ncols : t.nrows, w = tr ? t.nrows : t.ncols;
    std::vector<midx_t> rmap, cmap;
    const midx_t NONE = 0xFFFFFFFFu;
    if (keep_rows) {
      rmap.assign(h, NONE);
      for (size_t i = 0; i < keep_rows->size(); ++i) rmap.at((*keep_rows)[i]) = (midx_t)i;
      h = (midx_t)keep_rows->size();
    }
    if (keep_cols) {
      cmap.assign(w, NONE);
      for (size_t i = 0; i < keep_cols->size(); ++i) cmap.at((*keep_cols)[i]) = (midx_t)i;
      w = (midx_t)keep_cols->size();
    }
    std::vector<midx_t> r, c;
    std::vector<T> v;
    for (size_t i = 0; i < t.vals.size(); ++i) {
      midx_t row = tr ? t.cols[i] : t.rows[i], col = tr ? t.rows[i] : t.cols[i];
      if (keep_rows) { row = rmap[row]; if (row == NONE) continue; }
      if (keep_cols) { col = cmap[col]; if (col == NONE) continue; }
      r.push_back(row); c.push_back(col); v.push_back(t.vals[i]);
    }
    build(h, w, tr, r, c, v);
  }
};

// Header of a blocked blob (reference include/matrix.hpp:63-68), with the section count chosen at
// run time: [height][n_sections][section_offst[N]][section_start_row[N]], all size_t.
inline size_t blocked_header_size(size_t n_sections) { return sizeof(size_t) * (2 + 2 * n_sections); }

// A matrix cut into n_sections row sections of height rows / n_sections, the last one taking the
// remainder (reference include/matrix_impl.hpp:388-538, 621-639).
template <typename T = double> class BlockedCSRMatrix {
  std::vector<std::byte> blob;
  size_t n_sections_ = 0;
  std::vector<CSRMatrix<T>> csrs;
  std::vector<size_t> start_rows;
  std::vector<int> slot_of_section; // per section of the unfiltered matrix: its position in csrs, -1 when filtered away

  size_t *hdr() { return reinterpret_cast<size_t *>(blob.data()); }
  const size_t *hdr() const { return reinterpret_cast<const size_t *>(blob.data()); }

  void bind_sections() {
    height = (midx_t)hdr()[0];
    n_sections_ = hdr()[1];
    const size_t *offs = hdr() + 2, *starts = hdr() + 2 + full_sections; // header slots: full count
    csrs.clear(); start_rows.clear(); non_zeros = 0;
    for (size_t i = 0; i < n_sections_; ++i) {
      size_t b = offs[i], e = (i + 1 == n_sections_) ? blob.size() : offs[i + 1];
      csrs.emplace_back(Data{blob.data() + b, e - b});
      start_rows.push_back(starts[i]);
      non_zeros += csrs.back().non_zeros;
    }
    width = n_sections_ ? csrs[0].width : 0;
    slot_of_section.assign(full_sections, -1);
    const size_t sh = full_sections ? height / full_sections : 0;
    for (size_t i = 0; i < n_sections_ && sh; ++i)
      slot_of_section[std::min<size_t>(start_rows[i] / sh, full_sections - 1)] = (int)i;
  }

public:
  midx_t height = 0, width = 0, non_zeros = 0;
  size_t full_sections = 0; // section count of the unfiltered matrix (row -> section rule)

  BlockedCSRMatrix() = default;

  // Cut `full` (all rows of a B block) into sections.
  BlockedCSRMatrix(const CSRMatrix<T> &full, size_t n_sections) : full_sections(n_sections) {
    if (n_sections < 2 || full.height / n_sections == 0)
      throw std::runtime_error("BlockedCSRMatrix: fewer rows than sections");
    size_t sh = full.height / n_sections;
    blob.assign(blocked_header_size(n_sections), std::byte(0));
    std::vector<size_t> offs(n_sections), starts(n_sections);
    for (size_t s = 0; s < n_sections; ++s) {
      size_t r0 = sh * s, r1 = (s + 1 == n_sections) ? full.height : sh * (s + 1);
      size_t z0 = full.row_ptr[r0], z1 = full.row_ptr[r1];
      size_t sz = CSRMatrix<T>::expected_data_size(r1 - r0, z1 - z0);
      offs[s] = blob.size(); starts[s] = r0;
      blob.resize(blob.size() + sz, std::byte(0));
      std::byte *d = blob.data() + offs[s];
      Fields fs{false, (midx_t)(r1 - r0), full.width, (midx_t)(z1 - z0)};
      std::memcpy(d, &fs, sizeof(fs));
      midx_t *rp = reinterpret_cast<midx_t *>(d + sizeof(Fields));
      for (size_t r = r0; r <= r1; ++r) rp[r - r0] = (midx_t)(full.row_ptr[r] - z0);
      std::memcpy(rp + (r1 - r0 + 1), full.col_idx + z0, (z1 - z0) * sizeof(midx_t));
      std::memcpy(reinterpret_cast<std::byte *>(rp + (r1 - r0 + 1)) + (z1 - z0) * sizeof(midx_t),
                  reinterpret_cast<const std::byte *>(full.values) + z0 * sizeof(T), (z1 - z0) * sizeof(T));
    }
    hdr()[0] = full.height; hdr()[1] = n_sections;
    std::memcpy(hdr() + 2, offs.data(), n_sections * sizeof(size_t));
    std::memcpy(hdr() + 2 + n_sections, starts.data(), n_sections * sizeof(size_t));
    bind_sections();
  }

  // View over received bytes (reference :480-485); `full_sections_` is the sender's section count.
  BlockedCSRMatrix(std::vector<std::byte> bytes, size_t full_sections_) : blob(std::move(bytes)), full_sections(full_sections_) {
    bind_sections();
  }

  size_t n_sections() const { return n_sections_; }
  const CSRMatrix<T> &section(size_t i) const { return csrs.at(i); }
  size_t section_start_row(size_t i) const { return start_rows.at(i); }

  // Row i of the whole matrix; throws std::out_of_range when its section was filtered away, like
  // the reference's unordered_map::at (:487-493).
  SmallVec<T> row(midx_t i) const {
    size_t sh = height / full_sections;
    size_t s = std::min<size_t>(i / sh, full_sections - 1), start = s * sh;
    const int k = slot_of_section.at(s); // (one table look-up, as the reference's map from start row to section)
    if (k >= 0) return csrs[(size_t)k].row((midx_t)(i - start));
    throw std::out_of_range("BlockedCSRMatrix::row: section is not present");
  }

  // Keep the sections whose bit is set (reference :507-534).
  std::vector<std::byte> filter(const std::vector<bool> &bitmap) const {
    size_t keep = 0;
    for (size_t i = 0; i < n_sections_; ++i) keep += bitmap.at(i) ? 1 : 0;
    // the header keeps the FULL section count worth of slots, as the reference's fixed-size
    // BlockedFields does; only the first `keep` are meaningful
    std::vector<std::byte> out(blocked_header_size(full_sections), std::byte(0));
    std::vector<size_t> offs(full_sections, 0), starts(full_sections, 0);
    size_t j = 0;
    for (size_t i = 0; i < n_sections_; ++i) {
      if (!bitmap[i]) continue;
      auto [p, sz] = csrs[i].serialize();
      offs[j] = out.size(); starts[j] = start_rows[i];
      out.insert(out.end(), p, p + sz);
      ++j;
    }
    size_t *h = reinterpret_cast<size_t *>(out.data());
    h[0] = height; h[1] = keep;
    std::memcpy(h + 2, offs.data(), full_sections * sizeof(size_t));
    std::memcpy(h + 2 + full_sections, starts.data(), full_sections * sizeof(size_t));
    return out;
  }

  Data serialize() { return {blob.data(), blob.size()}; }
};

} // namespace matrix
