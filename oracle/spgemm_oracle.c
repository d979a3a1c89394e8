/*
 * CPU oracle for the SpGEMM hot path.  TEST INFRASTRUCTURE ONLY: nothing in the
 * product path (spgemm_amd/, the C-ABI library) may include, link or call this
 * file.  It is the checker for tests/, for __graft_entry__.smoke() and for the
 * cpu_baseline leg of bench.py.
 *
 * It restates, in plain C, the algorithm of the reference's multiply path:
 *
 *   - the row-wise Gustavson triple loop of the Outer / Drop / DropParallel /
 *     DropAtOnce(Parallel) operators
 *       reference: src/mults/outer.cpp:61-73, src/mults/drop.cpp:76-88,
 *                  src/mults/drop_parallel.cpp:43-59,
 *                  src/mults/drop_at_once_parallel.cpp:120-131
 *   - the sparse accumulator Cells<T>::add (one map per output row; a duplicate
 *     position is summed; an entry whose running sum becomes exactly 0 is erased)
 *       reference: include/matrix_impl.hpp:81-95
 *   - the inner-product (merge-join) multiply of the Baseline operator
 *       reference: src/mults/baseline.cpp:48-71
 *   - the section bitmap
 *       reference: src/bitmap.cpp:12-40
 *
 * Pinning: the restatement is checked against the reference's own committed
 * fixtures (matrices/test: A, B, C_expected; matrices/first: A, C_expected),
 * copied as data under tests/golden/ref_test and tests/golden/ref_first.
 *
 * Deliberate differences from the reference (documented, not bugs):
 *   - the reference keeps std::unordered_map<col,{value,insertion index}> per
 *     row; this file uses an open-addressing table with the same observable
 *     behaviour (membership, summed value, erase-on-zero, insertion order).
 *   - the reference scatters by insertion index when it builds the CSR and an
 *     index can be reused after an erase (matrix_impl.hpp:85-93,308-312); the
 *     oracle never reuses an index: it emits the surviving entries either in
 *     first-insertion order or sorted by column.
 *   - the multiply is done in IEEE double with separate multiply and add (the
 *     reference is built with g++ -O3 for baseline x86-64, which has no fused
 *     multiply-add); this file must be built with -ffp-contract=off.
 */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef uint32_t midx_t; /* reference: include/matrix.hpp:9 */

#define SLOT_EMPTY 0u
#define SLOT_FULL 1u
#define SLOT_ERASED 2u

typedef struct {
  midx_t col;
  uint32_t state;
  uint64_t order; /* monotonically increasing insertion stamp */
  double val;
} slot_t;

typedef struct {
  slot_t *slots;
  size_t cap;  /* power of two */
  size_t used; /* FULL + ERASED */
  size_t live; /* FULL */
  uint64_t stamp;
} rowmap_t;

static void rowmap_init(rowmap_t *m) {
  m->cap = 64;
  m->slots = (slot_t *)calloc(m->cap, sizeof(slot_t));
  m->used = m->live = 0;
  m->stamp = 0;
}

static void rowmap_clear(rowmap_t *m) {
  if (m->used)
    memset(m->slots, 0, m->cap * sizeof(slot_t));
  m->used = m->live = 0;
  m->stamp = 0;
}

static inline size_t hash_col(midx_t c) {
  uint64_t x = (uint64_t)c * 0x9E3779B97F4A7C15ull;
  return (size_t)(x >> 20);
}

static slot_t *rowmap_find(rowmap_t *m, midx_t col, slot_t **free_slot) {
  size_t mask = m->cap - 1;
  size_t h = hash_col(col) & mask;
  slot_t *first_free = NULL;
  for (;;) {
    slot_t *s = &m->slots[h];
    if (s->state == SLOT_EMPTY) {
      if (!first_free)
        first_free = s;
      *free_slot = first_free;
      return NULL;
    }
    if (s->state == SLOT_ERASED) {
      if (!first_free)
        first_free = s;
    } else if (s->col == col) {
      return s;
    }
    h = (h + 1) & mask;
  }
}

static void rowmap_grow(rowmap_t *m) {
  slot_t *old = m->slots;
  size_t oldcap = m->cap;
  m->cap = oldcap * 2;
  m->slots = (slot_t *)calloc(m->cap, sizeof(slot_t));
  m->used = m->live = 0;
  for (size_t i = 0; i < oldcap; ++i) {
    if (old[i].state == SLOT_FULL) {
      slot_t *fs = NULL;
      rowmap_find(m, old[i].col, &fs);
      *fs = old[i];
      m->used++;
      m->live++;
    }
  }
  free(old);
}

/* Cells<T>::add, reference include/matrix_impl.hpp:81-95 (release build: the
 * assert(val != 0) is compiled out, so a zero first product is inserted). */
static inline void cells_add(rowmap_t *m, midx_t col, double val) {
  slot_t *fs = NULL;
  slot_t *s = rowmap_find(m, col, &fs);
  if (!s) {
    if (fs->state == SLOT_EMPTY)
      m->used++;
    fs->state = SLOT_FULL;
    fs->col = col;
    fs->val = val;
    fs->order = m->stamp++;
    m->live++;
    if (m->used * 2 > m->cap)
      rowmap_grow(m);
  } else {
    s->val += val;
    if (s->val == 0) {
      s->state = SLOT_ERASED;
      m->live--;
    }
  }
}

typedef struct {
  midx_t col;
  double val;
  uint64_t order;
} entry_t;

static int cmp_col(const void *a, const void *b) {
  midx_t x = ((const entry_t *)a)->col, y = ((const entry_t *)b)->col;
  return (x > y) - (x < y);
}
static int cmp_order(const void *a, const void *b) {
  uint64_t x = ((const entry_t *)a)->order, y = ((const entry_t *)b)->order;
  return (x > y) - (x < y);
}

typedef struct {
  uint64_t rows;
  uint64_t nnz;
  uint64_t nprod;
  double seconds_accumulate; /* wall time of the accumulate loop only */
  uint64_t *row_ptr;         /* rows + 1 */
  midx_t *col_idx;
  double *val;
} oracle_result_t;

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/*
 * Row-wise Gustavson into one map per output row, OpenMP over rows as in
 * DropParallel::mult_step (reference src/mults/drop_parallel.cpp:43-59):
 *
 *   for row in A: for a_ik in A.row(row): for b_kj in B.row(k):
 *       cells.add({row, col_offset + j}, a_ik * b_kj)
 *
 * The loop is run over `nblocks` B blocks in the given order (the reference
 * visits one B column block per ring round and all of them accumulate into
 * the same cells; block s adds partitions[s].start_col = col_offset[s]).
 * With nblocks == 1 and col_offset[0] == 0 it is the plain C = A * B.
 *
 * sort_cols != 0: each output row sorted by column; else first-insertion order.
 */
oracle_result_t *oracle_spgemm_blocks(uint64_t rows_a, const uint64_t *a_ptr,
                                      const midx_t *a_col, const double *a_val,
                                      int nblocks,
                                      const uint64_t *const *b_ptr,
                                      const midx_t *const *b_col,
                                      const double *const *b_val,
                                      const midx_t *col_offset, int sort_cols,
                                      int nthreads) {
  oracle_result_t *r = (oracle_result_t *)calloc(1, sizeof(*r));
  r->rows = rows_a;
  r->row_ptr = (uint64_t *)calloc(rows_a + 1, sizeof(uint64_t));
  entry_t **row_entries = (entry_t **)calloc(rows_a ? rows_a : 1, sizeof(entry_t *));
  uint64_t nprod_total = 0;
#ifdef _OPENMP
  if (nthreads > 0)
    omp_set_num_threads(nthreads);
#else
  (void)nthreads;
#endif
  double t0 = now_s();
#pragma omp parallel reduction(+ : nprod_total)
  {
    rowmap_t map;
    rowmap_init(&map);
#pragma omp for schedule(dynamic, 64)
    for (int64_t row = 0; row < (int64_t)rows_a; ++row) {
      rowmap_clear(&map);
      for (int s = 0; s < nblocks; ++s) {
        const uint64_t *bp = b_ptr[s];
        const midx_t *bc = b_col[s];
        const double *bv = b_val[s];
        midx_t off = col_offset[s];
        for (uint64_t ka = a_ptr[row]; ka < a_ptr[row + 1]; ++ka) {
          midx_t k = a_col[ka];
          double av = a_val[ka];
          for (uint64_t jb = bp[k]; jb < bp[k + 1]; ++jb) {
            double res = av * bv[jb];
            cells_add(&map, off + bc[jb], res);
            nprod_total++;
          }
        }
      }
      size_t n = map.live;
      r->row_ptr[row + 1] = n;
      if (n) {
        entry_t *e = (entry_t *)malloc(n * sizeof(entry_t));
        size_t w = 0;
        for (size_t i = 0; i < map.cap; ++i)
          if (map.slots[i].state == SLOT_FULL) {
            e[w].col = map.slots[i].col;
            e[w].val = map.slots[i].val;
            e[w].order = map.slots[i].order;
            ++w;
          }
        row_entries[row] = e;
      }
    }
    free(map.slots);
  }
  r->seconds_accumulate = now_s() - t0;
  r->nprod = nprod_total;
  for (uint64_t i = 0; i < rows_a; ++i)
    r->row_ptr[i + 1] += r->row_ptr[i];
  r->nnz = r->row_ptr[rows_a];
  r->col_idx = (midx_t *)malloc((r->nnz ? r->nnz : 1) * sizeof(midx_t));
  r->val = (double *)malloc((r->nnz ? r->nnz : 1) * sizeof(double));
#pragma omp parallel for schedule(dynamic, 256)
  for (int64_t row = 0; row < (int64_t)rows_a; ++row) {
    size_t n = (size_t)(r->row_ptr[row + 1] - r->row_ptr[row]);
    if (!n)
      continue;
    entry_t *e = row_entries[row];
    qsort(e, n, sizeof(entry_t), sort_cols ? cmp_col : cmp_order);
    uint64_t base = r->row_ptr[row];
    for (size_t i = 0; i < n; ++i) {
      r->col_idx[base + i] = e[i].col;
      r->val[base + i] = e[i].val;
    }
    free(e);
  }
  free(row_entries);
  return r;
}

oracle_result_t *oracle_spgemm(uint64_t rows_a, const uint64_t *a_ptr,
                               const midx_t *a_col, const double *a_val,
                               const uint64_t *b_ptr, const midx_t *b_col,
                               const double *b_val, int sort_cols,
                               int nthreads) {
  midx_t zero = 0;
  return oracle_spgemm_blocks(rows_a, a_ptr, a_col, a_val, 1, &b_ptr, &b_col,
                              &b_val, &zero, sort_cols, nthreads);
}

/*
 * A tuned CPU Gustavson with a dense sparse accumulator (SPA) per thread: value and stamp arrays
 * of one entry per column of B, a list of the touched columns.  Two passes (count, then fill), as
 * CPU SpGEMM libraries do.  NOT the reference's algorithm: it is the "fair second line" of the CPU
 * baseline next to the map-per-row port of reference src/mults/drop_parallel.cpp:43-59, and keeps an
 * entry iff its accumulated value is != 0.  Products are added in the same (A row, B row) order, so
 * the values are the reference's; rows come out sorted by column.
 * seconds_accumulate = both passes, without the per-row sort of the output.
 */
oracle_result_t *oracle_spgemm_spa(uint64_t rows_a, const uint64_t *a_ptr,
                                   const midx_t *a_col, const double *a_val,
                                   uint64_t cols_b, const uint64_t *b_ptr,
                                   const midx_t *b_col, const double *b_val,
                                   int nthreads) {
  oracle_result_t *r = (oracle_result_t *)calloc(1, sizeof(*r));
  r->rows = rows_a;
  r->row_ptr = (uint64_t *)calloc(rows_a + 1, sizeof(uint64_t));
  uint64_t nprod_total = 0;
#ifdef _OPENMP
  if (nthreads > 0)
    omp_set_num_threads(nthreads);
#else
  (void)nthreads;
#endif
  double t0 = now_s();
  /* pass 1: structural count per row */
#pragma omp parallel reduction(+ : nprod_total)
  {
    uint64_t *stamp = (uint64_t *)calloc(cols_b ? cols_b : 1, sizeof(uint64_t));
#pragma omp for schedule(dynamic, 64)
    for (int64_t row = 0; row < (int64_t)rows_a; ++row) {
      uint64_t n = 0;
      for (uint64_t ka = a_ptr[row]; ka < a_ptr[row + 1]; ++ka) {
        midx_t k = a_col[ka];
        for (uint64_t jb = b_ptr[k]; jb < b_ptr[k + 1]; ++jb) {
          midx_t c = b_col[jb];
          if (stamp[c] != (uint64_t)row + 1) {
            stamp[c] = (uint64_t)row + 1;
            ++n;
          }
          nprod_total++;
        }
      }
      r->row_ptr[row + 1] = n;
    }
    free(stamp);
  }
  for (uint64_t i = 0; i < rows_a; ++i)
    r->row_ptr[i + 1] += r->row_ptr[i];
  uint64_t cap = r->row_ptr[rows_a];
  midx_t *col = (midx_t *)malloc((cap ? cap : 1) * sizeof(midx_t));
  double *val = (double *)malloc((cap ? cap : 1) * sizeof(double));
  /* pass 2: accumulate and gather (touch order) */
#pragma omp parallel
  {
    uint64_t *stamp = (uint64_t *)calloc(cols_b ? cols_b : 1, sizeof(uint64_t));
    double *acc = (double *)malloc((cols_b ? cols_b : 1) * sizeof(double));
#pragma omp for schedule(dynamic, 64)
    for (int64_t row = 0; row < (int64_t)rows_a; ++row) {
      uint64_t base = r->row_ptr[row], n = 0;
      for (uint64_t ka = a_ptr[row]; ka < a_ptr[row + 1]; ++ka) {
        midx_t k = a_col[ka];
        double av = a_val[ka];
        for (uint64_t jb = b_ptr[k]; jb < b_ptr[k + 1]; ++jb) {
          midx_t c = b_col[jb];
          double p = av * b_val[jb];
          if (stamp[c] != (uint64_t)row + 1) {
            stamp[c] = (uint64_t)row + 1;
            acc[c] = p;
            col[base + n++] = c;
          } else {
            acc[c] += p;
          }
        }
      }
      for (uint64_t i = 0; i < n; ++i)
        val[base + i] = acc[col[base + i]];
    }
    free(stamp);
    free(acc);
  }
  r->seconds_accumulate = now_s() - t0;
  r->nprod = nprod_total;
  /* sort every row by column, drop exact zeros, compact (not timed) */
  uint64_t *keep = (uint64_t *)calloc(rows_a + 1, sizeof(uint64_t));
#pragma omp parallel for schedule(dynamic, 256)
  for (int64_t row = 0; row < (int64_t)rows_a; ++row) {
    uint64_t base = r->row_ptr[row], n = r->row_ptr[row + 1] - base;
    if (!n)
      continue;
    entry_t *e = (entry_t *)malloc(n * sizeof(entry_t));
    uint64_t w = 0;
    for (uint64_t i = 0; i < n; ++i)
      if (val[base + i] != 0) {
        e[w].col = col[base + i];
        e[w].val = val[base + i];
        e[w].order = 0;
        ++w;
      }
    qsort(e, w, sizeof(entry_t), cmp_col);
    for (uint64_t i = 0; i < w; ++i) {
      col[base + i] = e[i].col;
      val[base + i] = e[i].val;
    }
    keep[row + 1] = w;
    free(e);
  }
  for (uint64_t i = 0; i < rows_a; ++i)
    keep[i + 1] += keep[i];
  r->nnz = keep[rows_a];
  r->col_idx = (midx_t *)malloc((r->nnz ? r->nnz : 1) * sizeof(midx_t));
  r->val = (double *)malloc((r->nnz ? r->nnz : 1) * sizeof(double));
  for (uint64_t row = 0; row < rows_a; ++row) {
    uint64_t n = keep[row + 1] - keep[row];
    memcpy(r->col_idx + keep[row], col + r->row_ptr[row], n * sizeof(midx_t));
    memcpy(r->val + keep[row], val + r->row_ptr[row], n * sizeof(double));
  }
  memcpy(r->row_ptr, keep, (rows_a + 1) * sizeof(uint64_t));
  free(keep);
  free(col);
  free(val);
  return r;
}

/*
 * Baseline operator: inner-product SpGEMM over a TRANSPOSED B block (row j of
 * bt_* is column j of B), sorted merge-join of A row and B column, emits only
 * res != 0.  Reference src/mults/baseline.cpp:48-71.  Requires column indices
 * sorted inside every row of both operands (as the reference silently does).
 * Quadratic in rows_a * cols_b: small cases only.
 */
oracle_result_t *oracle_baseline_inner(uint64_t rows_a, const uint64_t *a_ptr,
                                       const midx_t *a_col, const double *a_val,
                                       uint64_t cols_b, const uint64_t *bt_ptr,
                                       const midx_t *bt_col,
                                       const double *bt_val,
                                       midx_t col_offset) {
  oracle_result_t *r = (oracle_result_t *)calloc(1, sizeof(*r));
  r->rows = rows_a;
  r->row_ptr = (uint64_t *)calloc(rows_a + 1, sizeof(uint64_t));
  size_t cap = 1024, n = 0;
  r->col_idx = (midx_t *)malloc(cap * sizeof(midx_t));
  r->val = (double *)malloc(cap * sizeof(double));
  double t0 = now_s();
  for (uint64_t row = 0; row < rows_a; ++row) {
    uint64_t row_len = a_ptr[row + 1] - a_ptr[row];
    const midx_t *row_pos = a_col + a_ptr[row];
    const double *row_data = a_val + a_ptr[row];
    for (uint64_t col = 0; col < cols_b; ++col) {
      uint64_t col_len = bt_ptr[col + 1] - bt_ptr[col];
      const midx_t *col_pos = bt_col + bt_ptr[col];
      const double *col_data = bt_val + bt_ptr[col];
      double res = 0;
      uint64_t ce = 0, re = 0;
      while (ce < col_len && re < row_len) {
        if (col_pos[ce] < row_pos[re]) {
          ce++;
        } else if (col_pos[ce] > row_pos[re]) {
          re++;
        } else {
          res += row_data[re] * col_data[ce];
          r->nprod++;
          re++;
          ce++;
        }
      }
      if (res != 0) {
        if (n == cap) {
          cap *= 2;
          r->col_idx = (midx_t *)realloc(r->col_idx, cap * sizeof(midx_t));
          r->val = (double *)realloc(r->val, cap * sizeof(double));
        }
        r->col_idx[n] = col_offset + (midx_t)col;
        r->val[n] = res;
        ++n;
      }
    }
    r->row_ptr[row + 1] = n;
  }
  r->seconds_accumulate = now_s() - t0;
  r->nnz = n;
  return r;
}

/*
 * Section bitmap, reference src/bitmap.cpp:12-40: bit s is set iff some stored
 * entry of the A block has  min(col / (width / n_sections), n_sections-1) == s.
 * words_out holds ceil(n_sections/64) 64-bit words, bit s = word s/64 bit s%64.
 * Returns -1 when width / n_sections == 0 (the reference asserts there).
 */
int oracle_section_bitmap(uint64_t nnz, const midx_t *a_col, midx_t width,
                          uint32_t n_sections, uint64_t *words_out) {
  midx_t section_width = width / n_sections;
  if (section_width == 0)
    return -1;
  size_t nwords = (n_sections + 63) / 64;
  memset(words_out, 0, nwords * sizeof(uint64_t));
  for (uint64_t i = 0; i < nnz; ++i) {
    midx_t s = a_col[i] / section_width;
    if (s > n_sections - 1)
      s = n_sections - 1;
    words_out[s >> 6] |= (uint64_t)1 << (s & 63);
  }
  return 0;
}

uint64_t oracle_result_rows(const oracle_result_t *r) { return r->rows; }
uint64_t oracle_result_nnz(const oracle_result_t *r) { return r->nnz; }
uint64_t oracle_result_nprod(const oracle_result_t *r) { return r->nprod; }
double oracle_result_seconds(const oracle_result_t *r) {
  return r->seconds_accumulate;
}
void oracle_result_fetch(const oracle_result_t *r, uint64_t *row_ptr,
                         midx_t *col_idx, double *val) {
  memcpy(row_ptr, r->row_ptr, (r->rows + 1) * sizeof(uint64_t));
  if (r->nnz) {
    memcpy(col_idx, r->col_idx, r->nnz * sizeof(midx_t));
    memcpy(val, r->val, r->nnz * sizeof(double));
  }
}
void oracle_result_free(oracle_result_t *r) {
  if (!r)
    return;
  free(r->row_ptr);
  free(r->col_idx);
  free(r->val);
  free(r);
}
int oracle_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/*
 * Test-run diagnostics (no counterpart in the reference): a handler for SIGABRT, SIGSEGV, SIGBUS,
 * SIGFPE and SIGILL that writes the signal, the faulting address, the id and name of the thread the
 * signal was delivered to and that thread's NATIVE call stack to standard error, then hands over to
 * the handler that was installed before it (Python's faulthandler shows Python frames only, and of
 * every thread alike, so a fault raised inside a C library or by a runtime thread cannot be told
 * apart from the log).
 */
#include <sys/prctl.h>
#include <sys/syscall.h>
#define N_TRACED 5
static const int traced_sig[N_TRACED] = {SIGABRT, SIGSEGV, SIGBUS, SIGFPE, SIGILL};
static struct sigaction traced_prev[N_TRACED];
static void put_str(const char *s) {
  ssize_t ignored = write(2, s, strlen(s));
  (void)ignored;
}
static void put_hex(uint64_t v) {
  char buf[19] = "0x";
  for (int i = 0; i < 16; ++i)
    buf[2 + i] = "0123456789abcdef"[(v >> (60 - 4 * i)) & 15];
  buf[18] = 0;
  put_str(buf);
}
static void fault_trace_handler(int sig, siginfo_t *info, void *uctx) {
  void *frames[64];
  char name[17] = {0};
  (void)uctx;
  put_str("\n--- fatal signal ");
  put_hex((uint64_t)sig);
  put_str(" at address ");
  put_hex((uint64_t)(uintptr_t)(info ? info->si_addr : 0));
  put_str(", thread ");
  put_hex((uint64_t)syscall(SYS_gettid));
  put_str(" of process ");
  put_hex((uint64_t)getpid());
  prctl(PR_GET_NAME, name, 0, 0, 0);
  put_str(" named '");
  put_str(name);
  put_str("': native stack ---\n");
  int n = backtrace(frames, 64);
  backtrace_symbols_fd(frames, n, 2);
  /* then whoever was there before (faulthandler, or the default action) */
  for (int i = 0; i < N_TRACED; ++i)
    if (traced_sig[i] == sig)
      sigaction(sig, &traced_prev[i], NULL);
  raise(sig);
}
int oracle_install_abort_trace(void) {
  struct sigaction sa;
  void *warm[4];
  int rc = 0;
  static int installed = 0;
  if (installed) /* a second install would save this handler as its own predecessor */
    return 0;
  installed = 1;
  backtrace(warm, 4); /* loads libgcc's unwinder now, not inside the handler */
  memset(&sa, 0, sizeof(sa));
  sa.sa_sigaction = fault_trace_handler;
  sigemptyset(&sa.sa_mask);
  sa.sa_flags = SA_NODEFER | SA_SIGINFO | SA_ONSTACK;
  for (int i = 0; i < N_TRACED; ++i)
    rc |= sigaction(traced_sig[i], &sa, &traced_prev[i]);
  return rc;
}
