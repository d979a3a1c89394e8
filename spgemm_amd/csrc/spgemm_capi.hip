// C-ABI of the gfx950 SpGEMM engine (include/spgemm_amd.h): contexts, device matrices, the
// symbolic -> scan -> numeric pipeline, and the partition / bitmap / exchange helpers.
#include "../../include/spgemm_amd.h"
#include "kernels.h"
#include "internal.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <numeric>
#include <random>
#include <string>
#include <unordered_set>
#include <vector>

using namespace sg;

// ---- errors -------------------------------------------------------------------------------------
static thread_local std::string g_err;

static int fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIP_TRY(expr)                                                                               \
  do {                                                                                              \
    hipError_t e_ = (expr);                                                                         \
    if (e_ != hipSuccess)                                                                           \
      return fail(e_ == hipErrorOutOfMemory ? SPGEMM_ERR_NOMEM : SPGEMM_ERR_HIP, "%s failed: %s (%s:%d)", \
                  #expr, hipGetErrorString(e_), __FILE__, __LINE__);                                 \
  } while (0)

#define TRY(expr)                                                                                   \
  do {                                                                                              \
    int rc_ = (expr);                                                                               \
    if (rc_ != SPGEMM_OK) return rc_;                                                               \
  } while (0)

// ---- objects ------------------------------------------------------------------------------------
struct spgemm_mat {
  uint64_t rows = 0, cols = 0, nnz = 0;
  uint64_t *d_ptr = nullptr;
  uint32_t *d_col = nullptr;
  double *d_val = nullptr;
  bool owns = false;
  int device = 0;
  uint32_t rowinfo_sh = 1;    // layout of d_rowinfo (RowInfo): 1 = halves side by side, 0 = in two arrays
  uint4 *d_rowinfo = nullptr; // {min col, max col, length} per row, built on first use as B
  float *d_colcdf = nullptr;  // CDF_BINS + 1 prefix counts of the entries per column range (+ the histogram behind them)
  uint32_t cdf_shift = 0;     // column range b holds the columns c with c >> cdf_shift == b
  size_t bytes_ptr = 0, bytes_col = 0, bytes_val = 0, bytes_rowinfo = 0, bytes_colcdf = 0;
  uint64_t *d_prow = nullptr; // scratch of the padded-record layout (row starts, scan partials, counts)
  size_t bytes_prow = 0;
  uint4 *d_prec = nullptr;    // rows of up to PREC_MAX_LEN entries as padded 16-byte records (short-row numeric kernel)
  size_t bytes_prec = 0;
  uint64_t prec_slots = 0;    // record slots in d_prec; the padded column array follows them
  Rec12 *d_packed = nullptr;  // the entries as 12-byte records {column, value}, read by the big-row numeric kernel (option)
  size_t bytes_packed = 0;
  uint32_t *d_samp = nullptr; // last column of every block of 32 entries (the big-row numeric kernel searches over it)
  size_t bytes_samp = 0;
  uint32_t *d_wb = nullptr;   // big-row symbolic kernel: first entry of every row in every window of 2^wb_ws columns (k_b_winbounds)
  size_t bytes_wb = 0;
  uint32_t wb_nwin = 0, wb_ws = 0;
  // used by the section helpers of the B exchange, built on first use, dropped by spgemm_mat_invalidate:
  std::vector<uint64_t> h_ptr;  // host copy of the row pointer
  uint32_t *d_rowlen = nullptr; // row lengths as uint32 (what travels in the exchange)
  size_t bytes_rowlen = 0;
};

struct Block {
  void *p;
  size_t bytes;
};

struct Growable {
  void *p = nullptr;
  size_t cap = 0;
};

struct spgemm_ctx {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  hipEvent_t ev[6] = {};
  ClassParams cp;
  uint32_t long_len = 2048;      // A rows longer than this are analysed by a whole workgroup
  uint32_t row_groups = 1;       // accumulate runs of identical A rows together
  uint32_t flat_walk = 1;        // rows with short B rows: lay products end to end over the lanes
  uint32_t global_blocks_sym = 512;
  uint32_t global_blocks_num = 2048;
  Growable arena;     // per-multiply row arrays
  Growable temp_cols; // sorted structural columns (symbolic -> numeric)
  Growable gscratch;  // bitmap scratch of the global path (kept zero between uses)
  size_t gscratch_zeroed = 0;
  Growable big_cur;   // big-row symbolic: {position, end} per A entry (rows longer than BIG_CUR)
  Growable slice_buf; // big-row numeric: item counts, offsets, items, per-workgroup lists
  uint32_t big_rows = 1;         // big rows of a sorted B: LDS bitmap windows / output slices
  uint32_t slice_blocks = 0;     // workgroups of the slice kernel (0: fill the device)
  bool hw_lane_order = false;    // this device passed k_probe_lane_order when the context was made
  uint32_t lane_order_adds = 1;  // option: 0 = never rely on it (claim rounds instead)
  uint32_t slice_queues = 1;     // work queues of the slice kernel (8: one per XCD; 1: one for the device)
  uint32_t grp_compact = 2;      // row-group numeric kernel: two-level bitmap (1), flat (0), chosen per launch (2)
  uint32_t short_sym_spread = 1; // short-row symbolic kernel: log2 of (table slots / 2 / products)
  uint32_t analyze_lanes = 0;    // lanes per A row in the analysis (8, 16, 32; 0: chosen by the mean row length)
  uint32_t short_key32 = 1;      // 1: 32-bit table entries in the short-row numeric kernel where B's columns allow
  uint32_t walk_tail = 2;        // scalar walk of the window kernels fetches entries 64..127 of a B row ahead: 0 no, 1 yes,
                                 // 2 where B's rows hold 48 entries and more on average
  uint32_t short_packed = 2;     // 1: the short-row numeric kernel reads B's rows as padded 16-byte records; 2: and the
                                 // count-only symbolic kernel their columns from a padded array of its own
  uint32_t slice_run_min = 64;   // chunks in a run of the slice kernel: at least this many (4..65536) ...
  uint32_t slice_run_div = 64;   // ... and one more per this many entries of the A row
  uint32_t slice_sorted = 3;     // the slice kernel takes its items 0: in row order; 1: in the order of their first column; 2: the dearest first; 3: by the cost's power of two, then by first column
  uint32_t slice_cost_w = 4;     // slice_sorted = 2: weight of an A entry of the row against a product, in quarters
  uint32_t slice_packed = 0;     // 1: the slice kernel reads B as 12-byte {column, value} records
  uint32_t slice_sampled = 1;    // 1: its searches inside rows of B go over a 1-in-32 sample of B's columns first
  uint32_t big_win_words = 32768; // bitmap window of the big-row symbolic kernel, in 32-column words
  uint32_t hash_num_slice = 1024; // LDS-hash numeric rows with more outputs go to the slice kernel
  uint32_t short_rpw = 16;       // most rows one wavefront of the short-row kernels takes in a row
  uint32_t short_num_rpw = 8;    // ... of the numeric one (its wavefronts take rows a launch-wide stride apart)
  uint32_t big_sym_bounds = 1;   // big-row symbolic kernel with precomputed window bounds of B's rows (0: cursor walk)
  uint32_t poison_output = 0;    // 1: the arrays of C are filled with 0xFF bytes before the numeric pass (tests: a skipped entry shows)
  uint32_t big_sym_sorted = 1;   // ... takes its rows the dearest first (0: in row order)
  uint32_t grp_runs = 1;         // window kernels walk runs of B rows with one column list as one (0: row by row)
  uint32_t rpw_fill = 16;        // ... but no more than leave this many wavefronts of work per wavefront slot of the device (0: no such limit)
  uint32_t grp_rpw = 8;          // most groups one wavefront of the run kernel takes one after the other (<= 64)
  uint32_t sym_rpw = 8;          // the same for the symbolic window kernel of a B with runs (1: one row per wavefront, the plain kernel)
  int cu_count = 256;
  void *h_pinned = nullptr; // small host staging buffer
  void *d_pinned = nullptr; // the same buffer as the device addresses it (k_report writes results into it)
  uint32_t *d_small = nullptr; // 4 KiB of device words for flags and small results
  unsigned char *noalloc_scratch = nullptr; // this device's copy of `scratch` (hipGetSymbolAddress)
  std::vector<Block> pool;  // freed output buffers, reused by later multiplies
  std::vector<Block> buffers; // live blocks handed out by spgemm_buffer_alloc
  std::unordered_set<spgemm_mat *> live_mats; // every matrix handle made on this context and not yet freed
  size_t pool_bytes = 0;
  spgemm_stats stats;
  // per-launch profile (option "profile_launches")
  int profile = 0; // 0 off, 1 per-launch events, 2 events + per-launch row sums
  std::vector<hipEvent_t> prof_events;
  struct LaunchRec {
    spgemm_launch_info info;
    const uint32_t *rows_bin;
    int ev0, ev1;
  };
  std::vector<LaunchRec> launches;
  uint64_t *r_nprod_prof = nullptr;
  const uint8_t *plan_gcnt = nullptr;
  // state carried from spgemm_symbolic to spgemm_numeric
  struct Plan {
    bool valid = false;
    const spgemm_mat *A = nullptr, *B = nullptr;
    uint32_t *r_cmin = nullptr, *r_cmax = nullptr, *r_nnz = nullptr;
    uint8_t *r_key = nullptr, *r_numkey = nullptr, *r_gcnt = nullptr;
    uint32_t *r_lead = nullptr;
    uint64_t *a_p0 = nullptr;  // short rows: B row start / length per A entry
    uint32_t *a_len = nullptr;
    uint64_t *a_pk = nullptr;  // short rows: start of the B row's padded records per A entry
    uint64_t *t_off = nullptr;
    uint32_t *perm = nullptr;
    uint32_t *blk_hist = nullptr;
    uint64_t *blk_off = nullptr, *partial = nullptr, *bin_off = nullptr;
    unsigned long long *zero_count = nullptr;
    uint32_t pblocks = 0;
    uint64_t nnz_structural = 0;
    // read back together with nnz_structural (one host round trip after the symbolic phase)
    bool num_ready = false;          // the three fields below are valid
    uint64_t num_boff[MAX_BINS + 1]; // start of every numeric bin in `perm`
    uint32_t max_len_big = 0;        // longest A row among the big rows
    uint32_t gmax_narrow = GRP_MAX, gmax_wide = GRP_MAX_WIDE; // largest row group among rows of up to GRP_ALEN entries / longer
    bool b_sorted = false; // every B row has strictly increasing columns
    uint32_t b_maxrun = 0; // longest run of rows of B with one column list (0 or 1: none)
  } plan;
  size_t arena_used = 0;
};

static constexpr size_t PINNED_BYTES = 8192;
static void release_mat(spgemm_ctx *ctx, spgemm_mat *m);
static constexpr size_t POOL_LIMIT_BLOCKS = 32;

// ---- scratch of the allocation-free entry point -----------------------------------------------------
// spgemm_assemble_row_ptr_noalloc keeps its row lengths and scan partials here instead of in the
// context's arena (which grows through hipMalloc).  One copy per GPU, held until the process ends;
// every call on a GPU shares it.  The array's name is only a host-side stand-in: host code uses the
// address hipGetSymbolAddress returned for the context's device (spgemm_ctx::noalloc_scratch).
static constexpr uint64_t NOALLOC_MAX_ROWS = 1ull << 22;
static constexpr size_t NOALLOC_LEN_BYTES = ((NOALLOC_MAX_ROWS + 1) * 4 + 255) / 256 * 256;
static constexpr size_t NOALLOC_PARTIAL_BYTES =
    (((NOALLOC_MAX_ROWS + (uint64_t)SCAN_BLOCK * SCAN_ITEMS - 1) / ((uint64_t)SCAN_BLOCK * SCAN_ITEMS) + 2) * 8 + 255) / 256 * 256;
static __device__ __align__(256) unsigned char scratch[NOALLOC_LEN_BYTES + NOALLOC_PARTIAL_BYTES];

// ---- per-device state shared by the contexts of a process -----------------------------------------------
// probe: result of k_probe_lane_order on the device (-1: not run yet); one launch and one host
// round trip per device and process instead of per context.
// scratch guard: every context of a GPU shares that GPU's copy of `scratch`.  The calls of one
// context are ordered by its stream; a call from ANOTHER stream first waits (on the device, through
// an event recorded behind the previous call) until that call's kernels are done with the array.
struct DeviceShared {
  std::mutex mu;
  int probe = -1;
  hipEvent_t scratch_done = nullptr; // recorded after the last user of `scratch` on this device
  hipStream_t scratch_stream = nullptr;
  bool scratch_used = false;
};
static constexpr int MAX_DEVICES = 64;
static DeviceShared g_dev[MAX_DEVICES];

// ---- helpers ------------------------------------------------------------------------------------
static bool walk_tail(const spgemm_ctx *ctx, const spgemm_mat *B) {
  return ctx->walk_tail == 2 ? B->nnz >= 48 * B->rows : ctx->walk_tail != 0;
}

static int set_device(const spgemm_ctx *ctx) {
  HIP_TRY(hipSetDevice(ctx->device));
  return SPGEMM_OK;
}

static int grow(spgemm_ctx *ctx, Growable &g, size_t bytes) {
  if (bytes <= g.cap) return SPGEMM_OK;
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  if (g.p) HIP_TRY(hipFree(g.p));
  g.p = nullptr;
  g.cap = 0;
  size_t want = bytes + bytes / 8 + 4096;
  HIP_TRY(hipMalloc(&g.p, want));
  g.cap = want;
  return SPGEMM_OK;
}

static int pool_alloc(spgemm_ctx *ctx, size_t bytes, void **out, size_t *got) {
  if (bytes == 0) bytes = 256;
  int best = -1;
  for (size_t i = 0; i < ctx->pool.size(); ++i)
    if (ctx->pool[i].bytes >= bytes && (best < 0 || ctx->pool[i].bytes < ctx->pool[best].bytes)) best = (int)i;
  if (best >= 0 && ctx->pool[best].bytes <= bytes + bytes / 2 + (1 << 20)) {
    *out = ctx->pool[best].p;
    *got = ctx->pool[best].bytes;
    ctx->pool_bytes -= ctx->pool[best].bytes;
    ctx->pool.erase(ctx->pool.begin() + best);
    return SPGEMM_OK;
  }
  hipError_t e = hipMalloc(out, bytes);
  if (e == hipErrorOutOfMemory && !ctx->pool.empty()) {
    for (auto &b : ctx->pool) (void)hipFree(b.p);
    ctx->pool.clear();
    ctx->pool_bytes = 0;
    e = hipMalloc(out, bytes);
  }
  if (e != hipSuccess)
    return fail(e == hipErrorOutOfMemory ? SPGEMM_ERR_NOMEM : SPGEMM_ERR_HIP, "hipMalloc(%zu) failed: %s", bytes,
                hipGetErrorString(e));
  *got = bytes;
  return SPGEMM_OK;
}

static void pool_release(spgemm_ctx *ctx, void *p, size_t bytes) {
  if (!p) return;
  if (ctx->pool.size() >= POOL_LIMIT_BLOCKS) {
    // drop the block that has waited longest (blocks are appended as they come back and taken from wherever
    // they fit): after a change of workload the blocks of the old one go, not the small ones of the new
    // one, which would then be allocated and freed in every multiply
    (void)hipFree(ctx->pool[0].p);
    ctx->pool_bytes -= ctx->pool[0].bytes;
    ctx->pool.erase(ctx->pool.begin());
  }
  ctx->pool.push_back({p, bytes});
  ctx->pool_bytes += bytes;
}

static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

struct ArenaCursor {
  char *base;
  size_t used;
  size_t cap;
  template <class T> T *take(size_t n) {
    size_t bytes = align_up(n * sizeof(T), 256);
    T *p = reinterpret_cast<T *>(base + used);
    used += bytes;
    return p;
  }
};

static uint32_t div_up_u64(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }

// exclusive scan of n uint32 -> out[n+1] uint64, using `partial` (nparts + 1 uint64)
static uint32_t scan_parts(uint64_t n) { return div_up_u64(n ? n : 1, (uint64_t)SCAN_BLOCK * SCAN_ITEMS); }

static int launch_scan(spgemm_ctx *ctx, const uint32_t *in, uint64_t n, uint64_t *out, uint64_t *partial) {
  if (n <= SCAN_SMALL_MAX && reinterpret_cast<const void *>(in) != reinterpret_cast<const void *>(out)) {
    hipLaunchKernelGGL(k_scan_small, dim3(1), dim3(1024), 0, ctx->stream, in, (uint32_t)n, out);
    HIP_TRY(hipGetLastError());
    return SPGEMM_OK;
  }
  uint32_t nparts = scan_parts(n);
  hipLaunchKernelGGL(k_scan_partials, dim3(nparts), dim3(SCAN_BLOCK), 0, ctx->stream, in, n, partial);
  const uint32_t own_spine = nparts <= SCAN_OWN_SPINE_MAX ? 1u : 0u;
  if (!own_spine) hipLaunchKernelGGL(k_scan_spine, dim3(1), dim3(1024), 0, ctx->stream, partial, nparts);
  hipLaunchKernelGGL(k_scan_final, dim3(nparts), dim3(SCAN_BLOCK), 0, ctx->stream, in, n, partial, out, nparts, own_spine);
  HIP_TRY(hipGetLastError());
  return SPGEMM_OK;
}

// stable partition of rows by key; bin_off (nbins + 1 uint64, device)
struct PartScratch {
  uint32_t nblocks;
  uint32_t *blk_hist;
  uint64_t *blk_off;
  uint64_t *partial;
};

static uint32_t part_blocks(uint64_t n) { return div_up_u64(n ? n : 1, (uint64_t)PART_BLOCK * PART_ROUNDS); }

static int launch_partition(spgemm_ctx *ctx, const uint8_t *key, uint32_t n, uint32_t nbins, PartScratch ps,
                            uint32_t *rows_out, uint64_t *bin_off) {
  uint32_t nb = ps.nblocks;
  hipLaunchKernelGGL(k_part_hist, dim3(nb), dim3(PART_BLOCK), 0, ctx->stream, key, n, nbins, nb, ps.blk_hist);
  TRY(launch_scan(ctx, ps.blk_hist, (uint64_t)nbins * nb, ps.blk_off, ps.partial));
  hipLaunchKernelGGL(k_part_scatter, dim3(nb), dim3(PART_BLOCK), 0, ctx->stream, key, n, nbins, nb, ps.blk_off,
                     rows_out, bin_off);
  HIP_TRY(hipGetLastError());
  return SPGEMM_OK;
}

// layout of d_rowinfo: 2 * rows + 1 records of 16 bytes (the last one holds counters), then the
// `same` bits (one per row, in 32-bit words)
static size_t rowinfo_same_words(const spgemm_mat *B) { return (size_t)((B->rows + 31) / 32) + 1; }
static RowInfo rowinfo_of(const spgemm_mat *B) {
  return RowInfo{B->d_rowinfo, B->rowinfo_sh, B->rowinfo_sh ? 1ull : B->rows,
                 reinterpret_cast<uint32_t *>(B->d_rowinfo + 2 * B->rows + 1)};
}

static int ensure_rowinfo(spgemm_ctx *ctx, const spgemm_mat *B) {
  if (B->d_rowinfo) return SPGEMM_OK;
  spgemm_mat *mb = const_cast<spgemm_mat *>(B);
  // short rows of A (which read both halves of a record) need short rows of B
  mb->rowinfo_sh = B->nnz <= 32 * B->rows ? 1u : 0u;
  TRY(pool_alloc(ctx, (2 * B->rows + 1) * sizeof(uint4) + rowinfo_same_words(B) * 4, (void **)&mb->d_rowinfo, &mb->bytes_rowinfo));
  // counters (unsorted rows, longest run of rows with one column list) and the `same` bits start at zero
  HIP_TRY(hipMemsetAsync(mb->d_rowinfo + 2 * B->rows, 0, sizeof(uint4) + rowinfo_same_words(B) * 4, ctx->stream));
  if (B->rows) {
    uint32_t blocks = (uint32_t)std::min<uint64_t>(div_up_u64(B->rows * 8, 256), 8192);
    hipLaunchKernelGGL(k_b_rowinfo, dim3(blocks), dim3(256), 0, ctx->stream, B->rows, B->d_ptr, B->d_col,
                       rowinfo_of(B));
    hipLaunchKernelGGL(k_b_maxrun, dim3(div_up_u64(div_up_u64(B->rows, 32), 256)), dim3(256), 0, ctx->stream, B->rows, rowinfo_of(B), 255u);
    HIP_TRY(hipGetLastError());
  }
  return SPGEMM_OK;
}

// Where the padded records of every row of B would start (the records themselves: ensure_prec),
// into the second half of the row-info records.  Only for a B that short rows of A may refer to.
static int ensure_prec_layout(spgemm_ctx *ctx, const spgemm_mat *B) {
  if (B->d_prow || !B->rows) return SPGEMM_OK;
  spgemm_mat *mb = const_cast<spgemm_mat *>(B);
  {
    const size_t b_start = align_up((B->rows + 1) * 8, 256), b_part = align_up(((size_t)scan_parts(B->rows) + 2) * 8, 256);
    TRY(pool_alloc(ctx, b_start + b_part + B->rows * 4, (void **)&mb->d_prow, &mb->bytes_prow));
    uint64_t *partial = reinterpret_cast<uint64_t *>(reinterpret_cast<char *>(mb->d_prow) + b_start);
    uint32_t *count = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(mb->d_prow) + b_start + b_part);
    hipLaunchKernelGGL(k_prec_count, dim3(div_up_u64(B->rows, 256)), dim3(256), 0, ctx->stream, B->rows, rowinfo_of(B), count);
    TRY(launch_scan(ctx, count, B->rows, mb->d_prow, partial));
    hipLaunchKernelGGL(k_prec_patch, dim3(div_up_u64(B->rows, 256)), dim3(256), 0, ctx->stream, B->rows, mb->d_prow,
                       rowinfo_of(B));
    HIP_TRY(hipGetLastError());
  }
  return SPGEMM_OK;
}

// B's rows as padded records for the short-row numeric kernel: built when that kernel first needs
// them, kept on the handle like the row info.  At most 7 records of padding per row.
static int ensure_prec(spgemm_ctx *ctx, const spgemm_mat *B) {
  if (B->d_prec || !B->nnz) return SPGEMM_OK;
  spgemm_mat *mb = const_cast<spgemm_mat *>(B);
  // 16 bytes per record slot, then the columns alone at 4 bytes per record slot
  // (a multiple of 8 slots, so that the column array starts on a 128-byte line as the records do)
  const uint64_t slots = (B->nnz + 7 * std::min<uint64_t>(B->rows, B->nnz) + 7) & ~(uint64_t)7;
  TRY(pool_alloc(ctx, slots * (sizeof(uint4) + 4), (void **)&mb->d_prec, &mb->bytes_prec));
  mb->prec_slots = slots;
  hipLaunchKernelGGL(k_prec_fill, dim3(div_up_u64(B->rows * 8, 256)), dim3(256), 0, ctx->stream, B->rows, B->d_ptr, B->d_col,
                     B->d_val, rowinfo_of(B), mb->d_prec, reinterpret_cast<uint32_t *>(mb->d_prec + slots));
  HIP_TRY(hipGetLastError());
  return SPGEMM_OK;
}

// The column sample of B (k_b_colsample) for the big-row numeric kernel: built when that kernel first
// needs it, kept on the handle like the row info (which it is released with).
static int ensure_colsample(spgemm_ctx *ctx, const spgemm_mat *B) {
  if (B->d_samp || !B->nnz) return SPGEMM_OK;
  spgemm_mat *mb = const_cast<spgemm_mat *>(B);
  const uint64_t nb = (B->nnz + 31) / 32;
  TRY(pool_alloc(ctx, nb * 4 + 16, (void **)&mb->d_samp, &mb->bytes_samp));
  hipLaunchKernelGGL(k_b_colsample, dim3((uint32_t)div_up_u64(nb, 256)), dim3(256), 0, ctx->stream, B->nnz, B->d_col, mb->d_samp);
  HIP_TRY(hipGetLastError());
  return SPGEMM_OK;
}

// B as 16-byte {column, value} records for the big-row numeric kernel: built when that kernel first
// needs them, kept on the handle like the row info.
static int ensure_packed(spgemm_ctx *ctx, const spgemm_mat *B) {
  if (B->d_packed || !B->nnz) return SPGEMM_OK;
  spgemm_mat *mb = const_cast<spgemm_mat *>(B);
  TRY(pool_alloc(ctx, B->nnz * sizeof(Rec12) + 16, (void **)&mb->d_packed, &mb->bytes_packed));
  const uint32_t blocks = (uint32_t)std::min<uint64_t>(div_up_u64(B->nnz, 256), (uint64_t)ctx->cu_count * 16);
  hipLaunchKernelGGL(k_pack_records, dim3(blocks), dim3(256), 0, ctx->stream, B->nnz, B->d_col, B->d_val, mb->d_packed);
  HIP_TRY(hipGetLastError());
  return SPGEMM_OK;
}

// Window bounds of B's rows for the big-row symbolic kernel (k_sym_big2): windows of 2^ws columns, as
// wide as the bitmap allows (win_words_max 32-column words) and at most BIG2_MAX_WIN of them; a wider
// B has none (*ok = false: the cursor-walking kernel takes its big rows).  Built when that kernel
// first needs them, kept on the handle like the row info.  B's rows must be sorted.
static int ensure_winbounds(spgemm_ctx *ctx, const spgemm_mat *B, uint32_t win_words_max, bool *ok) {
  uint32_t ws = 5;
  while ((1u << (ws - 5u)) < win_words_max && ws < 19u) ++ws; // the widest window the bitmap holds
  if ((1u << (ws - 5u)) > win_words_max) --ws;
  const uint64_t nwin = B->cols ? ((B->cols - 1) >> ws) + 1 : 1;
  *ok = ws >= 10u && nwin <= BIG2_MAX_WIN && B->rows > 0;
  if (!*ok) return SPGEMM_OK;
  spgemm_mat *mb = const_cast<spgemm_mat *>(B);
  if (B->d_wb && B->wb_ws == ws && B->wb_nwin == (uint32_t)nwin) return SPGEMM_OK;
  if (mb->d_wb) {
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    pool_release(ctx, mb->d_wb, mb->bytes_wb);
    mb->d_wb = nullptr;
  }
  const uint64_t n = B->rows * (nwin + 1);
  TRY(pool_alloc(ctx, n * 4, (void **)&mb->d_wb, &mb->bytes_wb));
  mb->wb_ws = ws;
  mb->wb_nwin = (uint32_t)nwin;
  for (uint64_t r0 = 0; r0 < B->rows; r0 += (1ull << 24)) { // 2^24 rows per launch keep the grid below 2^32 threads
    const uint64_t nr = std::min<uint64_t>(B->rows - r0, 1ull << 24);
    hipLaunchKernelGGL(k_b_winbounds, dim3(div_up_u64(nr * (nwin + 1), 256)), dim3(256), 0, ctx->stream, nr, B->d_ptr + r0, B->d_col,
                       (uint32_t)nwin, ws, mb->d_wb + r0 * (nwin + 1));
  }
  HIP_TRY(hipGetLastError());
  return SPGEMM_OK;
}

// Column map of the short-row numeric kernel (entries of B per column range, prefix summed): built
// when the first launch of that kernel needs it, kept on the handle like the row info.
static int ensure_colcdf(spgemm_ctx *ctx, const spgemm_mat *B) {
  if (B->d_colcdf) return SPGEMM_OK;
  spgemm_mat *mb = const_cast<spgemm_mat *>(B);
  uint32_t shift = 0;
  while (B->cols && ((B->cols - 1) >> shift) >= CDF_BINS) ++shift;
  mb->cdf_shift = shift;
  const size_t cdf_bytes = align_up((CDF_BINS + 2) * sizeof(float), 256); // + the flag of an even spread
  TRY(pool_alloc(ctx, cdf_bytes + CDF_BINS * sizeof(uint32_t), (void **)&mb->d_colcdf, &mb->bytes_colcdf));
  uint32_t *hist = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(mb->d_colcdf) + cdf_bytes);
  HIP_TRY(hipMemsetAsync(hist, 0, CDF_BINS * sizeof(uint32_t), ctx->stream));
  if (B->nnz) {
    const uint32_t blocks = (uint32_t)std::min<uint64_t>(div_up_u64(B->nnz, 256 * 16), 2048);
    hipLaunchKernelGGL(k_col_hist, dim3(blocks), dim3(256), 0, ctx->stream, B->nnz, B->d_col, shift, hist);
  }
  hipLaunchKernelGGL(k_col_cdf, dim3(1), dim3(1024), 0, ctx->stream, hist, mb->d_colcdf,
                     B->cols ? (uint32_t)((B->cols - 1) >> shift) + 1u : 0u);
  HIP_TRY(hipGetLastError());
  return SPGEMM_OK;
}

// Wavefronts per workgroup (1..4) for a kernel that needs `region` bytes of LDS per wavefront: the
// count that leaves the least of a CU's 160 KiB unused (at most 32 wavefronts fit anyway); the
// larger workgroup on a tie.
static uint32_t pick_wpb(size_t region) {
  uint32_t best_waves = 0, wpb = 1;
  for (uint32_t w = 1; w <= 4; ++w) {
    const uint32_t waves = std::min<uint32_t>(32, (uint32_t)((160 * 1024) / (w * region)) * w);
    if (waves >= best_waves) { best_waves = waves; wpb = w; }
  }
  return wpb;
}

// Rows a wavefront of the short-row kernels takes one after the other (it pipelines their loads):
// as many as leave about 64 wavefronts of work per CU (rpw_fill = 0: no such limit, so that small
// inputs reach the pipelined form too), at most short_rpw.
static uint32_t short_rows_per_wave(const spgemm_ctx *ctx, uint32_t rows, bool numeric = false) {
  const uint64_t waves = (uint64_t)ctx->cu_count * 64;
  const uint64_t most = numeric ? ctx->short_num_rpw : ctx->short_rpw;
  return (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(most, ctx->rpw_fill ? rows / waves : rows));
}

// ---- LDS limits ------------------------------------------------------------------------------------
template <class K> static void raise_lds_limit(K kernel) {
  hipFuncAttributes fa;
  size_t fixed = 0; // LDS the kernel declares statically counts against the same 160 KiB
  if (hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(kernel)) == hipSuccess) fixed = fa.sharedSizeBytes;
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)(160 * 1024 - fixed));
  (void)hipGetLastError(); // a refusal shows again, with the kernel's name, at the launch that needs the room
}
static void raise_all_lds_limits() { // per device: called with the context's device current
  raise_lds_limit(k_sym_window<true>); raise_lds_limit(k_sym_window<false>); raise_lds_limit(k_sym_window<false, true>);
  raise_lds_limit(k_sym_window_runs<false>); raise_lds_limit(k_sym_window_runs<true>);
  raise_lds_limit(k_num_window_grp<0, false>); raise_lds_limit(k_num_window_grp<1, false>); raise_lds_limit(k_num_window_grp<2, false>);
  raise_lds_limit(k_num_window_grp<0, true>); raise_lds_limit(k_num_window_grp<1, true>); raise_lds_limit(k_num_window_grp<2, true>);
  raise_lds_limit(k_num_window_grp<0, false, true>); raise_lds_limit(k_num_window_grp<1, false, true>); raise_lds_limit(k_num_window_grp<2, false, true>);
  raise_lds_limit(k_num_window_grp<0, true, true>); raise_lds_limit(k_num_window_grp<1, true, true>); raise_lds_limit(k_num_window_grp<2, true, true>);
  raise_lds_limit(k_sym_hash<true>); raise_lds_limit(k_sym_hash<false>);
  raise_lds_limit(k_num_window<true>); raise_lds_limit(k_num_window<false>);
  raise_lds_limit(k_num_hash<true>); raise_lds_limit(k_num_hash<false>);
  raise_lds_limit(k_sym_big); raise_lds_limit(k_sym_big2);
  raise_lds_limit(k_sym_hash_wg<256>); raise_lds_limit(k_sym_hash_wg<512>); raise_lds_limit(k_sym_hash_wg<1024>);
  raise_lds_limit(k_sym_count<4>); raise_lds_limit(k_sym_count<SH_STEPS>);
  raise_lds_limit(k_num_short<4, false>); raise_lds_limit(k_num_short<SH_STEPS, false>);
  raise_lds_limit(k_num_short<4, true>); raise_lds_limit(k_num_short<SH_STEPS, true>);
  raise_lds_limit(k_probe_lane_order);
#define SG_RAISE_RUN(W, C, G, T) raise_lds_limit(k_num_grp_run<W, C, 3, G, T>); raise_lds_limit(k_num_grp_run<W, C, 5, G, T>); raise_lds_limit(k_num_grp_run<W, C, 8, G, T>);
  SG_RAISE_RUN(false, false, 3, 2) SG_RAISE_RUN(false, false, 5, 2) SG_RAISE_RUN(false, false, 8, 2)
  SG_RAISE_RUN(false, true, 3, 2) SG_RAISE_RUN(false, true, 5, 2) SG_RAISE_RUN(false, true, 8, 2)
  SG_RAISE_RUN(false, false, 3, 8) SG_RAISE_RUN(false, false, 5, 8) SG_RAISE_RUN(false, false, 8, 8)
  SG_RAISE_RUN(false, true, 3, 8) SG_RAISE_RUN(false, true, 5, 8) SG_RAISE_RUN(false, true, 8, 8)
  SG_RAISE_RUN(true, false, 3, 8) SG_RAISE_RUN(true, false, 4, 8) SG_RAISE_RUN(true, true, 3, 8) SG_RAISE_RUN(true, true, 4, 8)
#undef SG_RAISE_RUN
}

// ---- library ------------------------------------------------------------------------------------
extern "C" int spgemm_version(void) { return 100; }
extern "C" const char *spgemm_last_error(void) { return g_err.c_str(); }

extern "C" int spgemm_device_count(int *count) {
  if (!count) return fail(SPGEMM_ERR_INVALID, "count is NULL");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    *count = 0;
    return fail(SPGEMM_ERR_HIP, "hipGetDeviceCount failed: %s", hipGetErrorString(e));
  }
  *count = n;
  return SPGEMM_OK;
}

extern "C" int spgemm_create(int device, spgemm_ctx **out) {
  if (!out) return fail(SPGEMM_ERR_INVALID, "ctx out pointer is NULL");
  *out = nullptr;
  int n = 0;
  TRY(spgemm_device_count(&n));
  if (n <= 0) return fail(SPGEMM_ERR_HIP, "no HIP device is visible; this library has no CPU path");
  if (device < 0 || device >= n || device >= MAX_DEVICES) return fail(SPGEMM_ERR_INVALID, "device %d out of range (0..%d)", device, std::min(n, MAX_DEVICES) - 1);
  HIP_TRY(hipSetDevice(device));
  spgemm_ctx *ctx = new (std::nothrow) spgemm_ctx();
  if (!ctx) return fail(SPGEMM_ERR_NOMEM, "host allocation failed");
  ctx->device = device;
  ctx->cp.win_max_span = 65536;
  ctx->cp.win_ratio = 128;
  ctx->cp.hash_max_nprod = 8192;
  ctx->cp.g_thresh0 = 6;
  ctx->cp.g_thresh1 = 12;
  ctx->cp.g_thresh2 = 24;
  ctx->cp.short_max_nprod = SH_MAX_NPROD;
  memset(&ctx->stats, 0, sizeof(ctx->stats));
  // a failure below releases whatever was created so far (spgemm_destroy copes with a partial context)
  const char *what = "hipStreamCreate";
  hipError_t e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking);
  ctx->stream = ctx->own_stream;
  for (auto &ev : ctx->ev)
    if (e == hipSuccess) { what = "hipEventCreate"; e = hipEventCreate(&ev); }
  if (e == hipSuccess) { what = "hipHostMalloc"; e = hipHostMalloc(&ctx->h_pinned, PINNED_BYTES); }
  if (e == hipSuccess) { what = "hipHostGetDevicePointer"; e = hipHostGetDevicePointer(&ctx->d_pinned, ctx->h_pinned, 0); }
  if (e == hipSuccess) { what = "hipMalloc"; e = hipMalloc((void **)&ctx->d_small, 4096); }
  // address of this device's copy of the allocation-free entry point's scratch array
  if (e == hipSuccess) { what = "hipGetSymbolAddress"; e = hipGetSymbolAddress((void **)&ctx->noalloc_scratch, HIP_SYMBOL(scratch)); }
  if (e != hipSuccess) {
    (void)spgemm_destroy(ctx);
    return fail(e == hipErrorOutOfMemory ? SPGEMM_ERR_NOMEM : SPGEMM_ERR_HIP, "%s failed: %s", what, hipGetErrorString(e));
  }
  raise_all_lds_limits();
  {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0)
      ctx->cu_count = cus;
  }
  // does one LDS fp64 add apply lanes that share an address in lane order?  (see k_probe_lane_order)
  // Asked once per device and process; the event that guards the allocation-free entry point's
  // scratch array is made at the same time.
  {
    DeviceShared &ds = g_dev[device];
    std::lock_guard<std::mutex> lock(ds.mu);
    if (!ds.scratch_done) {
      e = hipEventCreateWithFlags(&ds.scratch_done, hipEventDisableTiming);
      if (e != hipSuccess) {
        ds.scratch_done = nullptr;
        (void)spgemm_destroy(ctx);
        return fail(SPGEMM_ERR_HIP, "hipEventCreate failed: %s", hipGetErrorString(e));
      }
    }
    if (ds.probe < 0) {
      unsigned long long *d_bad = reinterpret_cast<unsigned long long *>(ctx->d_small);
      unsigned long long bad = 1;
      e = hipMemsetAsync(d_bad, 0, 8, ctx->stream);
      if (e == hipSuccess) {
        hipLaunchKernelGGL(k_probe_lane_order, dim3(PROBE_BLOCKS), dim3(256), PROBE_LDS_BYTES, ctx->stream, PROBE_TRIALS, d_bad);
        e = hipGetLastError();
      }
      if (e == hipSuccess) e = hipMemcpyAsync(&bad, d_bad, 8, hipMemcpyDeviceToHost, ctx->stream);
      if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
      if (e != hipSuccess) {
        (void)spgemm_destroy(ctx);
        return fail(SPGEMM_ERR_HIP, "lane order probe failed: %s", hipGetErrorString(e));
      }
      ds.probe = bad == 0 ? 1 : 0;
    }
    ctx->hw_lane_order = ds.probe == 1;
  }
  *out = ctx;
  return SPGEMM_OK;
}

extern "C" int spgemm_destroy(spgemm_ctx *ctx) {
  if (!ctx) return SPGEMM_OK;
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  // matrices never freed by the caller: their device arrays go back to the pool, which is emptied below
  for (spgemm_mat *m : std::vector<spgemm_mat *>(ctx->live_mats.begin(), ctx->live_mats.end())) release_mat(ctx, m);
  ctx->live_mats.clear();
  for (auto &b : ctx->pool) (void)hipFree(b.p);
  for (auto &b : ctx->buffers) (void)hipFree(b.p);
  if (ctx->arena.p) (void)hipFree(ctx->arena.p);
  if (ctx->temp_cols.p) (void)hipFree(ctx->temp_cols.p);
  if (ctx->gscratch.p) (void)hipFree(ctx->gscratch.p);
  if (ctx->big_cur.p) (void)hipFree(ctx->big_cur.p);
  if (ctx->slice_buf.p) (void)hipFree(ctx->slice_buf.p);
  if (ctx->h_pinned) (void)hipHostFree(ctx->h_pinned);
  if (ctx->d_small) (void)hipFree(ctx->d_small);
  for (auto &ev : ctx->ev)
    if (ev) (void)hipEventDestroy(ev);
  for (auto &ev : ctx->prof_events) (void)hipEventDestroy(ev);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
  delete ctx;
  return SPGEMM_OK;
}

extern "C" int spgemm_ctx_device(const spgemm_ctx *ctx, int *device) {
  if (!ctx || !device) return fail(SPGEMM_ERR_INVALID, "NULL argument");
  *device = ctx->device;
  return SPGEMM_OK;
}

extern "C" int spgemm_set_stream(spgemm_ctx *ctx, void *hip_stream) {
  if (!ctx) return fail(SPGEMM_ERR_INVALID, "ctx is NULL");
  TRY(set_device(ctx));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  ctx->stream = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : ctx->own_stream;
  ctx->plan.valid = false;
  return SPGEMM_OK;
}

extern "C" int spgemm_internal_stream(spgemm_ctx *ctx, hipStream_t *stream) {
  if (!ctx || !stream) return fail(SPGEMM_ERR_INVALID, "NULL argument");
  TRY(set_device(ctx));
  *stream = ctx->stream;
  return SPGEMM_OK;
}

extern "C" int spgemm_internal_fail(int code, const char *message) { return fail(code, "%s", message); }

extern "C" int spgemm_internal_alloc_mat(spgemm_ctx *ctx, uint64_t rows, uint64_t cols, uint64_t nnz, spgemm_mat **out) {
  if (!ctx || !out) return fail(SPGEMM_ERR_INVALID, "NULL argument");
  *out = nullptr;
  TRY(set_device(ctx));
  spgemm_mat *m = new (std::nothrow) spgemm_mat();
  if (!m) return fail(SPGEMM_ERR_NOMEM, "host allocation failed");
  m->rows = rows; m->cols = cols; m->nnz = nnz; m->owns = true; m->device = ctx->device;
  int rc = pool_alloc(ctx, (rows + 1) * 8, (void **)&m->d_ptr, &m->bytes_ptr);
  if (rc == SPGEMM_OK) rc = pool_alloc(ctx, nnz * 4, (void **)&m->d_col, &m->bytes_col);
  if (rc == SPGEMM_OK) rc = pool_alloc(ctx, nnz * 8, (void **)&m->d_val, &m->bytes_val);
  if (rc != SPGEMM_OK) {
    pool_release(ctx, m->d_ptr, m->bytes_ptr);
    pool_release(ctx, m->d_col, m->bytes_col);
    pool_release(ctx, m->d_val, m->bytes_val);
    delete m;
    return rc;
  }
  ctx->live_mats.insert(m);
  *out = m;
  return SPGEMM_OK;
}

extern "C" int spgemm_internal_pool_alloc(spgemm_ctx *ctx, size_t bytes, void **p, size_t *got) {
  return pool_alloc(ctx, bytes, p, got);
}
extern "C" void spgemm_internal_pool_release(spgemm_ctx *ctx, void *p, size_t got) { pool_release(ctx, p, got); }

// Measured rate of a streaming copy of `bytes` bytes (read + write counted, GB/s, best of `reps`):
// the achievable HBM rate the roofline figures are put next to.  Used by the benchmark only.
extern "C" int spgemm_internal_stream_copy_rate(spgemm_ctx *ctx, size_t bytes, int reps, double *gbs) {
  if (!ctx || !gbs || bytes < 4096 || reps < 1) return fail(SPGEMM_ERR_INVALID, "bad argument");
  TRY(set_device(ctx));
  void *src = nullptr, *dst = nullptr;
  size_t gs = 0, gd = 0;
  TRY(pool_alloc(ctx, bytes, &src, &gs));
  int rc = pool_alloc(ctx, bytes, &dst, &gd);
  if (rc != SPGEMM_OK) { pool_release(ctx, src, gs); return rc; }
  hipEvent_t e0 = nullptr, e1 = nullptr;
  hipError_t e = hipEventCreate(&e0);
  if (e == hipSuccess) e = hipEventCreate(&e1);
  if (e == hipSuccess) e = hipMemsetAsync(src, 1, bytes, ctx->stream);
  if (e == hipSuccess) e = hipMemsetAsync(dst, 0, bytes, ctx->stream);
  float best = 0.f;
  const uint64_t n16 = bytes / 16;
  const uint32_t blocks = (uint32_t)std::min<uint64_t>(div_up_u64(n16, 256 * COPY_UNROLL), (uint64_t)ctx->cu_count * 32);
  for (int r = 0; r < reps + 1 && e == hipSuccess; ++r) { // first pass warms up
    e = hipEventRecord(e0, ctx->stream);
    hipLaunchKernelGGL(k_stream_copy, dim3(blocks), dim3(256), 0, ctx->stream, (const uint4 *)src, (uint4 *)dst, n16);
    if (e == hipSuccess) e = hipEventRecord(e1, ctx->stream);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    if (r > 0 && ms > 0.f && (best == 0.f || ms < best)) best = ms;
  }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  (void)hipStreamSynchronize(ctx->stream);
  pool_release(ctx, src, gs);
  pool_release(ctx, dst, gd);
  if (e != hipSuccess) return fail(SPGEMM_ERR_HIP, "stream copy failed: %s", hipGetErrorString(e));
  *gbs = best > 0.f ? 2.0 * (double)(n16 * 16) / ((double)best * 1e-3) / 1e9 : 0.0;
  return SPGEMM_OK;
}

extern "C" int spgemm_synchronize(spgemm_ctx *ctx) {
  if (!ctx) return fail(SPGEMM_ERR_INVALID, "ctx is NULL");
  TRY(set_device(ctx));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return SPGEMM_OK;
}

extern "C" int spgemm_device_property(const spgemm_ctx *ctx, const char *name, int64_t *value) {
  if (!ctx || !name || !value) return fail(SPGEMM_ERR_INVALID, "NULL argument");
  const std::string n(name);
  if (n == "lane_order_probe") *value = ctx->hw_lane_order ? 1 : 0;
  else if (n == "cu_count") *value = ctx->cu_count;
  else return fail(SPGEMM_ERR_INVALID, "unknown device property %s", name);
  return SPGEMM_OK;
}

extern "C" int spgemm_set_option(spgemm_ctx *ctx, const char *name, int64_t value) {
  if (!ctx || !name) return fail(SPGEMM_ERR_INVALID, "ctx or name is NULL");
  std::string n(name);
  if (value < 0) return fail(SPGEMM_ERR_INVALID, "option %s: negative value", name);
  if (n == "window_max_span") {
    if (value > 65536) return fail(SPGEMM_ERR_INVALID, "window_max_span is at most 65536");
    ctx->cp.win_max_span = (uint32_t)value;
  } else if (n == "window_ratio") {
    ctx->cp.win_ratio = (uint32_t)value;
  } else if (n == "hash_max_nprod") {
    if (value > 8192) return fail(SPGEMM_ERR_INVALID, "hash_max_nprod is at most 8192");
    ctx->cp.hash_max_nprod = (uint32_t)value;
  } else if (n == "g_thresh0") {
    ctx->cp.g_thresh0 = (uint32_t)value;
  } else if (n == "g_thresh1") {
    ctx->cp.g_thresh1 = (uint32_t)value;
  } else if (n == "g_thresh2") {
    ctx->cp.g_thresh2 = (uint32_t)value;
  } else if (n == "short_max_nprod") {
    if (value > SH_MAX_NPROD) return fail(SPGEMM_ERR_INVALID, "short_max_nprod is at most %u", SH_MAX_NPROD);
    ctx->cp.short_max_nprod = (uint32_t)value;
  } else if (n == "short_rpw") {
    if (value < 1 || value > 4096) return fail(SPGEMM_ERR_INVALID, "short_rpw must be 1..4096");
    ctx->short_rpw = (uint32_t)value;
  } else if (n == "short_num_rpw") {
    if (value < 1 || value > 4096) return fail(SPGEMM_ERR_INVALID, "short_num_rpw must be 1..4096");
    ctx->short_num_rpw = (uint32_t)value;
  } else if (n == "flat_walk") {
    ctx->flat_walk = value != 0;
  } else if (n == "row_groups") {
    ctx->row_groups = value != 0;
  } else if (n == "profile_launches") {
    if (value > 2) return fail(SPGEMM_ERR_INVALID, "profile_launches is 0, 1 or 2");
    ctx->profile = (int)value;
  } else if (n == "global_blocks_sym") {
    if (value < 1 || value > 4096) return fail(SPGEMM_ERR_INVALID, "global_blocks_sym must be 1..4096");
    ctx->global_blocks_sym = (uint32_t)value;
  } else if (n == "big_rows") {
    ctx->big_rows = value != 0;
  } else if (n == "big_win_words") {
    if (value < 1024 || value > 32768 || value % 1024) // the kernel's own limit (BIG_WIN_MAX) caps it at launch
      return fail(SPGEMM_ERR_INVALID, "big_win_words must be a multiple of 1024 in 1024..32768");
    ctx->big_win_words = (uint32_t)value;
  } else if (n == "hash_num_slice") {
    if (value < 64 || value > 4096) return fail(SPGEMM_ERR_INVALID, "hash_num_slice must be 64..4096");
    ctx->hash_num_slice = (uint32_t)value;
  } else if (n == "lane_order_adds") {
    if (value < 0 || value > 1) return fail(SPGEMM_ERR_INVALID, "lane_order_adds must be 0 or 1");
    ctx->lane_order_adds = (uint32_t)value;
  } else if (n == "slice_queues") {
    if (value < 1 || value > (int64_t)SLICE_MAX_QUEUES) return fail(SPGEMM_ERR_INVALID, "slice_queues must be 1..8");
    ctx->slice_queues = (uint32_t)value;
  } else if (n == "grp_compact") {
    if (value < 0 || value > 2) return fail(SPGEMM_ERR_INVALID, "grp_compact must be 0, 1 or 2");
    ctx->grp_compact = (uint32_t)value;
  } else if (n == "short_sym_spread") {
    if (value < 0 || value > 2) return fail(SPGEMM_ERR_INVALID, "short_sym_spread must be 0..2");
    ctx->short_sym_spread = (uint32_t)value;
  } else if (n == "analyze_lanes") {
    if (value != 0 && value != 8 && value != 16 && value != 32) return fail(SPGEMM_ERR_INVALID, "analyze_lanes must be 0, 8, 16 or 32");
    ctx->analyze_lanes = (uint32_t)value;
  } else if (n == "short_key32") {
    if (value < 0 || value > 1) return fail(SPGEMM_ERR_INVALID, "short_key32 must be 0 or 1");
    ctx->short_key32 = (uint32_t)value;
  } else if (n == "walk_tail") {
    if (value < 0 || value > 2) return fail(SPGEMM_ERR_INVALID, "walk_tail must be 0, 1 or 2");
    ctx->walk_tail = (uint32_t)value;
  } else if (n == "poison_output") {
    ctx->poison_output = value != 0;
  } else if (n == "big_sym_sorted") {
    if (value > 1) return fail(SPGEMM_ERR_INVALID, "big_sym_sorted must be 0 or 1");
    ctx->big_sym_sorted = (uint32_t)value;
  } else if (n == "big_sym_bounds") {
    if (value > 1) return fail(SPGEMM_ERR_INVALID, "big_sym_bounds must be 0 or 1");
    ctx->big_sym_bounds = (uint32_t)value;
  } else if (n == "sym_rpw") {
    if (value < 1 || value > 64) return fail(SPGEMM_ERR_INVALID, "sym_rpw must be 1..64");
    ctx->sym_rpw = (uint32_t)value;
  } else if (n == "rpw_fill") {
    if (value < 0 || value > 1024) return fail(SPGEMM_ERR_INVALID, "rpw_fill must be 0..1024");
    ctx->rpw_fill = (uint32_t)value;
  } else if (n == "grp_rpw") {
    if (value < 1 || value > 64) return fail(SPGEMM_ERR_INVALID, "grp_rpw must be 1..64");
    ctx->grp_rpw = (uint32_t)value;
  } else if (n == "grp_runs") {
    if (value > 1) return fail(SPGEMM_ERR_INVALID, "grp_runs must be 0 or 1");
    ctx->grp_runs = (uint32_t)value;
  } else if (n == "short_packed") {
    if (value < 0 || value > 2) return fail(SPGEMM_ERR_INVALID, "short_packed must be 0, 1 or 2");
    ctx->short_packed = (uint32_t)value;
  } else if (n == "slice_run_min") {
    if (value < 4 || value > 65536) return fail(SPGEMM_ERR_INVALID, "slice_run_min must be 4..65536");
    ctx->slice_run_min = (uint32_t)value;
  } else if (n == "slice_run_div") {
    if (value < 1 || value > 65536) return fail(SPGEMM_ERR_INVALID, "slice_run_div must be 1..65536");
    ctx->slice_run_div = (uint32_t)value;
  } else if (n == "slice_sorted") {
    if (value < 0 || value > 3) return fail(SPGEMM_ERR_INVALID, "slice_sorted must be 0..3");
    ctx->slice_sorted = (uint32_t)value;
  } else if (n == "slice_cost_w") {
    if (value < 1 || value > 1024) return fail(SPGEMM_ERR_INVALID, "slice_cost_w must be 1..1024");
    ctx->slice_cost_w = (uint32_t)value;
  } else if (n == "slice_sampled") {
    if (value < 0 || value > 1) return fail(SPGEMM_ERR_INVALID, "slice_sampled must be 0 or 1");
    ctx->slice_sampled = (uint32_t)value;
  } else if (n == "slice_packed") {
    if (value < 0 || value > 1) return fail(SPGEMM_ERR_INVALID, "slice_packed must be 0 or 1");
    ctx->slice_packed = (uint32_t)value;
  } else if (n == "slice_blocks") {
    if (value < 0 || value > 65536) return fail(SPGEMM_ERR_INVALID, "slice_blocks must be 0..65536");
    ctx->slice_blocks = (uint32_t)value;
  } else if (n == "global_blocks_num") {
    if (value < 1 || value > 16384) return fail(SPGEMM_ERR_INVALID, "global_blocks_num must be 1..16384");
    ctx->global_blocks_num = (uint32_t)value;
  } else {
    return fail(SPGEMM_ERR_INVALID, "unknown option %s", name);
  }
  ctx->plan.valid = false;
  return SPGEMM_OK;
}

// ---- matrices -----------------------------------------------------------------------------------
static int validate_device(spgemm_ctx *ctx, const spgemm_mat *m) {
  uint32_t *d_bad = ctx->d_small;
  HIP_TRY(hipMemsetAsync(d_bad, 0, 4, ctx->stream));
  uint64_t work = std::max<uint64_t>(m->rows, m->nnz);
  uint32_t blocks = (uint32_t)std::min<uint64_t>(div_up_u64(work ? work : 1, 256), 4096);
  hipLaunchKernelGGL(k_validate, dim3(blocks), dim3(256), 0, ctx->stream, m->rows, m->cols, m->nnz, m->d_ptr,
                     m->d_col, d_bad);
  HIP_TRY(hipGetLastError());
  uint32_t bad = 0;
  HIP_TRY(hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  if (bad & 3u) return fail(SPGEMM_ERR_INVALID, "row_ptr is not a monotone 0..nnz sequence");
  if (bad & 4u) return fail(SPGEMM_ERR_INVALID, "col_idx holds an index >= cols");
  return SPGEMM_OK;
}

static int check_dims(uint64_t rows, uint64_t cols) {
  // rows: the one-thread-per-row kernels round the grid up to whole workgroups of 256 and keep the
  // thread id in 32 bits
  if (rows >= 0xFFFFFF00ull || cols >= 0xFFFFFFFFull)
    return fail(SPGEMM_ERR_UNSUPPORTED, "rows must be below 2^32 - 256 and cols below 2^32 - 1 (row/column ids are 32 bit)");
  return SPGEMM_OK;
}

extern "C" int spgemm_mat_from_host(spgemm_ctx *ctx, uint64_t rows, uint64_t cols, uint64_t nnz,
                                    const uint64_t *row_ptr, const uint32_t *col_idx, const double *val,
                                    spgemm_mat **out) {
  if (!ctx || !out || !row_ptr || (nnz && (!col_idx || !val))) return fail(SPGEMM_ERR_INVALID, "NULL argument");
  *out = nullptr;
  TRY(check_dims(rows, cols));
  if (row_ptr[0] != 0 || row_ptr[rows] != nnz) return fail(SPGEMM_ERR_INVALID, "row_ptr[0] != 0 or row_ptr[rows] != nnz");
  for (uint64_t r = 0; r < rows; ++r)
    if (row_ptr[r] > row_ptr[r + 1]) return fail(SPGEMM_ERR_INVALID, "row_ptr decreases at row %llu", (unsigned long long)r);
  for (uint64_t j = 0; j < nnz; ++j)
    if (col_idx[j] >= cols) return fail(SPGEMM_ERR_INVALID, "col_idx[%llu] = %u >= cols", (unsigned long long)j, col_idx[j]);
  TRY(set_device(ctx));
  spgemm_mat *m = new (std::nothrow) spgemm_mat();
  if (!m) return fail(SPGEMM_ERR_NOMEM, "host allocation failed");
  m->rows = rows; m->cols = cols; m->nnz = nnz; m->owns = true; m->device = ctx->device;
  ctx->live_mats.insert(m);
  int rc = pool_alloc(ctx, (rows + 1) * 8, (void **)&m->d_ptr, &m->bytes_ptr);
  if (rc == SPGEMM_OK) rc = pool_alloc(ctx, nnz * 4, (void **)&m->d_col, &m->bytes_col);
  if (rc == SPGEMM_OK) rc = pool_alloc(ctx, nnz * 8, (void **)&m->d_val, &m->bytes_val);
  if (rc != SPGEMM_OK) { spgemm_mat_free(ctx, m); return rc; }
  hipError_t e = hipMemcpyAsync(m->d_ptr, row_ptr, (rows + 1) * 8, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess && nnz) e = hipMemcpyAsync(m->d_col, col_idx, nnz * 4, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess && nnz) e = hipMemcpyAsync(m->d_val, val, nnz * 8, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) {
    spgemm_mat_free(ctx, m);
    return fail(SPGEMM_ERR_HIP, "host to device copy failed: %s", hipGetErrorString(e));
  }
  *out = m;
  return SPGEMM_OK;
}

extern "C" int spgemm_mat_wrap_device(spgemm_ctx *ctx, uint64_t rows, uint64_t cols, uint64_t nnz,
                                      const uint64_t *d_row_ptr, const uint32_t *d_col_idx, const double *d_val,
                                      int validate, spgemm_mat **out) {
  if (!ctx || !out || !d_row_ptr || (nnz && (!d_col_idx || !d_val))) return fail(SPGEMM_ERR_INVALID, "NULL argument");
  *out = nullptr;
  TRY(check_dims(rows, cols));
  TRY(set_device(ctx));
  spgemm_mat *m = new (std::nothrow) spgemm_mat();
  if (!m) return fail(SPGEMM_ERR_NOMEM, "host allocation failed");
  m->rows = rows; m->cols = cols; m->nnz = nnz; m->owns = false; m->device = ctx->device;
  m->d_ptr = const_cast<uint64_t *>(d_row_ptr);
  m->d_col = const_cast<uint32_t *>(d_col_idx);
  m->d_val = const_cast<double *>(d_val);
  if (validate) {
    int rc = validate_device(ctx, m);
    if (rc != SPGEMM_OK) { delete m; return rc; }
  }
  ctx->live_mats.insert(m);
  *out = m;
  return SPGEMM_OK;
}

extern "C" int spgemm_mat_info(const spgemm_mat *m, uint64_t *rows, uint64_t *cols, uint64_t *nnz) {
  if (!m) return fail(SPGEMM_ERR_INVALID, "matrix is NULL");
  if (rows) *rows = m->rows;
  if (cols) *cols = m->cols;
  if (nnz) *nnz = m->nnz;
  return SPGEMM_OK;
}

extern "C" int spgemm_mat_device_ptrs(const spgemm_mat *m, const uint64_t **d_row_ptr, const uint32_t **d_col_idx,
                                      const double **d_val) {
  if (!m) return fail(SPGEMM_ERR_INVALID, "matrix is NULL");
  if (d_row_ptr) *d_row_ptr = m->d_ptr;
  if (d_col_idx) *d_col_idx = m->d_col;
  if (d_val) *d_val = m->d_val;
  return SPGEMM_OK;
}

extern "C" int spgemm_mat_to_host(spgemm_ctx *ctx, const spgemm_mat *m, uint64_t *row_ptr, uint32_t *col_idx,
                                  double *val) {
  if (!ctx || !m) return fail(SPGEMM_ERR_INVALID, "NULL argument");
  TRY(set_device(ctx));
  if (row_ptr) HIP_TRY(hipMemcpyAsync(row_ptr, m->d_ptr, (m->rows + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
  if (col_idx && m->nnz) HIP_TRY(hipMemcpyAsync(col_idx, m->d_col, m->nnz * 4, hipMemcpyDeviceToHost, ctx->stream));
  if (val && m->nnz) HIP_TRY(hipMemcpyAsync(val, m->d_val, m->nnz * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return SPGEMM_OK;
}

// Invariants of a result matrix, checked on the device (a result of 10^10 entries is not brought to
// the host to look at it): row pointer monotone and ending at nnz, columns in range, every row
// strictly increasing (sorted, no duplicates), no stored zero.
extern "C" int spgemm_mat_check(spgemm_ctx *ctx, const spgemm_mat *m, uint64_t *rows_not_increasing,
                                uint64_t *zero_values) {
  if (!ctx || !m || !rows_not_increasing || !zero_values) return fail(SPGEMM_ERR_INVALID, "NULL argument");
  TRY(set_device(ctx));
  TRY(validate_device(ctx, m));
  unsigned long long *d = reinterpret_cast<unsigned long long *>(ctx->d_small) + 16;
  HIP_TRY(hipMemsetAsync(d, 0, 16, ctx->stream));
  if (m->rows) {
    for (uint64_t r0 = 0; r0 < m->rows; r0 += (1ull << 26)) { // 2^26 rows per launch keep the grid below 2^32 threads
      const uint64_t nr = std::min<uint64_t>(m->rows - r0, 1ull << 26);
      hipLaunchKernelGGL(k_check_rows, dim3(div_up_u64(nr * 16, 256)), dim3(256), 0, ctx->stream, nr, m->d_ptr + r0,
                         m->d_col, m->d_val, d);
    }
    HIP_TRY(hipGetLastError());
  }
  unsigned long long *h = reinterpret_cast<unsigned long long *>(ctx->h_pinned);
  HIP_TRY(hipMemcpyAsync(h, d, 16, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  *rows_not_increasing = h[0];
  *zero_values = h[1];
  return SPGEMM_OK;
}

// Rows [row0, row0 + nrows) of a device matrix to host arrays: row_ptr (nrows + 1 entries, as
// stored: absolute offsets), col_idx / val (capacity `cap` entries, the rows' entries back to back).
extern "C" int spgemm_mat_rows_to_host(spgemm_ctx *ctx, const spgemm_mat *m, uint64_t row0, uint64_t nrows,
                                       uint64_t *row_ptr, uint32_t *col_idx, double *val, uint64_t cap) {
  if (!ctx || !m || !row_ptr) return fail(SPGEMM_ERR_INVALID, "NULL argument");
  if (row0 + nrows > m->rows) return fail(SPGEMM_ERR_INVALID, "row range beyond the matrix");
  TRY(set_device(ctx));
  HIP_TRY(hipMemcpyAsync(row_ptr, m->d_ptr + row0, (nrows + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  const uint64_t j0 = row_ptr[0], n = row_ptr[nrows] - j0;
  if (row_ptr[nrows] < j0 || row_ptr[nrows] > m->nnz) return fail(SPGEMM_ERR_STATE, "row pointer is not monotone");
  if (n > cap) return fail(SPGEMM_ERR_INVALID, "the rows hold %llu entries, capacity is %llu", (unsigned long long)n, (unsigned long long)cap);
  if (n && col_idx) HIP_TRY(hipMemcpyAsync(col_idx, m->d_col + j0, n * 4, hipMemcpyDeviceToHost, ctx->stream));
  if (n && val) HIP_TRY(hipMemcpyAsync(val, m->d_val + j0, n * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return SPGEMM_OK;
}

extern "C" int spgemm_mat_invalidate(spgemm_ctx *ctx, spgemm_mat *m) {
  if (!ctx || !m) return fail(SPGEMM_ERR_INVALID, "NULL argument");
  TRY(set_device(ctx));
  if (ctx->plan.valid && (ctx->plan.A == m || ctx->plan.B == m)) ctx->plan.valid = false;
  if (m->d_rowlen) {
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    pool_release(ctx, m->d_rowlen, m->bytes_rowlen);
    m->d_rowlen = nullptr;
    m->bytes_rowlen = 0;
  }
  m->h_ptr.clear();
  if (m->d_rowinfo) {
    HIP_TRY(hipStreamSynchronize(ctx->stream)); // a queued multiply may still read the row info
    pool_release(ctx, m->d_rowinfo, m->bytes_rowinfo);
    if (m->d_colcdf) pool_release(ctx, m->d_colcdf, m->bytes_colcdf);
    if (m->d_packed) pool_release(ctx, m->d_packed, m->bytes_packed);
    if (m->d_prow) pool_release(ctx, m->d_prow, m->bytes_prow);
    if (m->d_prec) pool_release(ctx, m->d_prec, m->bytes_prec);
    if (m->d_wb) pool_release(ctx, m->d_wb, m->bytes_wb);
    m->d_wb = nullptr;
    m->bytes_wb = 0;
    if (m->d_samp) pool_release(ctx, m->d_samp, m->bytes_samp);
    m->d_samp = nullptr;
    m->bytes_samp = 0;
    m->d_rowinfo = nullptr;
    m->d_colcdf = nullptr;
    m->d_packed = nullptr;
    m->d_prow = nullptr;
    m->d_prec = nullptr;
    m->bytes_rowinfo = m->bytes_colcdf = m->bytes_packed = m->bytes_prow = m->bytes_prec = 0;
  }
  return SPGEMM_OK;
}

// Returns what the handle holds to the context's pool (or to nobody, for wrapped arrays) and deletes it.
static void release_mat(spgemm_ctx *ctx, spgemm_mat *m) {
  if (ctx->plan.valid && (ctx->plan.A == m || ctx->plan.B == m)) ctx->plan.valid = false;
  if (m->d_rowinfo || m->d_rowlen || m->owns) (void)hipStreamSynchronize(ctx->stream);
  if (m->d_rowinfo) pool_release(ctx, m->d_rowinfo, m->bytes_rowinfo);
  if (m->d_colcdf) pool_release(ctx, m->d_colcdf, m->bytes_colcdf);
  if (m->d_packed) pool_release(ctx, m->d_packed, m->bytes_packed);
  if (m->d_prow) pool_release(ctx, m->d_prow, m->bytes_prow);
  if (m->d_prec) pool_release(ctx, m->d_prec, m->bytes_prec);
  if (m->d_wb) pool_release(ctx, m->d_wb, m->bytes_wb);
  if (m->d_samp) pool_release(ctx, m->d_samp, m->bytes_samp);
  if (m->d_rowlen) pool_release(ctx, m->d_rowlen, m->bytes_rowlen);
  if (m->owns) {
    pool_release(ctx, m->d_ptr, m->bytes_ptr);
    pool_release(ctx, m->d_col, m->bytes_col);
    pool_release(ctx, m->d_val, m->bytes_val);
  }
  delete m;
}

extern "C" int spgemm_mat_free(spgemm_ctx *ctx, spgemm_mat *m) {
  if (!m) return SPGEMM_OK;
  if (!ctx) return fail(SPGEMM_ERR_INVALID, "ctx is NULL");
  // only a handle that this context made and has not freed yet (a second free, or a free through
  // another context, is refused instead of being followed)
  auto it = ctx->live_mats.find(m);
  if (it == ctx->live_mats.end())
    return fail(SPGEMM_ERR_INVALID, "matrix handle does not belong to this context or was freed already");
  ctx->live_mats.erase(it);
  (void)hipSetDevice(ctx->device);
  release_mat(ctx, m);
  return SPGEMM_OK;
}

// ---- per-launch profile ----------------------------------------------------------------------------
static int prof_begin(spgemm_ctx *ctx, uint32_t phase, uint32_t path, uint32_t lanes, uint32_t rows, uint32_t p0,
                      uint32_t p1, size_t lds, const uint32_t *rows_bin) {
  if (!ctx->profile) return SPGEMM_OK;
  size_t need = (ctx->launches.size() + 1) * 2;
  while (ctx->prof_events.size() < need) {
    hipEvent_t e;
    HIP_TRY(hipEventCreate(&e));
    ctx->prof_events.push_back(e);
  }
  spgemm_ctx::LaunchRec r;
  memset(&r.info, 0, sizeof(r.info));
  r.info.phase = phase; r.info.path = path; r.info.lanes = lanes; r.info.rows = rows;
  r.info.param0 = p0; r.info.param1 = p1; r.info.lds_bytes = (uint32_t)lds;
  r.rows_bin = rows_bin;
  r.ev0 = (int)(ctx->launches.size() * 2);
  r.ev1 = r.ev0 + 1;
  ctx->launches.push_back(r);
  HIP_TRY(hipEventRecord(ctx->prof_events[r.ev0], ctx->stream));
  return SPGEMM_OK;
}
static int prof_end(spgemm_ctx *ctx) {
  if (!ctx->profile) return SPGEMM_OK;
  HIP_TRY(hipEventRecord(ctx->prof_events[ctx->launches.back().ev1], ctx->stream));
  return SPGEMM_OK;
}

// Sum nnz_a / nprod / nnz_c over the rows of every launch of `phase` (after the phase's events).
static int prof_collect(spgemm_ctx *ctx, const spgemm_mat *A, const uint32_t *r_nnz, uint32_t phase) {
  if (ctx->profile < 2) return SPGEMM_OK;
  for (auto &r : ctx->launches) {
    if (r.info.phase != phase) continue;
    unsigned long long *d = reinterpret_cast<unsigned long long *>(ctx->d_small) + 8;
    HIP_TRY(hipMemsetAsync(d, 0, 24, ctx->stream));
    hipLaunchKernelGGL(k_bin_sums, dim3(std::min<uint32_t>(div_up_u64(r.info.rows, 256), 1024)), dim3(256), 0,
                       ctx->stream, r.rows_bin, r.info.rows, A->d_ptr, ctx->r_nprod_prof, r_nnz, ctx->plan_gcnt,
                       (uint32_t)((r.info.phase == 0 && r.info.path == 0) || r.info.path == 3 || r.info.path == 5), d);
    HIP_TRY(hipGetLastError());
    unsigned long long h[3];
    HIP_TRY(hipMemcpyAsync(h, d, 24, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    r.info.nnz_a = h[0]; r.info.nprod = h[1]; r.info.nnz_c = h[2];
  }
  return SPGEMM_OK;
}

// Queues ONE launch that copies the given runs of device words to their places in the pinned host
// buffer (`host_field` is an address inside ctx->h_pinned); the host reads them after its next
// synchronisation of the stream.
struct Report {
  ReportSegs segs{};
  int n = 0;
  void add(const spgemm_ctx *ctx, void *host_field, const void *dev_src, size_t bytes) {
    const size_t off = reinterpret_cast<char *>(host_field) - reinterpret_cast<char *>(ctx->h_pinned);
    segs.s[n++] = ReportSeg{reinterpret_cast<const uint32_t *>(dev_src),
                            reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(ctx->d_pinned) + off), (uint32_t)(bytes / 4)};
  }
  int launch(spgemm_ctx *ctx) {
    for (int i = n; i < 6; ++i) segs.s[i] = ReportSeg{nullptr, nullptr, 0};
    hipLaunchKernelGGL(k_report, dim3(1), dim3(256), 0, ctx->stream, segs);
    HIP_TRY(hipGetLastError());
    return SPGEMM_OK;
  }
};

// ---- the multiply -------------------------------------------------------------------------------
struct HostCounts {
  uint64_t bin_off[MAX_BINS + 1];
  uint64_t total_a; // total_ub (symbolic phase) or nnz (numeric phase)
  uint64_t total_nprod;
  uint32_t long_count;
  uint32_t gmax_seen[2]; // largest row group: narrow rows, wide rows
  uint32_t b_unsorted;  // B rows whose columns are not strictly increasing
  uint32_t b_maxrun;    // longest run of rows of B with one column list (next to b_unsorted in the row info's counters)
  uint32_t max_len_big; // longest A row among the numeric phase's big rows
  uint64_t nprod_parts[NPROD_SLOTS];
};
static_assert(sizeof(HostCounts) <= PINNED_BYTES, "pinned staging buffer too small");


// symbolic phase: analysis, binning, per-bin symbolic kernels, scan of the row counts.
// On success ctx->plan describes the row arrays (in ctx->arena) and d_row_ptr holds the
// structural row pointer.
static int run_symbolic(spgemm_ctx *ctx, const spgemm_mat *A, const spgemm_mat *B, uint64_t *d_row_ptr) {
  if (A->cols != B->rows) return fail(SPGEMM_ERR_INVALID, "A.cols (%llu) != B.rows (%llu)", (unsigned long long)A->cols, (unsigned long long)B->rows);
  if (A->device != ctx->device || B->device != ctx->device) return fail(SPGEMM_ERR_INVALID, "matrix lives on another device");
  ctx->plan.valid = false;
  ctx->launches.clear();
  const uint32_t rows = (uint32_t)A->rows;
  spgemm_stats &st = ctx->stats;
  memset(&st, 0, sizeof(st));
  st.rows_a = A->rows; st.nnz_a = A->nnz; st.nnz_b = B->nnz;
  HIP_TRY(hipEventRecord(ctx->ev[0], ctx->stream));
  if (rows == 0) {
    HIP_TRY(hipMemsetAsync(d_row_ptr, 0, 8, ctx->stream));
    for (int i = 1; i <= 2; ++i) HIP_TRY(hipEventRecord(ctx->ev[i], ctx->stream));
    ctx->plan = spgemm_ctx::Plan();
    ctx->plan.valid = true; ctx->plan.A = A; ctx->plan.B = B;
    return SPGEMM_OK;
  }
  TRY(ensure_rowinfo(ctx, B));

  // arena layout
  const uint32_t pblocks = part_blocks(rows);
  const size_t hist_n = (size_t)MAX_BINS * pblocks;
  const uint32_t max_long = (uint32_t)std::min<uint64_t>(rows, A->nnz / (ctx->long_len + 1) + 1);
  size_t need = 0;
  need += 4 * align_up((size_t)rows * 4, 256);            // cmin, cmax, ub, nnz
  if (ctx->profile >= 2) need += align_up((size_t)rows * 8, 256) + 256 * 64; // per-row products, per-launch sums
  need += 4 * align_up((size_t)rows, 256);                // key, numkey, same, gcnt
  need += align_up((size_t)rows * 4, 256);                // lead
  need += align_up((size_t)rows * 4, 256);                // perm
  need += align_up(((size_t)rows + 1) * 8, 256);          // t_off
  need += align_up((size_t)max_long * 4, 256) + 256;      // long rows + counter
  need += align_up(hist_n * 4, 256) + align_up((hist_n + 1) * 8, 256); // partition histogram + scan
  need += align_up(((size_t)scan_parts(std::max<uint64_t>(rows, hist_n)) + 2) * 8, 256); // scan partials
  need += align_up((MAX_BINS + 1) * 8, 256) + NPROD_SLOTS * 8 + 256 + 256; // bin_off, counters
  need += align_up((32 + SLICE_MAX_QUEUES * SLICE_QUEUE_STRIDE) * 8, 256);  // zero count, flags, work queues of the slice kernel
  const bool short_aux = ctx->cp.short_max_nprod > 0;
  // padded records of B's rows: only where short rows are likely at all (mean products per row of A
  // within twice the short-row limit); without them the short-row kernel reads B's own arrays
  // (a B that holds only the sections a rank needs has empty rows: its mean row length is taken to
  // be at least A's)
  const double mean_a = (double)A->nnz / (double)std::max<uint64_t>(A->rows, 1);
  const double mean_prod = mean_a * std::max(mean_a, (double)B->nnz / (double)std::max<uint64_t>(B->rows, 1));
  const bool short_pk = short_aux && ctx->short_packed && (B->d_prow || mean_prod <= 2.0 * ctx->cp.short_max_nprod);
  if (short_pk) TRY(ensure_prec_layout(ctx, B));
  if (short_aux) need += align_up(A->nnz * 8, 256) + align_up(A->nnz * 4, 256); // short rows: B row start and length per A entry
  if (short_pk) need += align_up(A->nnz * 8, 256);                               // ... and the start of its padded records
  TRY(grow(ctx, ctx->arena, need));
  ArenaCursor ar{reinterpret_cast<char *>(ctx->arena.p), 0, ctx->arena.cap};
  uint32_t *r_cmin = ar.take<uint32_t>(rows), *r_cmax = ar.take<uint32_t>(rows);
  uint32_t *r_ub = ar.take<uint32_t>(rows), *r_nnz = ar.take<uint32_t>(rows);
  uint8_t *r_key = ar.take<uint8_t>(rows), *r_numkey = ar.take<uint8_t>(rows);
  uint8_t *r_same = ar.take<uint8_t>(rows), *r_gcnt = ar.take<uint8_t>(rows);
  uint32_t *r_lead = ar.take<uint32_t>(rows);
  uint32_t *perm = ar.take<uint32_t>(rows);
  uint64_t *t_off = ar.take<uint64_t>((size_t)rows + 1);
  uint32_t *long_rows = ar.take<uint32_t>(max_long);
  uint32_t *long_count = ar.take<uint32_t>(64);
  uint32_t *blk_hist = ar.take<uint32_t>(hist_n);
  uint64_t *blk_off = ar.take<uint64_t>(hist_n + 1);
  uint64_t *partial = ar.take<uint64_t>((size_t)scan_parts(std::max<uint64_t>(rows, hist_n)) + 2);
  uint64_t *bin_off = ar.take<uint64_t>(MAX_BINS + 1);
  uint64_t *total_nprod = ar.take<uint64_t>(NPROD_SLOTS);
  unsigned long long *zero_count = reinterpret_cast<unsigned long long *>(ar.take<uint64_t>(32 + SLICE_MAX_QUEUES * SLICE_QUEUE_STRIDE));
  uint64_t *r_nprod = ctx->profile >= 2 ? ar.take<uint64_t>(rows) : nullptr;
  // (with padded records and padded columns both short-row kernels start from a_pk)
  uint64_t *a_p0 = short_aux && !(short_pk && ctx->short_packed >= 2) ? ar.take<uint64_t>(A->nnz) : nullptr;
  uint32_t *a_len = short_aux ? ar.take<uint32_t>(A->nnz) : nullptr;
  uint64_t *a_pk = short_pk ? ar.take<uint64_t>(A->nnz) : nullptr;
  ctx->r_nprod_prof = r_nprod;
  ctx->plan_gcnt = r_gcnt;
  ctx->launches.clear();
  ctx->arena_used = ar.used;
  if (ar.used > ar.cap) return fail(SPGEMM_ERR_STATE, "internal: arena layout exceeds its size");

  {
    const uint32_t n_nnz16 = div_up_u64((uint64_t)rows * 4, 16), n_key16 = div_up_u64(rows, 16);
    hipLaunchKernelGGL(k_sym_clear, dim3(std::min<uint32_t>(div_up_u64(std::max<uint32_t>(n_nnz16, 64), 256), 2048)),
                       dim3(256), 0, ctx->stream, long_count, 64u, reinterpret_cast<uint32_t *>(total_nprod),
                       (uint32_t)(NPROD_SLOTS * 2), reinterpret_cast<uint4 *>(r_nnz), n_nnz16,
                       reinterpret_cast<uint4 *>(r_numkey), n_key16);
  }

  AnalyzeOut ao{r_cmin, r_cmax, r_ub, r_key, r_nprod, total_nprod, long_rows, long_count, r_same, a_p0, a_len, a_pk};
  // lanes per A row: measured, not derived - rows of a few dozen entries and more whose B rows
  // neighbours share (shell, diel: 0.39 against 0.53 ms and 0.44 against 0.56) go faster with 8,
  // rows of 16 entries with random B rows (ER, R-MAT) with 16 or 32
  const uint32_t analyze_lanes = ctx->analyze_lanes ? ctx->analyze_lanes : (A->nnz >= 24 * A->rows ? 8u : 16u);
  if (analyze_lanes == 8)
    hipLaunchKernelGGL(k_analyze<8>, dim3(div_up_u64((uint64_t)rows * 8, 256)), dim3(256), 0, ctx->stream, rows,
                       A->d_ptr, A->d_col, rowinfo_of(B), ao, ctx->cp, ctx->long_len);
  else if (analyze_lanes == 32)
    hipLaunchKernelGGL(k_analyze<32>, dim3(div_up_u64((uint64_t)rows * 32, 256)), dim3(256), 0, ctx->stream, rows,
                       A->d_ptr, A->d_col, rowinfo_of(B), ao, ctx->cp, ctx->long_len);
  else
  hipLaunchKernelGGL(k_analyze<16>, dim3(div_up_u64((uint64_t)rows * 16, 256)), dim3(256), 0, ctx->stream, rows,
                     A->d_ptr, A->d_col, rowinfo_of(B), ao, ctx->cp, ctx->long_len);
  if (max_long > 1 || A->nnz > ctx->long_len)
    hipLaunchKernelGGL(k_analyze_long, dim3(std::min<uint32_t>(max_long, 1024)), dim3(256), 0, ctx->stream, A->d_ptr,
                       A->d_col, rowinfo_of(B), ao, ctx->cp);
  hipLaunchKernelGGL(k_group, dim3(div_up_u64(rows, 256)), dim3(256), 0, ctx->stream, rows, A->d_ptr, r_same, r_key, r_ub,
                     r_gcnt, r_lead, ctx->row_groups, long_count + 1);
  HIP_TRY(hipGetLastError());
  TRY(launch_scan(ctx, r_ub, rows, t_off, partial));
  PartScratch ps{pblocks, blk_hist, blk_off, partial};
  TRY(launch_partition(ctx, r_key, rows, SYM_NBINS, ps, perm, bin_off));

  HostCounts *hc = reinterpret_cast<HostCounts *>(ctx->h_pinned);
  {
    Report rep;
    rep.add(ctx, hc->bin_off, bin_off, (SYM_NBINS + 1) * 8);
    rep.add(ctx, &hc->total_a, t_off + rows, 8);
    rep.add(ctx, hc->nprod_parts, total_nprod, NPROD_SLOTS * 8);
    rep.add(ctx, &hc->b_unsorted, &B->d_rowinfo[2 * B->rows].x, 8); // .x unsorted rows, .y longest run
    rep.add(ctx, hc->gmax_seen, long_count + 1, 8);
    TRY(rep.launch(ctx));
  }
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  const bool b_sorted = hc->b_unsorted == 0;
  const uint32_t b_maxrun = ctx->grp_runs ? hc->b_maxrun : 0u;
  // (only a B whose rows are all strictly increasing: the run walks rely on it)
  const uint32_t *same_bits = b_maxrun >= 2 && b_sorted ? rowinfo_of(B).same : nullptr;
  const uint32_t gmax_narrow = std::min<uint32_t>(std::max<uint32_t>(hc->gmax_seen[0], 1), GRP_MAX);
  const uint32_t gmax_wide = std::min<uint32_t>(std::max<uint32_t>(hc->gmax_seen[1], 1), GRP_MAX_WIDE);
  st.nprod = 0;
  for (uint32_t i = 0; i < NPROD_SLOTS; ++i) st.nprod += hc->nprod_parts[i];
  const uint64_t total_ub = hc->total_a;
  uint64_t boff[SYM_NBINS + 1];
  memcpy(boff, hc->bin_off, sizeof(boff));
  TRY(grow(ctx, ctx->temp_cols, std::max<uint64_t>(total_ub, 1) * 4));
  HIP_TRY(hipEventRecord(ctx->ev[1], ctx->stream));

  Csr cA{A->d_ptr, A->d_col, A->d_val}, cB{B->d_ptr, B->d_col, B->d_val};
  // with the big-row kernels available, rows of more than hash_num_slice outputs leave the LDS hash
  // kernel (whose largest tables leave room for two wavefronts per CU) for the slice kernel
  const uint32_t hash_num_max = (b_sorted && ctx->big_rows) ? ctx->hash_num_slice : 4096u;
  RowMeta rm{r_cmin, r_cmax, r_key, t_off, reinterpret_cast<uint32_t *>(ctx->temp_cols.p), r_nnz, r_numkey, r_gcnt, r_lead,
             hash_num_max, a_p0, a_len, a_pk};

  for (int key = 0; key < SYM_NBINS; ++key) {
    uint64_t cnt64 = boff[key + 1] - boff[key];
    if (!cnt64) continue;
    uint32_t cnt = (uint32_t)cnt64;
    const uint32_t *rows_bin = perm + boff[key];
    if (key < SYM_HASH0) {
      uint32_t tier = key >> 2;
      uint32_t words = (64u << tier) + 2u;
      const bool flat = (key & 3) < 3 && ctx->flat_walk; // short B rows: products laid end to end over the lanes
      size_t region = sym_window_region(words, flat);
      uint32_t wpb = pick_wpb(region);
      st.rows_window += cnt;
      TRY(prof_begin(ctx, 0, 0, flat ? 0 : 64, cnt, words, 0, region * wpb, rows_bin));
      auto sym_kern = k_sym_window<false, false>;
      if (flat) sym_kern = k_sym_window<true, false>;
      else if (walk_tail(ctx, B)) sym_kern = k_sym_window<false, true>;
      if (!flat && same_bits && ctx->sym_rpw > 1) {
        // B has runs of rows with one column list: the kernel whose wavefronts take several rows, the next one's description in flight
        const uint32_t rpw = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(ctx->sym_rpw, ctx->rpw_fill ? cnt / ((uint64_t)ctx->cu_count * 32 * ctx->rpw_fill) : cnt));
        // (units of 128 columns where the rows of B have more than 64 entries on average)
        hipLaunchKernelGGL(B->nnz > 64 * B->rows ? k_sym_window_runs<true> : k_sym_window_runs<false>,
                           dim3(div_up_u64(cnt, (uint64_t)wpb * rpw)), dim3(64 * wpb), region * wpb, ctx->stream,
                           rows_bin, cnt, cA, cB, rm, words, same_bits, wpb, rpw);
      } else
      hipLaunchKernelGGL(sym_kern, dim3(div_up_u64(cnt, wpb)), dim3(64 * wpb),
                         region * wpb, ctx->stream, rows_bin, cnt, cA, cB, rm, words, same_bits);
    } else if (key < SYM_GLOBAL0) {
      uint32_t tier = (key - SYM_HASH0) >> 2;
      uint32_t log2s = 7 + tier;
      if ((key & 3) == 0) { // short rows: count only
        log2s += ctx->short_sym_spread; // a table of 4x the products: probe chains half as long, and the kernel is bound by what it issues
        const size_t region = ((size_t)4 << log2s) + sh_scratch(false, tier <= 2 ? 4u : (uint32_t)SH_STEPS);
        const uint32_t wpb = pick_wpb(region);
        st.rows_hash += cnt;
        TRY(prof_begin(ctx, 0, 1, 1, cnt, log2s, 0, region * wpb, rows_bin));
        const uint32_t rpw = short_rows_per_wave(ctx, cnt);
        // tier t holds rows of at most 64 << t products: up to 4 steps of 64 products, or 8
        const uint32_t *pcol = nullptr;
        if (a_pk && ctx->short_packed >= 2) {
          TRY(ensure_prec(ctx, B));
          pcol = reinterpret_cast<const uint32_t *>(B->d_prec + B->prec_slots);
        }
        hipLaunchKernelGGL(tier <= 2 ? k_sym_count<4> : k_sym_count<SH_STEPS>, dim3(div_up_u64(cnt, (uint64_t)wpb * rpw)),
                           dim3(64 * wpb), region * wpb, ctx->stream, rows_bin, cnt, cA, cB, rm, log2s, wpb, rpw, pcol);
        HIP_TRY(hipGetLastError());
        TRY(prof_end(ctx));
        continue;
      }
      const bool flat = (key & 3) < 3 && ctx->flat_walk;
      size_t region = ((size_t)8 << log2s) + (flat ? FLAT_BYTES : 0);
      uint32_t wpb = pick_wpb(region);
      st.rows_hash += cnt;
      if ((!flat && log2s >= 12) || log2s >= 14) { // 32 KiB of table and more: the wavefronts of a workgroup share it
        TRY(prof_begin(ctx, 0, 1, log2s >= 14 ? 1024 : log2s >= 13 ? 512 : 256, cnt, log2s, 0, (size_t)8 << log2s, rows_bin));
        if (log2s >= 14) // 128 KiB: one workgroup of 16 wavefronts per CU
          hipLaunchKernelGGL(k_sym_hash_wg<1024>, dim3(cnt), dim3(1024), (size_t)8 << log2s, ctx->stream, rows_bin, cnt, cA, cB,
                             rm, log2s);
        else if (log2s >= 13)
          hipLaunchKernelGGL(k_sym_hash_wg<512>, dim3(cnt), dim3(512), (size_t)8 << log2s, ctx->stream, rows_bin, cnt, cA, cB,
                             rm, log2s);
        else
          hipLaunchKernelGGL(k_sym_hash_wg<256>, dim3(cnt), dim3(256), (size_t)8 << log2s, ctx->stream, rows_bin, cnt, cA, cB,
                             rm, log2s);
        HIP_TRY(hipGetLastError());
        TRY(prof_end(ctx));
        continue;
      }
      TRY(prof_begin(ctx, 0, 1, flat ? 0 : 64, cnt, log2s, 0, region * wpb, rows_bin));
      hipLaunchKernelGGL(flat ? k_sym_hash<true> : k_sym_hash<false>, dim3(div_up_u64(cnt, wpb)), dim3(64 * wpb),
                         region * wpb, ctx->stream, rows_bin, cnt, cA, cB, rm, log2s);
    } else if (b_sorted && ctx->big_rows) {
      bool bounds = false;
      if (ctx->big_sym_bounds) TRY(ensure_winbounds(ctx, B, std::min<uint32_t>(ctx->big_win_words, BIG_WIN_MAX), &bounds));
      st.rows_global += cnt;
      if (bounds) {
        // the pieces of B's rows per bitmap window are known in advance: no cursors, independent loads
        const uint32_t win_words = 1u << (B->wb_ws - 5u);
        const size_t lds = sym_big2_lds(win_words);
        // The rows in the order of what they cost (their products, as far as the upper bound of the row's
        // outputs tells), the dearest first: a workgroup takes one row, the rows of a hub are a hundred
        // times the mean, and one that starts late is what the launch ends on.
        const uint32_t *rows_use = rows_bin;
        if (ctx->big_sym_sorted && cnt > 1) {
          const size_t b_keys = align_up((size_t)cnt * 4, 256), b_rec = align_up((size_t)cnt * 12, 256);
          size_t b_tmp = 0;
          TRY(spgemm_internal_sort_records12(ctx->stream, nullptr, nullptr, nullptr, nullptr, cnt, nullptr, &b_tmp));
          b_tmp = align_up(b_tmp, 256);
          TRY(grow(ctx, ctx->big_cur, 2 * b_keys + 2 * b_rec + b_tmp)); // (the cursor walk's buffer: not in use on this path)
          char *bb = reinterpret_cast<char *>(ctx->big_cur.p);
          uint32_t *key_in = reinterpret_cast<uint32_t *>(bb), *key_out = reinterpret_cast<uint32_t *>(bb + b_keys);
          SliceItem *rec_in = reinterpret_cast<SliceItem *>(bb + 2 * b_keys), *rec_out = reinterpret_cast<SliceItem *>(bb + 2 * b_keys + b_rec);
          void *tmp = bb + 2 * b_keys + 2 * b_rec;
          hipLaunchKernelGGL(k_big_order_fill, dim3(div_up_u64(cnt, 256)), dim3(256), 0, ctx->stream, rows_bin, cnt, r_ub, key_in, rec_in);
          TRY(spgemm_internal_sort_records12(ctx->stream, key_in, key_out, rec_in, rec_out, cnt, tmp, &b_tmp));
          hipLaunchKernelGGL(k_big_order_rows, dim3(div_up_u64(cnt, 256)), dim3(256), 0, ctx->stream, rec_out, cnt, key_in);
          rows_use = key_in;
        }
        TRY(prof_begin(ctx, 0, 4, BIG_THREADS, cnt, win_words, 1, lds, rows_bin));
        hipLaunchKernelGGL(k_sym_big2, dim3(cnt), dim3(BIG_THREADS), lds, ctx->stream, rows_use, cnt, cA, cB, rm, B->d_wb,
                           B->wb_nwin, B->wb_ws);
#ifdef SG_SYM2_STAMPS
        {
          unsigned long long ph[8], zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
          HIP_TRY(hipMemcpyFromSymbol(ph, HIP_SYMBOL(g_big2_stamps), sizeof(ph)));
          HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_big2_stamps), zero, sizeof(zero)));
          fprintf(stderr, "sym_big2 stamps (Mcycles, all waves): pieces+scan %.1f walk %.1f barrier %.1f popcount %.1f barrier %.1f write %.1f barrier %.1f\n",
                  ph[0] / 1e6, ph[1] / 1e6, ph[2] / 1e6, ph[3] / 1e6, ph[4] / 1e6, ph[5] / 1e6, ph[6] / 1e6);
        }
#endif
      } else {
      // windows of up to 2^20 columns (128 KiB of LDS); a narrower matrix gets one window
      uint32_t win_words = (uint32_t)std::min<uint64_t>(std::min<uint32_t>(ctx->big_win_words, BIG_WIN_MAX), align_up((B->cols + 31) / 32, BIG_THREADS));
      size_t lds = sym_big_lds(win_words);
      if (A->nnz > BIG_CUR) TRY(grow(ctx, ctx->big_cur, A->nnz * 16));
      TRY(prof_begin(ctx, 0, 4, BIG_THREADS, cnt, win_words, 0, lds, rows_bin));
      unsigned long long *sym_stamps = reinterpret_cast<unsigned long long *>(ctx->d_small) + 32; // diagnostic builds only
#ifdef SG_SYM_STAMPS
      HIP_TRY(hipMemsetAsync(sym_stamps, 0, 128, ctx->stream));
#endif
      hipLaunchKernelGGL(k_sym_big, dim3(cnt), dim3(BIG_THREADS), lds, ctx->stream, rows_bin, cnt, cA, cB, rm, win_words,
                         reinterpret_cast<uint64_t *>(ctx->big_cur.p), A->nnz, sym_stamps);
#ifdef SG_SYM_STAMPS
      {
        unsigned long long ph[16];
        HIP_TRY(hipMemcpy(ph, sym_stamps, sizeof(ph), hipMemcpyDeviceToHost));
        fprintf(stderr, "  long: position loads %.1f first step loads %.1f (rest: adds and continuation, in long below); groups of four %llu, continuation steps %llu (per wave)\n",
                ph[8] / 1e6, ph[9] / 1e6, ph[10], ph[11]);
        fprintf(stderr, "sym_big stamps (Mcycles, all waves): classify %.1f barrier %.1f mid %.1f long %.1f barrier %.1f popcount %.1f barrier+scan+write %.1f barrier %.1f\n",
                ph[0] / 1e6, ph[1] / 1e6, ph[2] / 1e6, ph[3] / 1e6, ph[4] / 1e6, ph[5] / 1e6, ph[6] / 1e6, ph[7] / 1e6);
      }
#endif
      }
    } else {
      uint64_t words_total = (B->cols + 31) / 32 + 1;
      uint32_t blocks = std::min<uint32_t>(cnt, ctx->global_blocks_sym);
      size_t bytes = (size_t)blocks * words_total * 4;
      if (bytes > ctx->gscratch.cap) {
        TRY(grow(ctx, ctx->gscratch, std::max(bytes, (size_t)ctx->global_blocks_num * words_total * 8)));
        HIP_TRY(hipMemsetAsync(ctx->gscratch.p, 0, ctx->gscratch.cap, ctx->stream));
      }
      st.rows_global += cnt;
      TRY(prof_begin(ctx, 0, 2, 64, cnt, 0, 0, 0, rows_bin));
      hipLaunchKernelGGL(k_sym_global, dim3(blocks), dim3(256), 0, ctx->stream, rows_bin, cnt, cA, cB, rm,
                         reinterpret_cast<uint32_t *>(ctx->gscratch.p), words_total);
    }
    HIP_TRY(hipGetLastError());
    TRY(prof_end(ctx));
  }
  st.rows_empty = rows - st.rows_window - st.rows_hash - st.rows_global;
  HIP_TRY(hipEventRecord(ctx->ev[2], ctx->stream));
  TRY(prof_collect(ctx, A, r_nnz, 0));
  TRY(launch_scan(ctx, r_nnz, rows, d_row_ptr, partial));

  spgemm_ctx::Plan &pl = ctx->plan;
  pl.A = A; pl.B = B;
  pl.r_cmin = r_cmin; pl.r_cmax = r_cmax; pl.r_nnz = r_nnz;
  pl.r_key = r_key; pl.r_numkey = r_numkey; pl.t_off = t_off; pl.perm = perm;
  pl.r_gcnt = r_gcnt; pl.r_lead = r_lead;
  pl.a_p0 = a_p0; pl.a_len = a_len; pl.a_pk = a_pk;
  pl.gmax_narrow = gmax_narrow; pl.gmax_wide = gmax_wide;
  pl.blk_hist = blk_hist; pl.blk_off = blk_off; pl.partial = partial; pl.bin_off = bin_off;
  pl.zero_count = zero_count; pl.pblocks = pblocks;
  pl.b_sorted = b_sorted;
  pl.b_maxrun = b_maxrun;
  pl.valid = true;
  return SPGEMM_OK;
}

// Queued right behind the symbolic phase: rows partitioned by numeric bin, counters zeroed, longest
// big A row; the bin offsets, that length and the structural nnz are copied to the pinned buffer.
// numeric_counts_ready() is called after the caller's stream synchronisation and moves them into
// the plan: ONE host round trip between the symbolic and the numeric phase.
static int numeric_prepare(spgemm_ctx *ctx, const spgemm_mat *A, const uint64_t *d_row_ptr) {
  spgemm_ctx::Plan &pl = ctx->plan;
  pl.num_ready = false;
  const uint32_t rows = (uint32_t)A->rows;
  HostCounts *hc = reinterpret_cast<HostCounts *>(ctx->h_pinned);
  hc->total_a = 0;
  hc->max_len_big = 0;
  memset(hc->bin_off, 0, sizeof(hc->bin_off));
  if (rows == 0) return SPGEMM_OK;
  Report rep;
  rep.add(ctx, &hc->total_a, d_row_ptr + rows, 8);
  PartScratch ps{pl.pblocks, pl.blk_hist, pl.blk_off, pl.partial};
  TRY(launch_partition(ctx, pl.r_numkey, rows, NUM_NBINS, ps, pl.perm, pl.bin_off));
  HIP_TRY(hipMemsetAsync(pl.zero_count, 0, (32 + SLICE_MAX_QUEUES * SLICE_QUEUE_STRIDE) * 8, ctx->stream));
  // zero_count[2] (low word): longest big A row; zero_count[32 ...]: work counters of the slice kernel
  uint32_t *max_len_big = reinterpret_cast<uint32_t *>(pl.zero_count + 2);
  if (pl.b_sorted && ctx->big_rows) {
    hipLaunchKernelGGL(k_slice_maxlen, dim3(std::min<uint32_t>(div_up_u64(rows, 256), 2048)), dim3(256), 0, ctx->stream,
                       rows, A->d_ptr, pl.r_numkey, max_len_big);
    HIP_TRY(hipGetLastError());
    rep.add(ctx, &hc->max_len_big, max_len_big, 4);
  }
  rep.add(ctx, hc->bin_off, pl.bin_off, (NUM_NBINS + 1) * 8);
  TRY(rep.launch(ctx));
  return SPGEMM_OK;
}
static uint64_t numeric_counts_ready(spgemm_ctx *ctx) { // after the stream has been synchronised
  spgemm_ctx::Plan &pl = ctx->plan;
  const HostCounts *hc = reinterpret_cast<const HostCounts *>(ctx->h_pinned);
  memcpy(pl.num_boff, hc->bin_off, (NUM_NBINS + 1) * 8);
  pl.max_len_big = hc->max_len_big;
  pl.nnz_structural = hc->total_a;
  pl.num_ready = true;
  return pl.nnz_structural;
}

// numeric phase; needs ctx->plan from run_symbolic on the same operands.
static int run_numeric(spgemm_ctx *ctx, const spgemm_mat *A, const spgemm_mat *B, uint64_t *d_row_ptr,
                       uint32_t *d_col, double *d_val, uint64_t nnz_structural, uint64_t *nnz_final,
                       bool allow_realloc, spgemm_mat *Cmat) {
  spgemm_ctx::Plan &pl = ctx->plan;
  spgemm_stats &st = ctx->stats;
  const uint32_t rows = (uint32_t)A->rows;
  st.nnz_structural = nnz_structural;
  *nnz_final = nnz_structural;
  if (rows == 0 || nnz_structural == 0) {
    for (int i = 3; i <= 5; ++i) HIP_TRY(hipEventRecord(ctx->ev[i], ctx->stream));
    return SPGEMM_OK;
  }
  uint32_t *perm = pl.perm;
  uint64_t *partial = pl.partial;
  unsigned long long *zero_count = pl.zero_count;
  if (!pl.num_ready) return fail(SPGEMM_ERR_STATE, "internal: the numeric bins were not read back");
  const bool slices = pl.b_sorted && ctx->big_rows;
  HostCounts *hc = reinterpret_cast<HostCounts *>(ctx->h_pinned);
  const uint64_t list_cap = slices ? std::max<uint32_t>(pl.max_len_big, 64) : 0;
  uint64_t boff[NUM_NBINS + 1];
  memcpy(boff, pl.num_boff, sizeof(boff));
  if (ctx->poison_output) { // a test aid: an entry of C that no kernel writes then shows (column 2^32 - 1, value NaN)
    HIP_TRY(hipMemsetAsync(d_col, 0xFF, nnz_structural * 4, ctx->stream));
    HIP_TRY(hipMemsetAsync(d_val, 0xFF, nnz_structural * 8, ctx->stream));
  }
  HIP_TRY(hipEventRecord(ctx->ev[3], ctx->stream));

  Csr cA{A->d_ptr, A->d_col, A->d_val}, cB{B->d_ptr, B->d_col, B->d_val};
  RowMeta rm{pl.r_cmin, pl.r_cmax, pl.r_key, pl.t_off, reinterpret_cast<uint32_t *>(ctx->temp_cols.p), pl.r_nnz,
             pl.r_numkey, pl.r_gcnt, pl.r_lead, 4096u, pl.a_p0, pl.a_len, pl.a_pk};
  NumOut no{d_row_ptr, d_col, d_val, zero_count};
  const uint32_t hw_order = ctx->hw_lane_order && ctx->lane_order_adds ? 1u : 0u;
  const size_t lds_max = 160 * 1024;

  for (int key = 0; key < NUM_NBINS; ++key) {
    uint64_t cnt64 = boff[key + 1] - boff[key];
    if (!cnt64) continue;
    uint32_t cnt = (uint32_t)cnt64;
    const uint32_t *rows_bin = perm + boff[key];
    if (key >= NUM_GRP0) {
      uint32_t st_ = (uint32_t)(key - NUM_GRP0) >> 2, q = (uint32_t)(key - NUM_GRP0) & 3;
      const int tier = q < 2 ? 0 : q == 2 ? 1 : 2; // 0: up to 8 rows; 1, 2: wide groups (see num_key_of)
      uint32_t words = (64u << st_) + 2u, cap = q == 0 ? 128u : q == 2 ? GRP_CAP_MID : GRP_CAP_TOP;
      const uint32_t gmax = tier ? pl.gmax_wide : pl.gmax_narrow;
      const uint32_t alen = tier == 0 ? GRP_ALEN : tier == 1 ? GRP_ALEN_MID : GRP_ALEN_WIDE;
      // B has runs of rows with one column list: the kernel that ranks a column once per run and sums
      // in registers; instantiated for runs of up to 3, 5, 8 rows of B and groups of up to 3, 5, 8
      // (wide: 3, 4) rows of A, whose accumulator rows it keeps in LDS
      const bool use_runs = pl.b_maxrun >= 2 && pl.b_sorted;
      const int rsel = pl.b_maxrun <= 3 ? 0 : pl.b_maxrun <= 5 ? 1 : 2;
      const int gsel = tier ? (gmax <= 3 ? 0 : 1) : (gmax <= 3 ? 0 : gmax <= 5 ? 1 : 2);
      const uint32_t gmax_lds = !use_runs ? gmax : tier ? (gsel ? 4u : 3u) : (gsel == 0 ? 3u : gsel == 1 ? 5u : 8u);
      const bool compact = ctx->grp_compact == 2 ? grp_use_compact(words, cap, gmax_lds, alen) : ctx->grp_compact != 0;
      size_t region = use_runs ? grp_run_region(compact, words, cap, gmax_lds, tier != 0) : grp_region(compact, words, cap, gmax_lds, alen);
      if (region > lds_max) return fail(SPGEMM_ERR_UNSUPPORTED, "LDS region of %zu bytes does not fit", region);
      uint32_t wpb = pick_wpb(region);
      TRY(prof_begin(ctx, 1, use_runs ? 5 : 3, 64, cnt, words, cap, region * wpb, rows_bin));
      if (use_runs) {
        using run_kernel_t = void (*)(const uint32_t *, uint32_t, Csr, Csr, const uint32_t *, RowMeta, NumOut, uint32_t, uint32_t, uint32_t, uint32_t);
#define SG_RUN_ROW(W, C, G, T) {k_num_grp_run<W, C, 3, G, T>, k_num_grp_run<W, C, 5, G, T>, k_num_grp_run<W, C, 8, G, T>}
#define SG_RUN_G(C, T) {SG_RUN_ROW(false, C, 3, T), SG_RUN_ROW(false, C, 5, T), SG_RUN_ROW(false, C, 8, T)}
        // [cap 128 / 512][flat / two-level bitmap][rows of A per group][rows of B per run]
        static const run_kernel_t narrow[2][2][3][3] = {{SG_RUN_G(false, 2), SG_RUN_G(true, 2)}, {SG_RUN_G(false, 8), SG_RUN_G(true, 8)}};
        static const run_kernel_t wide[2][2][3] = {{SG_RUN_ROW(true, false, 3, 8), SG_RUN_ROW(true, false, 4, 8)},
                                                   {SG_RUN_ROW(true, true, 3, 8), SG_RUN_ROW(true, true, 4, 8)}};
#undef SG_RUN_G
#undef SG_RUN_ROW
        const run_kernel_t kern = tier ? wide[compact ? 1 : 0][gsel][rsel] : narrow[cap <= 128 ? 0 : 1][compact ? 1 : 0][gsel][rsel];
        // groups a wavefront takes one after the other: as many as leave 16 wavefronts of work per slot of the device
        const uint32_t rpw = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(ctx->grp_rpw, ctx->rpw_fill ? cnt / ((uint64_t)ctx->cu_count * 32 * ctx->rpw_fill) : cnt));
        hipLaunchKernelGGL(kern, dim3(div_up_u64(cnt, (uint64_t)wpb * rpw)), dim3(64 * wpb), region * wpb, ctx->stream, rows_bin, cnt, cA, cB,
                           rowinfo_of(B).same, rm, no, words, cap, wpb, rpw);
      } else {
      auto kern = compact ? (tier == 0 ? k_num_window_grp<0, true> : tier == 1 ? k_num_window_grp<1, true> : k_num_window_grp<2, true>)
                          : (tier == 0 ? k_num_window_grp<0, false> : tier == 1 ? k_num_window_grp<1, false> : k_num_window_grp<2, false>);
      if (walk_tail(ctx, B))
        kern = compact ? (tier == 0 ? k_num_window_grp<0, true, true> : tier == 1 ? k_num_window_grp<1, true, true> : k_num_window_grp<2, true, true>)
                       : (tier == 0 ? k_num_window_grp<0, false, true> : tier == 1 ? k_num_window_grp<1, false, true> : k_num_window_grp<2, false, true>);
      hipLaunchKernelGGL(kern,
                         dim3(div_up_u64(cnt, wpb)), dim3(64 * wpb),
                         region * wpb, ctx->stream, rows_bin, cnt, cA, cB, rm, no, words, cap, wpb, gmax);
      }
    } else if (key >= NUM_HASH0 && key < NUM_GLOBAL && (key & 3) == 0) { // short rows
      const uint32_t log2pm = 6 + ((uint32_t)(key - NUM_HASH0) >> 2);
      // 32-bit table entries (column << log2pm | product number) when every column fits beside the number
      const bool k32 = ctx->short_key32 && B->cols <= (1ull << (32 - log2pm)) - 1;
      const size_t region = num_short_region(log2pm, k32);
      if (region > lds_max) return fail(SPGEMM_ERR_UNSUPPORTED, "LDS region of %zu bytes does not fit", region);
      const uint32_t wpb = pick_wpb(region);
      TRY(ensure_colcdf(ctx, B));
      if (pl.a_pk) TRY(ensure_prec(ctx, B));
      ColMap cm{B->d_colcdf, B->cdf_shift, 1.0f / (float)(1ull << B->cdf_shift), pl.a_pk ? B->d_prec : nullptr, nullptr};
      TRY(prof_begin(ctx, 1, 1, 1, cnt, log2pm, 0, region * wpb, rows_bin));
      const uint32_t rpw = short_rows_per_wave(ctx, cnt, true);
      const bool four = num_short_steps(log2pm) == 4;
      auto kern = k32 ? (four ? k_num_short<4, true> : k_num_short<SH_STEPS, true>)
                      : (four ? k_num_short<4, false> : k_num_short<SH_STEPS, false>);
      hipLaunchKernelGGL(kern, dim3(div_up_u64(cnt, (uint64_t)wpb * rpw)),
                         dim3(64 * wpb), region * wpb, ctx->stream, rows_bin, cnt, cA, cB, rm, no, cm, log2pm, wpb, rpw);
    } else if (key < NUM_GLOBAL) {
      const bool flat = (key & 3) < 3 && ctx->flat_walk;
      const bool window = key < NUM_HASH0;
      size_t region;
      uint32_t p0, p1 = 0;
      if (window) {
        uint32_t st_ = (key >> 2) >> 2, ct = (key >> 2) & 3;
        p0 = (64u << st_) + 2u;   // words
        p1 = 128u << (2 * ct);    // cap
        region = (size_t)p0 * 8 + (size_t)p1 * 8 + (flat ? (size_t)p1 * 4 + FLAT_BYTES : 0);
      } else {
        uint32_t ct = (uint32_t)(key - NUM_HASH0) >> 2;
        p0 = 7 + ct;              // log2 S
        region = ((size_t)10 << p0) + (flat ? ((size_t)2 << p0) + FLAT_BYTES : 0);
      }
      if (region > lds_max) return fail(SPGEMM_ERR_UNSUPPORTED, "LDS region of %zu bytes does not fit", region);
      uint32_t wpb = pick_wpb(region);
      dim3 grid(div_up_u64(cnt, wpb)), block(64 * wpb);
      size_t lds = (size_t)wpb * region;
      TRY(prof_begin(ctx, 1, window ? 0 : 1, flat ? 0 : 64, cnt, p0, p1, lds, rows_bin));
      if (window)
        hipLaunchKernelGGL(flat ? k_num_window<true> : k_num_window<false>, grid, block, lds, ctx->stream, rows_bin, cnt,
                           cA, cB, rm, no, p0, p1, wpb, hw_order);
      else
        hipLaunchKernelGGL(flat ? k_num_hash<true> : k_num_hash<false>, grid, block, lds, ctx->stream, rows_bin, cnt, cA,
                           cB, rm, no, p0, wpb, hw_order);
    } else if (slices) {
      // items = runs of chunks of SLICE_W outputs; at most one per chunk
      const uint64_t max_items = nnz_structural / SLICE_W + cnt;
      uint32_t blocks = ctx->slice_blocks ? ctx->slice_blocks : (uint32_t)ctx->cu_count * std::min<uint32_t>(16u, (uint32_t)(lds_max / slice_lds(hw_order)));
      blocks = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(blocks, ((uint64_t)4 << 30) / (list_cap * SLICE_ENTRY_BYTES)));
      const size_t b_count = align_up((size_t)cnt * 4, 256), b_off = align_up(((size_t)cnt + 1) * 8, 256);
      const size_t b_items = align_up(max_items * sizeof(SliceItem), 256);
      const size_t b_lists = align_up((size_t)blocks * list_cap * SLICE_ENTRY_BYTES, 256);
      // sorted items: a run has at least four chunks, so a row of n outputs has at most n / (4 SLICE_W) + 2 items
      // (and never more than max_items, which is what the item arrays hold: with many rows of few chunks
      // the first bound is the larger one)
      const size_t sort_n = ctx->slice_sorted ? (size_t)std::min<uint64_t>(max_items, nnz_structural / (4ull * SLICE_W) + 2 * (uint64_t)cnt) : 0;
      const size_t b_keys = align_up(sort_n * 4, 256);
      size_t b_sort_tmp = 0;
      if (sort_n) {
        TRY(spgemm_internal_sort_records12(ctx->stream, nullptr, nullptr, nullptr, nullptr, sort_n, nullptr, &b_sort_tmp));
        b_sort_tmp = align_up(b_sort_tmp, 256);
      }
      TRY(grow(ctx, ctx->slice_buf, b_count + b_off + b_items + b_lists + (sort_n ? 2 * b_keys + b_items + b_sort_tmp : 0)));
      char *sb = reinterpret_cast<char *>(ctx->slice_buf.p);
      uint32_t *s_count = reinterpret_cast<uint32_t *>(sb);
      uint64_t *s_off = reinterpret_cast<uint64_t *>(sb + b_count);
      SliceItem *s_items = reinterpret_cast<SliceItem *>(sb + b_count + b_off);
      unsigned char *s_lists = reinterpret_cast<unsigned char *>(sb + b_count + b_off + b_items);
      uint32_t *s_key_in = reinterpret_cast<uint32_t *>(sb + b_count + b_off + b_items + b_lists);
      uint32_t *s_key_out = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(s_key_in) + b_keys);
      SliceItem *s_items_sorted = reinterpret_cast<SliceItem *>(reinterpret_cast<char *>(s_key_out) + b_keys);
      void *s_sort_tmp = reinterpret_cast<char *>(s_items_sorted) + b_items;
      if (ctx->slice_packed) TRY(ensure_packed(ctx, B));
      else if (ctx->slice_sampled) TRY(ensure_colsample(ctx, B));
      TRY(prof_begin(ctx, 1, 4, 64, cnt, SLICE_W, blocks, slice_lds(hw_order), rows_bin));
      hipLaunchKernelGGL(k_slice_count, dim3(div_up_u64(cnt, 256)), dim3(256), 0, ctx->stream, rows_bin, cnt, A->d_ptr,
                         pl.r_nnz, s_count, ctx->slice_run_min, ctx->slice_run_div);
      TRY(launch_scan(ctx, s_count, cnt, s_off, partial));
      if (sort_n) HIP_TRY(hipMemsetAsync(s_key_in, 0xFF, sort_n * 4, ctx->stream)); // unused tail: sorts behind every item
      hipLaunchKernelGGL(k_slice_fill, dim3(div_up_u64(cnt, 256)), dim3(256), 0, ctx->stream, rows_bin, cnt, A->d_ptr,
                         pl.r_nnz, s_off, s_items, sort_n ? s_key_in : nullptr, rm, ctx->slice_run_min, ctx->slice_run_div, ctx->slice_sorted >= 2 ? ctx->slice_cost_w | (ctx->slice_sorted == 3 ? 1u << 16 : 0u) : 0u);
      if (sort_n) {
        // Items in the order of their first column: the wavefronts that run together then work on
        // the same band of columns, in whatever rows, and so on the same stretch of the (hub) rows
        // of B, which stays in the caches instead of being fetched again for every row of A.
        TRY(spgemm_internal_sort_records12(ctx->stream, s_key_in, s_key_out, s_items, s_items_sorted, sort_n, s_sort_tmp,
                                           &b_sort_tmp));
      }
      hipLaunchKernelGGL(k_num_slice, dim3(blocks), dim3(64), slice_lds(hw_order), ctx->stream,
                         sort_n ? s_items_sorted : s_items, s_off + cnt, cA, cB, rm, no,
                         zero_count + 32, s_lists, list_cap, B->nnz, ctx->slice_packed ? B->d_packed : nullptr, ctx->slice_queues, hw_order,
                         ctx->slice_sampled && !ctx->slice_packed ? B->d_samp : nullptr);
    } else {
      uint64_t words_total = (B->cols + 31) / 32 + 1;
      uint32_t blocks = std::min<uint32_t>(cnt, ctx->global_blocks_num);
      size_t bytes = (size_t)blocks * words_total * 8;
      if (bytes > ctx->gscratch.cap) {
        TRY(grow(ctx, ctx->gscratch, bytes));
        HIP_TRY(hipMemsetAsync(ctx->gscratch.p, 0, ctx->gscratch.cap, ctx->stream));
      }
      TRY(prof_begin(ctx, 1, 2, 64, cnt, 0, 0, 0, rows_bin));
      hipLaunchKernelGGL(k_num_global, dim3(blocks), dim3(64), 0, ctx->stream, rows_bin, cnt, cA, cB, rm, no,
                         reinterpret_cast<unsigned long long *>(ctx->gscratch.p), words_total);
    }
    HIP_TRY(hipGetLastError());
    TRY(prof_end(ctx));
  }
  HIP_TRY(hipEventRecord(ctx->ev[4], ctx->stream));
  TRY(prof_collect(ctx, A, pl.r_nnz, 1));
  {
    Report rep;
    rep.add(ctx, &hc->total_a, zero_count, 8);
    rep.add(ctx, &hc->total_nprod, zero_count + 3, 8);
    TRY(rep.launch(ctx));
  }
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  if (hc->total_nprod) return fail(SPGEMM_ERR_STATE, "internal: the big-row numeric kernel met a damaged work list");
#ifdef SG_GRP_STAMPS
  {
    unsigned long long ph[8];
    HIP_TRY(hipMemcpy(ph, zero_count + 8, sizeof(ph), hipMemcpyDeviceToHost));
    fprintf(stderr, "grp_run stamps (Mticks, all waves): row state of the wave's groups %.1f wait for description %.1f bitmap set-up %.1f units %.1f write-out %.1f\n",
            ph[0] / 1e6, ph[1] / 1e6, ph[2] / 1e6, ph[3] / 1e6, ph[4] / 1e6);
  }
#endif
#ifdef SG_SLICE_STAMPS
  {
    unsigned long long ph[20];
    HIP_TRY(hipMemcpy(ph, zero_count + 8, sizeof(ph), hipMemcpyDeviceToHost));
    fprintf(stderr, "slice stamps (Mcycles, all waves): grab %.1f list %.1f fill %.1f scan/other %.1f per-row %.1f flat %.1f write %.1f\n",
            ph[0] / 1e6, ph[1] / 1e6, ph[2] / 1e6, ph[3] / 1e6, ph[4] / 1e6, ph[5] / 1e6, ph[6] / 1e6);
    fprintf(stderr, "  flat: first columns %.1f search %.1f scan+fetch %.1f (flat above = the add loop)\n", ph[7] / 1e6, ph[8] / 1e6, ph[9] / 1e6);
    fprintf(stderr, "  flat blocks %llu (searched %llu, in-register rows %llu) pieces %llu products %llu; pieces of one %llu, of more than four %llu with %llu products; long blocks %llu pieces %llu\n",
            ph[10], ph[11], ph[14], ph[12], ph[13], ph[15], ph[16], ph[17], ph[18], ph[19]);
  }
#endif
  uint64_t zeros = hc->total_a;
  if (zeros) {
    // Entries that summed to exactly 0.0 are dropped, as Cells::add erases them
    // (reference include/matrix_impl.hpp:90-93).
    uint64_t keep_nnz = nnz_structural - zeros;
    uint32_t *keep = pl.r_nnz; // reuse: per-row surviving count
    hipLaunchKernelGGL(k_count_nonzero, dim3(div_up_u64((uint64_t)rows * 8, 256)), dim3(256), 0, ctx->stream, rows,
                       d_row_ptr, d_val, keep);
    void *ncol = nullptr, *nval = nullptr, *nptr = nullptr;
    size_t bcol = 0, bval = 0, bptr = 0;
    int rc = pool_alloc(ctx, keep_nnz * 4, &ncol, &bcol);
    if (rc == SPGEMM_OK) rc = pool_alloc(ctx, keep_nnz * 8, &nval, &bval);
    if (rc == SPGEMM_OK) rc = pool_alloc(ctx, ((size_t)rows + 1) * 8, &nptr, &bptr);
    if (rc == SPGEMM_OK) rc = launch_scan(ctx, keep, rows, (uint64_t *)nptr, partial);
    if (rc == SPGEMM_OK) {
      hipLaunchKernelGGL(k_compact_rows, dim3(div_up_u64((uint64_t)rows * 64, 256)), dim3(256), 0, ctx->stream, rows,
                         d_row_ptr, d_col, d_val, (const uint64_t *)nptr, (uint32_t *)ncol, (double *)nval);
      hipError_t e = hipGetLastError();
      if (e != hipSuccess) rc = fail(SPGEMM_ERR_HIP, "k_compact_rows launch failed: %s", hipGetErrorString(e));
    }
    if (rc != SPGEMM_OK) { // nothing that was taken for the compaction stays behind
      (void)hipStreamSynchronize(ctx->stream);
      pool_release(ctx, nptr, bptr);
      pool_release(ctx, ncol, bcol);
      pool_release(ctx, nval, bval);
      return rc;
    }
    if (allow_realloc && Cmat) {
      hipError_t e = hipStreamSynchronize(ctx->stream);
      if (e != hipSuccess) {
        pool_release(ctx, nptr, bptr);
        pool_release(ctx, ncol, bcol);
        pool_release(ctx, nval, bval);
        return fail(SPGEMM_ERR_HIP, "compaction failed: %s", hipGetErrorString(e));
      }
      pool_release(ctx, Cmat->d_ptr, Cmat->bytes_ptr);
      pool_release(ctx, Cmat->d_col, Cmat->bytes_col);
      pool_release(ctx, Cmat->d_val, Cmat->bytes_val);
      Cmat->d_ptr = (uint64_t *)nptr; Cmat->bytes_ptr = bptr;
      Cmat->d_col = (uint32_t *)ncol; Cmat->bytes_col = bcol;
      Cmat->d_val = (double *)nval; Cmat->bytes_val = bval;
    } else {
      hipError_t e = hipMemcpyAsync(d_row_ptr, nptr, ((size_t)rows + 1) * 8, hipMemcpyDeviceToDevice, ctx->stream);
      if (e == hipSuccess && keep_nnz) e = hipMemcpyAsync(d_col, ncol, keep_nnz * 4, hipMemcpyDeviceToDevice, ctx->stream);
      if (e == hipSuccess && keep_nnz) e = hipMemcpyAsync(d_val, nval, keep_nnz * 8, hipMemcpyDeviceToDevice, ctx->stream);
      hipError_t e2 = hipStreamSynchronize(ctx->stream);
      pool_release(ctx, nptr, bptr);
      pool_release(ctx, ncol, bcol);
      pool_release(ctx, nval, bval);
      if (e != hipSuccess || e2 != hipSuccess)
        return fail(SPGEMM_ERR_HIP, "copying the compacted result failed: %s", hipGetErrorString(e != hipSuccess ? e : e2));
    }
    *nnz_final = keep_nnz;
    pl.valid = false; // r_nnz was overwritten
  }
  HIP_TRY(hipEventRecord(ctx->ev[5], ctx->stream));
  return SPGEMM_OK;
}

static int finish_stats(spgemm_ctx *ctx, uint64_t rows_c, uint64_t nnz_final) {
  spgemm_stats &st = ctx->stats;
  HIP_TRY(hipEventSynchronize(ctx->ev[5]));
  float ms = 0;
  HIP_TRY(hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1])); st.ms_analyze = ms;
  HIP_TRY(hipEventElapsedTime(&ms, ctx->ev[1], ctx->ev[2])); st.ms_symbolic = ms;
  HIP_TRY(hipEventElapsedTime(&ms, ctx->ev[2], ctx->ev[3])); st.ms_scan = ms;
  HIP_TRY(hipEventElapsedTime(&ms, ctx->ev[3], ctx->ev[4])); st.ms_numeric = ms;
  HIP_TRY(hipEventElapsedTime(&ms, ctx->ev[4], ctx->ev[5])); st.ms_compact = ms;
  HIP_TRY(hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[5])); st.ms_total = ms;
  for (auto &r : ctx->launches) {
    HIP_TRY(hipEventElapsedTime(&ms, ctx->prof_events[r.ev0], ctx->prof_events[r.ev1]));
    r.info.ms = ms;
  }
  st.nnz_c = nnz_final;
  st.bytes_algorithmic = 12ull * (st.nnz_a + st.nprod + st.nnz_c) + 4ull * (st.rows_a + 1) + 8ull * (rows_c + 1);
  st.temp_bytes = ctx->arena.cap + ctx->temp_cols.cap + ctx->gscratch.cap + ctx->big_cur.cap + ctx->slice_buf.cap;
  return SPGEMM_OK;
}

extern "C" int spgemm_multiply(spgemm_ctx *ctx, const spgemm_mat *A, const spgemm_mat *B, spgemm_mat **out) {
  if (!ctx || !A || !B || !out) return fail(SPGEMM_ERR_INVALID, "NULL argument");
  *out = nullptr;
  TRY(set_device(ctx));
  spgemm_mat *C = new (std::nothrow) spgemm_mat();
  if (!C) return fail(SPGEMM_ERR_NOMEM, "host allocation failed");
  C->rows = A->rows; C->cols = B->cols; C->owns = true; C->device = ctx->device;
  ctx->live_mats.insert(C);
  int rc = pool_alloc(ctx, (A->rows + 1) * 8, (void **)&C->d_ptr, &C->bytes_ptr);
  if (rc == SPGEMM_OK) rc = run_symbolic(ctx, A, B, C->d_ptr);
  uint64_t nnz_s = 0;
  if (rc == SPGEMM_OK) rc = numeric_prepare(ctx, A, C->d_ptr);
  if (rc == SPGEMM_OK) {
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) rc = fail(SPGEMM_ERR_HIP, "reading nnz failed: %s", hipGetErrorString(e));
    else nnz_s = numeric_counts_ready(ctx);
  }
  if (rc == SPGEMM_OK) rc = pool_alloc(ctx, nnz_s * 4, (void **)&C->d_col, &C->bytes_col);
  if (rc == SPGEMM_OK) rc = pool_alloc(ctx, nnz_s * 8, (void **)&C->d_val, &C->bytes_val);
  uint64_t nnz_f = 0;
  if (rc == SPGEMM_OK) rc = run_numeric(ctx, A, B, C->d_ptr, C->d_col, C->d_val, nnz_s, &nnz_f, true, C);
  if (rc == SPGEMM_OK) rc = finish_stats(ctx, C->rows, nnz_f);
  ctx->plan.valid = false;
  if (rc != SPGEMM_OK) {
    std::string keep = g_err;
    spgemm_mat_free(ctx, C);
    g_err = keep;
    return rc;
  }
  C->nnz = nnz_f;
  *out = C;
  return SPGEMM_OK;
}

extern "C" int spgemm_symbolic(spgemm_ctx *ctx, const spgemm_mat *A, const spgemm_mat *B, uint64_t *d_row_ptr,
                               uint64_t *nnz_structural) {
  if (!ctx || !A || !B || !d_row_ptr || !nnz_structural) return fail(SPGEMM_ERR_INVALID, "NULL argument");
  TRY(set_device(ctx));
  TRY(run_symbolic(ctx, A, B, d_row_ptr));
  TRY(numeric_prepare(ctx, A, d_row_ptr));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  *nnz_structural = numeric_counts_ready(ctx);
  return SPGEMM_OK;
}

extern "C" int spgemm_numeric(spgemm_ctx *ctx, const spgemm_mat *A, const spgemm_mat *B, uint64_t *d_row_ptr,
                              uint32_t *d_col_idx, double *d_val, uint64_t *nnz_final) {
  if (!ctx || !A || !B || !d_row_ptr || !nnz_final) return fail(SPGEMM_ERR_INVALID, "NULL argument");
  if (!ctx->plan.valid || ctx->plan.A != A || ctx->plan.B != B)
    return fail(SPGEMM_ERR_STATE, "spgemm_numeric must follow spgemm_symbolic on the same operands");
  if (ctx->plan.nnz_structural && (!d_col_idx || !d_val)) return fail(SPGEMM_ERR_INVALID, "output buffers are NULL");
  TRY(set_device(ctx));
  uint64_t nnz_f = 0;
  TRY(run_numeric(ctx, A, B, d_row_ptr, d_col_idx, d_val, ctx->plan.nnz_structural, &nnz_f, false, nullptr));
  TRY(finish_stats(ctx, A->rows, nnz_f));
  ctx->plan.valid = false;
  *nnz_final = nnz_f;
  return SPGEMM_OK;
}

extern "C" int spgemm_get_launches(spgemm_ctx *ctx, spgemm_launch_info *out, int max_out, int *count) {
  if (!ctx || !count) return fail(SPGEMM_ERR_INVALID, "NULL argument");
  int n = (int)ctx->launches.size();
  *count = n;
  if (out)
    for (int i = 0; i < n && i < max_out; ++i) out[i] = ctx->launches[i].info;
  return SPGEMM_OK;
}

extern "C" int spgemm_get_stats(const spgemm_ctx *ctx, spgemm_stats *out) {
  if (!ctx || !out) return fail(SPGEMM_ERR_INVALID, "NULL argument");
  *out = ctx->stats;
  return SPGEMM_OK;
}

// ---- section bitmap --------------------------------------------------------------------------------
extern "C" int spgemm_section_bitmap(spgemm_ctx *ctx, const spgemm_mat *A, uint32_t n_sections, uint64_t *words_out) {
  if (!ctx || !A || !words_out) return fail(SPGEMM_ERR_INVALID, "NULL argument");
  if (n_sections == 0 || n_sections > MAX_SECTIONS) return fail(SPGEMM_ERR_INVALID, "n_sections must be 1..%u", MAX_SECTIONS);
  uint32_t section_width = (uint32_t)(A->cols / n_sections);
  if (section_width == 0) return fail(SPGEMM_ERR_INVALID, "A.cols / n_sections == 0 (fewer columns than sections)");
  TRY(set_device(ctx));
  uint32_t nwords64 = (n_sections + 63) / 64;
  uint32_t *d_words = ctx->d_small;
  HIP_TRY(hipMemsetAsync(d_words, 0, (size_t)nwords64 * 8, ctx->stream));
  if (A->nnz) {
    uint32_t blocks = (uint32_t)std::min<uint64_t>(div_up_u64(A->nnz, 256 * 8), 2048);
    hipLaunchKernelGGL(k_section_bitmap, dim3(blocks), dim3(256), 0, ctx->stream, A->nnz, A->d_col, section_width,
                       n_sections, d_words);
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(hipMemcpyAsync(words_out, d_words, (size_t)nwords64 * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return SPGEMM_OK;
}

// ---- partitions (host) -------------------------------------------------------------------------------
extern "C" int spgemm_partition_naive(uint32_t height, uint32_t width, int n_parts, spgemm_partition *out) {
  if (!out || n_parts <= 0) return fail(SPGEMM_ERR_INVALID, "bad argument");
  uint32_t P = (uint32_t)n_parts;
  uint32_t rpp = height / P, er = height % P, cpp = width / P, ec = width % P;
  uint32_t cr = 0, cc = 0;
  for (uint32_t i = 0; i < P; ++i) {
    uint32_t tr = rpp + (i < er ? 1u : 0u), tc = cpp + (i < ec ? 1u : 0u);
    out[i].start_row = cr; out[i].end_row = cr + tr;
    out[i].start_col = cc; out[i].end_col = cc + tc;
    cr += tr; cc += tc;
  }
  return SPGEMM_OK;
}

extern "C" int spgemm_partition_balanced(const uint64_t *row_weight, uint32_t height, uint32_t width, int n_parts,
                                         spgemm_partition *out) {
  if (!out || n_parts <= 0 || (height && !row_weight)) return fail(SPGEMM_ERR_INVALID, "bad argument");
  uint32_t P = (uint32_t)n_parts;
  TRY(spgemm_partition_naive(0, width, n_parts, out)); // columns are always split evenly
  unsigned __int128 total = 0;
  for (uint32_t r = 0; r < height; ++r) total += row_weight[r];
  uint64_t per = (uint64_t)(total / P);
  uint32_t node = 0;
  uint64_t run = 0;
  out[0].start_row = 0;
  for (uint32_t r = 0; r < height && node + 1 < P; ++r) {
    run += row_weight[r];
    if (run > per) { // the row that crosses the share opens the next part
      run = 0;
      out[node].end_row = r;
      out[node + 1].start_row = r;
      ++node;
    }
  }
  out[node].end_row = height;
  for (uint32_t j = node + 1; j < P; ++j) out[j].start_row = out[j].end_row = height;
  return SPGEMM_OK;
}

extern "C" int spgemm_row_nprod(spgemm_ctx *ctx, const spgemm_mat *A, const spgemm_mat *B, uint64_t *row_nprod_out) {
  if (!ctx || !A || !B || !row_nprod_out) return fail(SPGEMM_ERR_INVALID, "NULL argument");
  if (A->cols != B->rows) return fail(SPGEMM_ERR_INVALID, "A.cols != B.rows");
  TRY(set_device(ctx));
  const uint32_t rows = (uint32_t)A->rows;
  if (!rows) return SPGEMM_OK;
  TRY(ensure_rowinfo(ctx, B));
  ctx->plan.valid = false;
  size_t need = 3 * align_up((size_t)rows * 4, 256) + 2 * align_up(rows, 256) + align_up((size_t)rows * 8, 256) + 1024 + NPROD_SLOTS * 8;
  const uint32_t max_long = rows;
  need += align_up((size_t)max_long * 4, 256);
  TRY(grow(ctx, ctx->arena, need));
  ArenaCursor ar{reinterpret_cast<char *>(ctx->arena.p), 0, ctx->arena.cap};
  uint32_t *c0 = ar.take<uint32_t>(rows), *c1 = ar.take<uint32_t>(rows), *ub = ar.take<uint32_t>(rows);
  uint8_t *key = ar.take<uint8_t>(rows);
  uint64_t *np = ar.take<uint64_t>(rows);
  uint32_t *long_rows = ar.take<uint32_t>(max_long);
  uint32_t *long_count = ar.take<uint32_t>(64);
  uint64_t *total = ar.take<uint64_t>(NPROD_SLOTS);
  HIP_TRY(hipMemsetAsync(long_count, 0, 256, ctx->stream));
  HIP_TRY(hipMemsetAsync(total, 0, NPROD_SLOTS * 8, ctx->stream));
  uint8_t *same = ar.take<uint8_t>(rows);
  AnalyzeOut ao{c0, c1, ub, key, np, total, long_rows, long_count, same, nullptr, nullptr};
  hipLaunchKernelGGL(k_analyze<16>, dim3(div_up_u64((uint64_t)rows * 16, 256)), dim3(256), 0, ctx->stream, rows,
                     A->d_ptr, A->d_col, rowinfo_of(B), ao, ctx->cp, ctx->long_len);
  hipLaunchKernelGGL(k_analyze_long, dim3(256), dim3(256), 0, ctx->stream, A->d_ptr, A->d_col, rowinfo_of(B), ao, ctx->cp);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(row_nprod_out, np, (size_t)rows * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return SPGEMM_OK;
}

// ---- shuffle (host) -----------------------------------------------------------------------------------
// Fisher-Yates driven by splitmix64, written out so that every build gives the same permutation
// for a seed (std::shuffle's use of the engine is implementation defined).
extern "C" int spgemm_shuffle_random(uint32_t n, uint64_t seed, uint32_t *perm_out) {
  if (n && !perm_out) return fail(SPGEMM_ERR_INVALID, "perm_out is NULL");
  for (uint32_t i = 0; i < n; ++i) perm_out[i] = i;
  uint64_t s = seed;
  auto next = [&s]() {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  };
  for (uint32_t i = n; i > 1; --i) {
    uint32_t j = (uint32_t)(((unsigned __int128)next() * i) >> 64);
    std::swap(perm_out[i - 1], perm_out[j]);
  }
  return SPGEMM_OK;
}

// ---- operand selection --------------------------------------------------------------------------------
extern "C" int spgemm_mat_select(spgemm_ctx *ctx, const spgemm_mat *M, const uint32_t *keep_rows, uint64_t n_keep_rows,
                                 const uint32_t *col_new_id, uint64_t new_cols, int sort_rows, spgemm_mat **out) {
  if (!ctx || !M || !out) return fail(SPGEMM_ERR_INVALID, "NULL argument");
  *out = nullptr;
  TRY(set_device(ctx));
  ctx->plan.valid = false;
  const uint64_t n_out = keep_rows ? n_keep_rows : M->rows;
  const uint64_t cols_out = col_new_id ? new_cols : M->cols;
  TRY(check_dims(n_out, cols_out));
  if (keep_rows)
    for (uint64_t r = 0; r < n_out; ++r)
      if (keep_rows[r] >= M->rows) return fail(SPGEMM_ERR_INVALID, "keep_rows[%llu] out of range", (unsigned long long)r);
  if (col_new_id)
    for (uint64_t c = 0; c < M->cols; ++c)
      if (col_new_id[c] != 0xFFFFFFFFu && col_new_id[c] >= cols_out)
        return fail(SPGEMM_ERR_INVALID, "col_new_id[%llu] out of range", (unsigned long long)c);
  // scratch: keep_rows, col_new_id, row lengths, scan partials
  size_t need = align_up(n_out * 4 + 4, 256) * 2 + align_up(M->cols * 4 + 4, 256) +
                align_up(((size_t)scan_parts(n_out) + 2) * 8, 256);
  TRY(grow(ctx, ctx->arena, need));
  ArenaCursor ar{reinterpret_cast<char *>(ctx->arena.p), 0, ctx->arena.cap};
  uint32_t *d_keep = ar.take<uint32_t>(n_out + 1);
  uint32_t *d_len = ar.take<uint32_t>(n_out + 1);
  uint32_t *d_new = ar.take<uint32_t>(M->cols + 1);
  uint64_t *partial = ar.take<uint64_t>((size_t)scan_parts(n_out) + 2);
  if (keep_rows && n_out) HIP_TRY(hipMemcpyAsync(d_keep, keep_rows, n_out * 4, hipMemcpyHostToDevice, ctx->stream));
  if (col_new_id && M->cols) HIP_TRY(hipMemcpyAsync(d_new, col_new_id, M->cols * 4, hipMemcpyHostToDevice, ctx->stream));
  spgemm_mat *R = new (std::nothrow) spgemm_mat();
  if (!R) return fail(SPGEMM_ERR_NOMEM, "host allocation failed");
  R->rows = n_out; R->cols = cols_out; R->owns = true; R->device = ctx->device;
  int rc = pool_alloc(ctx, (n_out + 1) * 8, (void **)&R->d_ptr, &R->bytes_ptr);
  if (rc != SPGEMM_OK) { delete R; return rc; }
  ctx->live_mats.insert(R);
  const uint32_t *kr = keep_rows ? d_keep : nullptr, *cn = col_new_id ? d_new : nullptr;
  if (n_out) {
    hipLaunchKernelGGL(k_select_count, dim3(div_up_u64(n_out * 8, 256)), dim3(256), 0, ctx->stream, n_out, kr, M->d_ptr,
                       M->d_col, cn, d_len);
    rc = launch_scan(ctx, d_len, n_out, R->d_ptr, partial);
  } else {
    hipError_t e = hipMemsetAsync(R->d_ptr, 0, 8, ctx->stream);
    if (e != hipSuccess) rc = fail(SPGEMM_ERR_HIP, "memset failed");
  }
  uint64_t nnz = 0;
  if (rc == SPGEMM_OK) {
    hipError_t e = hipMemcpyAsync(ctx->h_pinned, R->d_ptr + n_out, 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) rc = fail(SPGEMM_ERR_HIP, "reading nnz failed: %s", hipGetErrorString(e));
    nnz = *reinterpret_cast<uint64_t *>(ctx->h_pinned);
  }
  if (rc == SPGEMM_OK) rc = pool_alloc(ctx, nnz * 4, (void **)&R->d_col, &R->bytes_col);
  if (rc == SPGEMM_OK) rc = pool_alloc(ctx, nnz * 8, (void **)&R->d_val, &R->bytes_val);
  // Rows that have to be sorted by their new column ids are copied into scratch arrays first and
  // sorted from there into the result (segmented radix sort on the device, stable).
  const bool sorting = sort_rows && nnz && n_out;
  uint32_t *t_col = nullptr;
  double *t_val = nullptr;
  size_t t_col_got = 0, t_val_got = 0;
  if (rc == SPGEMM_OK && sorting) rc = pool_alloc(ctx, nnz * 4, (void **)&t_col, &t_col_got);
  if (rc == SPGEMM_OK && sorting) rc = pool_alloc(ctx, nnz * 8, (void **)&t_val, &t_val_got);
  if (rc == SPGEMM_OK && n_out && nnz) {
    hipLaunchKernelGGL(k_select_copy, dim3(div_up_u64(n_out * 64, 256)), dim3(256), 0, ctx->stream, n_out, kr, M->d_ptr,
                       M->d_col, M->d_val, cn, (const uint64_t *)R->d_ptr, sorting ? t_col : R->d_col, sorting ? t_val : R->d_val);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) rc = fail(SPGEMM_ERR_HIP, "select copy failed: %s", hipGetErrorString(e));
  }
  if (rc == SPGEMM_OK && sorting)
    rc = spgemm_internal_sort_rows(ctx, n_out, nnz, cols_out, R->d_ptr, t_col, R->d_col, t_val, R->d_val);
  if (t_col || t_val) {
    (void)hipStreamSynchronize(ctx->stream);
    if (t_col) pool_release(ctx, t_col, t_col_got);
    if (t_val) pool_release(ctx, t_val, t_val_got);
  }
  if (rc != SPGEMM_OK) {
    std::string keep = g_err;
    spgemm_mat_free(ctx, R);
    g_err = keep;
    return rc;
  }
  R->nnz = nnz;
  *out = R;
  return SPGEMM_OK;
}

// ---- B exchange helpers -----------------------------------------------------------------------------------
// Row bounds of section s of a matrix with total_rows rows cut in n_sections
// (reference include/matrix_impl.hpp:628-639).
static void section_rows(uint64_t total_rows, uint32_t n_sections, uint32_t s, uint64_t *r0, uint64_t *r1) {
  uint64_t sh = total_rows / n_sections;
  *r0 = sh * s;
  *r1 = (s == n_sections - 1) ? total_rows : sh * (s + 1);
}

// Host copy of the row pointer and device array of the row lengths, cached on the handle.
static int ensure_section_cache(spgemm_ctx *ctx, const spgemm_mat *M) {
  spgemm_mat *m = const_cast<spgemm_mat *>(M);
  if (m->h_ptr.size() == M->rows + 1 && (m->d_rowlen || M->rows == 0)) return SPGEMM_OK;
  m->h_ptr.resize(M->rows + 1);
  if (M->rows && !m->d_rowlen) TRY(pool_alloc(ctx, M->rows * 4, (void **)&m->d_rowlen, &m->bytes_rowlen));
  if (M->rows) {
    hipLaunchKernelGGL(k_row_len, dim3(div_up_u64(M->rows, 256)), dim3(256), 0, ctx->stream, M->rows, M->d_ptr, m->d_rowlen);
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(hipMemcpyAsync(m->h_ptr.data(), M->d_ptr, (M->rows + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return SPGEMM_OK;
}

extern "C" int spgemm_section_nnz(spgemm_ctx *ctx, const spgemm_mat *M, uint32_t n_sections, uint64_t first_row,
                                  uint64_t total_rows, uint64_t *nnz_out) {
  if (!ctx || !M || !nnz_out || n_sections == 0) return fail(SPGEMM_ERR_INVALID, "bad argument");
  if (first_row + M->rows > total_rows) return fail(SPGEMM_ERR_INVALID, "row block exceeds total_rows");
  if (total_rows / n_sections == 0) return fail(SPGEMM_ERR_INVALID, "total_rows / n_sections == 0");
  TRY(set_device(ctx));
  TRY(ensure_section_cache(ctx, M));
  const std::vector<uint64_t> &ptr = M->h_ptr;
  for (uint32_t s = 0; s < n_sections; ++s) {
    uint64_t r0, r1;
    section_rows(total_rows, n_sections, s, &r0, &r1);
    uint64_t lo = std::min(std::max(r0, first_row), first_row + M->rows) - first_row;
    uint64_t hi = std::min(std::max(r1, first_row), first_row + M->rows) - first_row;
    nnz_out[s] = ptr[hi] - ptr[lo];
  }
  return SPGEMM_OK;
}

extern "C" int spgemm_pack_sections(spgemm_ctx *ctx, const spgemm_mat *M, uint32_t n_sections, uint64_t first_row,
                                    uint64_t total_rows, const uint64_t *bitmap_words, uint32_t *d_row_len_out,
                                    uint32_t *d_col_idx_out, double *d_val_out, uint64_t *rows_out, uint64_t *nnz_out) {
  if (!ctx || !M || !bitmap_words || !rows_out || !nnz_out || n_sections == 0)
    return fail(SPGEMM_ERR_INVALID, "bad argument");
  if (first_row + M->rows > total_rows) return fail(SPGEMM_ERR_INVALID, "row block exceeds total_rows");
  if (total_rows / n_sections == 0) return fail(SPGEMM_ERR_INVALID, "total_rows / n_sections == 0");
  TRY(set_device(ctx));
  // the host copy of the row pointer and the row lengths are cached on the handle: after the first
  // call everything below is queued on the stream and the call returns without waiting for it
  TRY(ensure_section_cache(ctx, M));
  const std::vector<uint64_t> &ptr = M->h_ptr;
  const uint32_t *d_len = M->d_rowlen;
  uint64_t rows_w = 0, nnz_w = 0;
  // adjacent selected sections are copied as one run
  uint32_t s = 0;
  while (s < n_sections) {
    if (!((bitmap_words[s >> 6] >> (s & 63)) & 1ull)) { ++s; continue; }
    uint32_t e = s;
    while (e + 1 < n_sections && ((bitmap_words[(e + 1) >> 6] >> ((e + 1) & 63)) & 1ull)) ++e;
    uint64_t r0, r1, tmp;
    section_rows(total_rows, n_sections, s, &r0, &tmp);
    section_rows(total_rows, n_sections, e, &tmp, &r1);
    uint64_t lo = std::min(std::max(r0, first_row), first_row + M->rows) - first_row;
    uint64_t hi = std::min(std::max(r1, first_row), first_row + M->rows) - first_row;
    if (hi > lo) {
      uint64_t j0 = ptr[lo], j1 = ptr[hi];
      if (d_row_len_out) HIP_TRY(hipMemcpyAsync(d_row_len_out + rows_w, d_len + lo, (hi - lo) * 4, hipMemcpyDeviceToDevice, ctx->stream));
      if (j1 > j0) {
        if (d_col_idx_out) HIP_TRY(hipMemcpyAsync(d_col_idx_out + nnz_w, M->d_col + j0, (j1 - j0) * 4, hipMemcpyDeviceToDevice, ctx->stream));
        if (d_val_out) HIP_TRY(hipMemcpyAsync(d_val_out + nnz_w, M->d_val + j0, (j1 - j0) * 8, hipMemcpyDeviceToDevice, ctx->stream));
      }
      rows_w += hi - lo;
      nnz_w += j1 - j0;
    }
    s = e + 1;
  }
  *rows_out = rows_w;
  *nnz_out = nnz_w;
  return SPGEMM_OK;
}

// Row pointer of a full-height B from the concatenated row lengths of its present sections.
// d_len: rows + 1 uint32 of scratch; partial: scan_parts(rows) + 2 uint64 of scratch.
static int assemble_impl(spgemm_ctx *ctx, uint64_t rows, uint32_t n_sections, const uint64_t *bitmap_words,
                         const uint32_t *d_row_len_concat, uint64_t n_row_len, uint64_t *d_row_ptr_out,
                         uint64_t *nnz_out, uint32_t *d_len, uint64_t *partial) {
  // the sections' rows must add up to n_row_len: checked before anything is queued
  uint64_t src = 0;
  for (uint32_t s = 0; s < n_sections; ++s) {
    if (!((bitmap_words[s >> 6] >> (s & 63)) & 1ull)) continue;
    uint64_t r0, r1;
    section_rows(rows, n_sections, s, &r0, &r1);
    src += r1 - r0;
  }
  if (src > n_row_len) return fail(SPGEMM_ERR_INVALID, "row length buffer shorter than the selected sections");
  if (src < n_row_len) return fail(SPGEMM_ERR_INVALID, "row length buffer longer than the selected sections");
  HIP_TRY(hipMemsetAsync(d_len, 0, rows * 4, ctx->stream));
  src = 0;
  uint32_t s = 0;
  while (s < n_sections) {
    if (!((bitmap_words[s >> 6] >> (s & 63)) & 1ull)) { ++s; continue; }
    uint32_t e = s;
    while (e + 1 < n_sections && ((bitmap_words[(e + 1) >> 6] >> ((e + 1) & 63)) & 1ull)) ++e;
    uint64_t r0, r1, tmp;
    section_rows(rows, n_sections, s, &r0, &tmp);
    section_rows(rows, n_sections, e, &tmp, &r1);
    HIP_TRY(hipMemcpyAsync(d_len + r0, d_row_len_concat + src, (r1 - r0) * 4, hipMemcpyDeviceToDevice, ctx->stream));
    src += r1 - r0;
    s = e + 1;
  }
  TRY(launch_scan(ctx, d_len, rows, d_row_ptr_out, partial));
  if (nnz_out) {
    HIP_TRY(hipMemcpyAsync(ctx->h_pinned, d_row_ptr_out + rows, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    *nnz_out = *reinterpret_cast<uint64_t *>(ctx->h_pinned);
  }
  return SPGEMM_OK;
}

extern "C" int spgemm_assemble_row_ptr(spgemm_ctx *ctx, uint64_t rows, uint32_t n_sections, const uint64_t *bitmap_words,
                                       const uint32_t *d_row_len_concat, uint64_t n_row_len, uint64_t *d_row_ptr_out,
                                       uint64_t *nnz_out) {
  if (!ctx || !bitmap_words || !d_row_ptr_out || n_sections == 0) return fail(SPGEMM_ERR_INVALID, "bad argument");
  if (rows / n_sections == 0) return fail(SPGEMM_ERR_INVALID, "rows / n_sections == 0");
  if (n_row_len && !d_row_len_concat) return fail(SPGEMM_ERR_INVALID, "d_row_len_concat is NULL");
  TRY(check_dims(rows, 0));
  TRY(set_device(ctx));
  ctx->plan.valid = false;
  size_t need = align_up(rows * 4 + 4, 256) + align_up(((size_t)scan_parts(rows) + 2) * 8, 256);
  TRY(grow(ctx, ctx->arena, need));
  ArenaCursor ar{reinterpret_cast<char *>(ctx->arena.p), 0, ctx->arena.cap};
  uint32_t *d_len = ar.take<uint32_t>(rows + 1);
  uint64_t *partial = ar.take<uint64_t>((size_t)scan_parts(rows) + 2);
  return assemble_impl(ctx, rows, n_sections, bitmap_words, d_row_len_concat, n_row_len, d_row_ptr_out, nnz_out, d_len,
                       partial);
}

// The same result as spgemm_assemble_row_ptr without any allocation: the row lengths and the scan
// partials live in this device's copy of `scratch` (address taken once, in spgemm_create).
extern "C" int spgemm_assemble_row_ptr_noalloc(spgemm_ctx *ctx, uint64_t rows, uint32_t n_sections,
                                               const uint64_t *bitmap_words, const uint32_t *d_row_len_concat,
                                               uint64_t n_row_len, uint64_t *d_row_ptr_out, uint64_t *nnz_out) {
  if (!ctx || !bitmap_words || !d_row_ptr_out || n_sections == 0) return fail(SPGEMM_ERR_INVALID, "bad argument");
  if (rows / n_sections == 0) return fail(SPGEMM_ERR_INVALID, "rows / n_sections == 0");
  if (rows > NOALLOC_MAX_ROWS)
    return fail(SPGEMM_ERR_UNSUPPORTED, "spgemm_assemble_row_ptr_noalloc handles at most %llu rows (got %llu); use "
                "spgemm_assemble_row_ptr", (unsigned long long)NOALLOC_MAX_ROWS, (unsigned long long)rows);
  if (!ctx->noalloc_scratch) return fail(SPGEMM_ERR_STATE, "internal: the scratch array has no device address");
  if (n_row_len && !d_row_len_concat) return fail(SPGEMM_ERR_INVALID, "d_row_len_concat is NULL");
  TRY(check_dims(rows, 0));
  TRY(set_device(ctx));
  uint32_t *d_len = reinterpret_cast<uint32_t *>(ctx->noalloc_scratch);
  uint64_t *partial = reinterpret_cast<uint64_t *>(ctx->noalloc_scratch + NOALLOC_LEN_BYTES);
  // The scratch array is this GPU's, not this context's: a call that comes from another stream than
  // the previous one waits, on the device, for the previous call's kernels (no allocation: the
  // event was made when the first context of the device was).  With nnz_out the call returns after
  // its own kernels have finished, so the lock also covers them.
  DeviceShared &ds = g_dev[ctx->device];
  std::lock_guard<std::mutex> lock(ds.mu);
  if (ds.scratch_used && ds.scratch_stream != ctx->stream) HIP_TRY(hipStreamWaitEvent(ctx->stream, ds.scratch_done, 0));
  const int rc = assemble_impl(ctx, rows, n_sections, bitmap_words, d_row_len_concat, n_row_len, d_row_ptr_out, nnz_out, d_len,
                               partial);
  // (recorded after a failed call too: part of its work may have been queued)
  HIP_TRY(hipEventRecord(ds.scratch_done, ctx->stream));
  ds.scratch_stream = ctx->stream;
  ds.scratch_used = true;
  return rc;
}

// ---- raw device buffers -------------------------------------------------------------------------------------
extern "C" int spgemm_buffer_alloc(spgemm_ctx *ctx, size_t bytes, void **d_ptr) {
  if (!ctx || !d_ptr) return fail(SPGEMM_ERR_INVALID, "NULL argument");
  TRY(set_device(ctx));
  void *p = nullptr;
  size_t got = 0;
  TRY(pool_alloc(ctx, bytes, &p, &got));
  // remember the block size on the host side so that free can return it to the pool
  ctx->buffers.push_back({p, got});
  *d_ptr = p;
  return SPGEMM_OK;
}

extern "C" int spgemm_buffer_free(spgemm_ctx *ctx, void *d_ptr) {
  if (!ctx) return fail(SPGEMM_ERR_INVALID, "ctx is NULL");
  if (!d_ptr) return SPGEMM_OK;
  for (size_t i = 0; i < ctx->buffers.size(); ++i)
    if (ctx->buffers[i].p == d_ptr) {
      (void)hipSetDevice(ctx->device);
      (void)hipStreamSynchronize(ctx->stream);
      pool_release(ctx, d_ptr, ctx->buffers[i].bytes);
      ctx->buffers.erase(ctx->buffers.begin() + i);
      return SPGEMM_OK;
    }
  return fail(SPGEMM_ERR_INVALID, "pointer was not allocated by spgemm_buffer_alloc on this context");
}

extern "C" int spgemm_buffer_upload(spgemm_ctx *ctx, void *d_dst, const void *h_src, size_t bytes) {
  if (!ctx || (bytes && (!d_dst || !h_src))) return fail(SPGEMM_ERR_INVALID, "NULL argument");
  if (!bytes) return SPGEMM_OK;
  TRY(set_device(ctx));
  HIP_TRY(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return SPGEMM_OK;
}

extern "C" int spgemm_upload_sections(spgemm_ctx *ctx, uint32_t n, const uint32_t *const *h_row_ptr,
                                      const uint32_t *const *h_col_idx, const void *const *h_val, const uint32_t *rows,
                                      const uint32_t *nnz, uint32_t *d_row_len_out, uint32_t *d_col_idx_out,
                                      double *d_val_out) {
  if (!ctx || (n && (!h_row_ptr || !h_col_idx || !h_val || !rows || !nnz))) return fail(SPGEMM_ERR_INVALID, "NULL argument");
  uint64_t rows_total = 0, nnz_total = 0;
  for (uint32_t i = 0; i < n; ++i) {
    if (!h_row_ptr[i] || (nnz[i] && (!h_col_idx[i] || !h_val[i]))) return fail(SPGEMM_ERR_INVALID, "section %u: NULL array", i);
    if (h_row_ptr[i][0] != 0 || h_row_ptr[i][rows[i]] != nnz[i])
      return fail(SPGEMM_ERR_INVALID, "section %u: its row pointer does not run from 0 to its entry count", i);
    rows_total += rows[i];
    nnz_total += nnz[i];
  }
  if ((rows_total && !d_row_len_out) || (nnz_total && (!d_col_idx_out || !d_val_out))) return fail(SPGEMM_ERR_INVALID, "NULL output buffer");
  TRY(set_device(ctx));
  // the row lengths of all sections in one piece (the blob holds row pointers), then one copy per array and section
  std::vector<uint32_t> len(rows_total);
  uint64_t ro = 0, zo = 0;
  for (uint32_t i = 0; i < n; ++i) {  // every check before the first copy: no return with copies in flight
    for (uint32_t r = 0; r < rows[i]; ++r) {
      if (h_row_ptr[i][r + 1] < h_row_ptr[i][r]) return fail(SPGEMM_ERR_INVALID, "section %u: row pointer decreases at row %u", i, r);
      len[ro + r] = h_row_ptr[i][r + 1] - h_row_ptr[i][r];
    }
    ro += rows[i];
  }
  ro = 0;
  for (uint32_t i = 0; i < n; ++i) {
    if (nnz[i]) {
      HIP_TRY(hipMemcpyAsync(d_col_idx_out + zo, h_col_idx[i], (size_t)nnz[i] * 4, hipMemcpyHostToDevice, ctx->stream));
      HIP_TRY(hipMemcpyAsync(d_val_out + zo, h_val[i], (size_t)nnz[i] * 8, hipMemcpyHostToDevice, ctx->stream));
    }
    ro += rows[i];
    zo += nnz[i];
  }
  if (rows_total) HIP_TRY(hipMemcpyAsync(d_row_len_out, len.data(), rows_total * 4, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return SPGEMM_OK;
}

extern "C" int spgemm_buffer_copy(spgemm_ctx *dst_ctx, void *d_dst, spgemm_ctx *src_ctx, const void *d_src, size_t bytes) {
  if (!dst_ctx || !src_ctx || (bytes && (!d_dst || !d_src))) return fail(SPGEMM_ERR_INVALID, "NULL argument");
  if (!bytes) return SPGEMM_OK;
  HIP_TRY(hipSetDevice(src_ctx->device));
  HIP_TRY(hipStreamSynchronize(src_ctx->stream));
  HIP_TRY(hipSetDevice(dst_ctx->device));
  if (dst_ctx->device == src_ctx->device)
    HIP_TRY(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, dst_ctx->stream));
  else
    HIP_TRY(hipMemcpyPeerAsync(d_dst, dst_ctx->device, d_src, src_ctx->device, bytes, dst_ctx->stream));
  return SPGEMM_OK;
}
