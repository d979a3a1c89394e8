/*
 * spgemm_amd.h - C ABI of the MI355X (gfx950) SpGEMM engine.
 *
 * This is the drop-in boundary for the multiply path of sven-weber/spgemm
 * (the `mults` operators and the partition / bitmap / exchange helpers around
 * them).  Plain pointers and sizes only; every function returns an int status
 * (SPGEMM_OK == 0) and never throws; spgemm_last_error() gives the message of
 * the last failure on the calling thread.  One context per caller thread and
 * per GPU.  There is no CPU fallback: every entry point that computes needs a
 * HIP device and fails with SPGEMM_ERR_HIP when none is usable.
 *
 * Data layout (device, structure of arrays):
 *   row_ptr  uint64_t[rows + 1]   (the reference's midx_t row_ptr is 32 bit,
 *                                  include/matrix.hpp:9; C can exceed 2^32 here)
 *   col_idx  uint32_t[nnz]        (reference midx_t)
 *   val      double[nnz]
 * Rows of an OUTPUT matrix are sorted by column and hold no entry whose
 * accumulated value is exactly 0.0 (the reference erases such entries,
 * include/matrix_impl.hpp:90-93).  Rows of an INPUT matrix may be in any
 * order; products are accumulated in the storage order of the A row, which is
 * the order the reference's loops use (src/mults/drop_parallel.cpp:46-57).
 */
#ifndef SPGEMM_AMD_H
#define SPGEMM_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPGEMM_OK 0
#define SPGEMM_ERR_INVALID 1     /* bad argument / malformed matrix */
#define SPGEMM_ERR_HIP 2         /* a HIP runtime call failed (or no device) */
#define SPGEMM_ERR_NOMEM 3       /* device or host allocation failed */
#define SPGEMM_ERR_STATE 4       /* call made in the wrong state */
#define SPGEMM_ERR_UNSUPPORTED 5 /* size beyond what this build supports */

typedef struct spgemm_ctx spgemm_ctx;
typedef struct spgemm_mat spgemm_mat;

/* reference include/partition.hpp:10-15 (Partition) */
typedef struct spgemm_partition {
  uint32_t start_row;
  uint32_t end_row;
  uint32_t start_col;
  uint32_t end_col;
} spgemm_partition;

/* Per-multiply figures; times are device times between HIP events recorded on
 * the context's stream.  Phase names extend the reference's measure points
 * (include/measure.hpp:16-30: "mult", "bitmaps", ...). */
typedef struct spgemm_stats {
  uint64_t rows_a;
  uint64_t nnz_a;
  uint64_t nnz_b;
  uint64_t nprod;          /* sum over A entries (i,k) of nnz(B[k,:]); flops = 2*nprod */
  uint64_t nnz_structural; /* distinct (i,j) hit by a product */
  uint64_t nnz_c;          /* after dropping exact zeros */
  uint64_t bytes_algorithmic; /* 12*(nnzA+nprod+nnzC) + 4*(rowsA+1) + 8*(rowsC+1) */
  uint64_t temp_bytes;     /* scratch held by the context after this multiply */
  uint32_t rows_window;    /* symbolic passes through an LDS bitmap window (one per row or row group) */
  uint32_t rows_hash;      /* rows accumulated through an LDS hash table */
  uint32_t rows_global;    /* rows accumulated in global-memory scratch */
  uint32_t rows_empty;     /* rows without a symbolic pass: no products, or follower of a row group */
  float ms_analyze;        /* B row info + A row analysis + binning */
  float ms_symbolic;
  float ms_scan;
  float ms_numeric;
  float ms_compact;        /* only when an exact zero had to be dropped */
  float ms_total;          /* first to last event of the multiply */
} spgemm_stats;

/* ---- library ---------------------------------------------------------------------------- */
int spgemm_version(void);
const char *spgemm_last_error(void);
int spgemm_device_count(int *count);

/* ---- context: the per-rank state the reference keeps in a mults::MatrixMultiplication
 *      object (include/mults.hpp:17-33) ----------------------------------------------------- */
int spgemm_create(int device, spgemm_ctx **ctx);
/* Waits for the context's stream, then releases everything the context holds, INCLUDING every
 * matrix handle that was made on it and not freed (spgemm_mat_from_host, _wrap_device, _multiply,
 * _select, _relabel_columns): such handles are dead afterwards and must not be used or freed.  The
 * reference's operator object owns its operands and result the same way (include/mults.hpp:35-50). */
int spgemm_destroy(spgemm_ctx *ctx);
int spgemm_ctx_device(const spgemm_ctx *ctx, int *device);
/* Run on a caller-owned hipStream_t (e.g. the framework's current stream).  NULL restores the
 * context's own stream, which is created non-blocking: it does NOT order with work queued on the
 * device's default (null) stream.  A caller whose producer runs on the default stream passes
 * hipStreamLegacy, not NULL. */
int spgemm_set_stream(spgemm_ctx *ctx, void *hip_stream);
int spgemm_synchronize(spgemm_ctx *ctx);
/* Tunables, by name; unknown names are an error.
 *   window_max_span, window_ratio, hash_max_nprod   row binning (window / LDS hash / big rows);
 *                                                   hash_max_nprod <= 8192 (default 8192: rows of more
 *                                                   than 4096 products only where B's rows are long)
 *   short_max_nprod                                 rows outside the window path with at most this many
 *                                                   products (<= 512) and at most 64 A entries take the
 *                                                   short-row kernels (count-only symbolic pass, numeric
 *                                                   pass that sorts the row's products itself); 0: none
 *   short_rpw                                       rows one wavefront of those kernels pipelines
 *   short_num_rpw                                   ... of the numeric one (default 8; its wavefronts take
 *                                                   rows a launch-wide stride apart)
 *   analyze_lanes                                   lanes per row of A in the analysis: 8, 16, 32, or
 *                                                   0 (default) = by the mean row length
 *   short_key32                                     1 (default): 32-bit entries in the sorting table
 *                                                   of that kernel when B has few enough columns
 *   grp_compact                                     row-group numeric kernel: 1 two-level bitmap over
 *                                                   the column hull (less LDS), 0 flat bitmap (fewer
 *                                                   instructions), 2 (default) chosen per launch
 *   walk_tail                                       window kernels fetch entries 64..127 of a row of B
 *                                                   ahead with its first 64: 0 no, 1 yes, 2 (default)
 *                                                   where B's rows hold 48 entries and more on average
 *   grp_runs                                        1 (default): where consecutive entries of an A row select
 *                                                   consecutive rows of B that have ONE column list (the
 *                                                   unknowns of a mesh node), the window kernels walk the list
 *                                                   once per run: the symbolic one skips the repeats, the
 *                                                   row-group numeric one ranks a column once and sums the
 *                                                   run's products in registers, in the CPU's order; 0: row by row
 *   sym_rpw, grp_rpw                                1..64 (default 8): rows / row groups a wavefront of those two
 *                                                   kernels takes one after the other, the next one's
 *                                                   description in flight (1: one per wavefront)
 *   rpw_fill                                        0..1024 (default 16): ... but no more than leave this many
 *                                                   wavefronts of work per wavefront slot of the device
 *                                                   (0: no such limit, small inputs take several too)
 *   short_packed                                    1: the numeric one reads the rows of B as 16-byte
 *                                                   {column, value} records padded to 128-byte lines
 *                                                   (built once per B handle, at most 7 records of
 *                                                   padding per row); 2 (default): and the symbolic one
 *                                                   their columns from a padded array of its own (rows
 *                                                   start at multiples of 32 bytes); 0: both read B's
 *                                                   arrays
 *   g_thresh0, g_thresh1, g_thresh2                 mean B-row length classes; <= g_thresh2: flat walk
 *   flat_walk, row_groups, big_rows                 0/1 switches of the flat walk, of the grouping of
 *                                                   identical A rows, of the big-row kernels (B sorted)
 *   big_win_words                                   bitmap window of the big-row symbolic kernel
 *                                                   (32-column words, multiple of 1024, <= 32768)
 *   big_sym_bounds                                  1 (default): for a B of at most 16 such windows the big-row
 *                                                   symbolic kernel takes the pieces of B's rows per window
 *                                                   from bounds computed once per B (no cursors, independent
 *                                                   loads); 0: it walks every row from a cursor
 *   big_sym_sorted                                  1 (default): that kernel takes its rows in the order
 *                                                   of their product bound, the dearest first; 0: row order
 *   slice_blocks                                    workgroups of the big-row numeric kernel (0: fill)
 *   slice_queues                                    1..8 work queues of that kernel (workgroup b starts
 *                                                   on queue b mod queues: 8 = one queue per XCD)
 *   slice_run_min, slice_run_div                    chunks per work item of that kernel: at least
 *                                                   run_min (default 64), one more per run_div A entries
 *   slice_sorted                                    its work items are taken 0: row by row; 1: in the order
 *                                                   of their first output column; 2: the dearest first
 *                                                   (slice_cost_w: weight of an A entry in the cost);
 *                                                   3 (default): by the cost's power of two, the dearest
 *                                                   first, and by first output column inside one
 *   slice_packed                                    1: it reads B as 12-byte {column, value} records
 *                                                   (built once per B handle)
 *   poison_output                                   1: C's column and value arrays are filled with 0xFF
 *                                                   bytes before the numeric pass, so that an entry no
 *                                                   kernel writes shows (a test aid; default 0)
 *   lane_order_adds                                 1 (default): where the device passed the probe run
 *                                                   by spgemm_create (one LDS fp64 add applies lanes
 *                                                   that share an address in lane order), the flat
 *                                                   walks add 64 products with one instruction;
 *                                                   0: always resolve such lanes in claim rounds
 *   hash_num_slice                                  LDS-hash rows with more outputs than this are
 *                                                   accumulated by the big-row numeric kernel
 *   global_blocks_sym, global_blocks_num            workgroups of the unsorted-B fallback kernels
 *   profile_launches                                0 off, 1 HIP events per launch, 2 + per-launch sums */
int spgemm_set_option(spgemm_ctx *ctx, const char *name, int64_t value);

/* What spgemm_create found out about the context's device (no counterpart in the reference).
 *   "lane_order_probe"   1 when one LDS fp64 add applied lanes that share an address in ascending lane
 *                        order in every trial of the probe (run once per device and process: 16,384
 *                        workgroup trials, four wavefronts adding at once into one table, strides of
 *                        1..64 entries, powers of two included).  The flat walks then add 64 products
 *                        with one instruction, and their bit parity with the CPU loop RESTS ON THIS
 *                        PROBE: the order is not documented by the hardware manuals.  Option
 *                        lane_order_adds = 0 takes the order-independent path instead.
 *   "cu_count"           compute units of the device */
int spgemm_device_property(const spgemm_ctx *ctx, const char *name, int64_t *value);

/* ---- matrices: replace matrix::ManagedCSRMatrix / CSRMatrix
 *      (include/matrix_impl.hpp:230-384) --------------------------------------------------- */
/* Copy a host CSR to the device. The caller keeps ownership of the host arrays. */
int spgemm_mat_from_host(spgemm_ctx *ctx, uint64_t rows, uint64_t cols, uint64_t nnz,
                         const uint64_t *row_ptr, const uint32_t *col_idx, const double *val,
                         spgemm_mat **out);
/* Wrap device arrays without copying; they must outlive the handle. validate != 0 runs a
 * device pass that checks row_ptr monotonicity and col_idx < cols.
 * The library caches per-row facts of a matrix the first time it is used as B (column range,
 * length, whether the row is sorted).  A caller that rewrites the wrapped arrays in place (the
 * reference refills its receive buffers every gemm, src/mults/drop_parallel.cpp:99-108) must call
 * spgemm_mat_invalidate before the next multiply; the dimensions and nnz of the handle stay fixed. */
int spgemm_mat_wrap_device(spgemm_ctx *ctx, uint64_t rows, uint64_t cols, uint64_t nnz,
                           const uint64_t *d_row_ptr, const uint32_t *d_col_idx,
                           const double *d_val, int validate, spgemm_mat **out);
int spgemm_mat_invalidate(spgemm_ctx *ctx, spgemm_mat *m);
int spgemm_mat_info(const spgemm_mat *m, uint64_t *rows, uint64_t *cols, uint64_t *nnz);
int spgemm_mat_device_ptrs(const spgemm_mat *m, const uint64_t **d_row_ptr,
                           const uint32_t **d_col_idx, const double **d_val);
int spgemm_mat_to_host(spgemm_ctx *ctx, const spgemm_mat *m, uint64_t *row_ptr,
                       uint32_t *col_idx, double *val);
/* Rows [row0, row0 + nrows) only: row_ptr gets nrows + 1 entries as stored (absolute offsets),
 * col_idx / val the rows' entries back to back (capacity `cap` entries; NULL skips the array).
 * What CSRMatrix::row(i) gives the reference's callers (include/matrix_impl.hpp:330-335), for
 * results too large to bring to the host whole. */
int spgemm_mat_rows_to_host(spgemm_ctx *ctx, const spgemm_mat *m, uint64_t row0, uint64_t nrows,
                            uint64_t *row_ptr, uint32_t *col_idx, double *val, uint64_t cap);
/* Invariants of a matrix, checked on the device: fails with SPGEMM_ERR_INVALID when the row pointer
 * is not a monotone 0..nnz sequence or a column index is >= cols; counts the rows whose columns
 * are not strictly increasing and the stored values that are exactly 0.0 (both 0 for every
 * result of spgemm_multiply).  The reference never checks its result's indices
 * (scripts/compare_matrices.py:18-50 compares the values only). */
int spgemm_mat_check(spgemm_ctx *ctx, const spgemm_mat *m, uint64_t *rows_not_increasing,
                     uint64_t *zero_values);
/* Frees a handle made on THIS context (its device arrays too, unless they were wrapped).  NULL is
 * a no-op; a handle that this context does not know (made on another context, or freed already)
 * is refused with SPGEMM_ERR_INVALID and left alone.  Handles still alive when the context is
 * destroyed are freed by spgemm_destroy. */
int spgemm_mat_free(spgemm_ctx *ctx, spgemm_mat *m);

/* ---- the multiply: replaces reset() + the local multiply inside gemm()
 *      (src/mults/outer.cpp:61-73, drop.cpp:76-88, drop_parallel.cpp:43-59,
 *       drop_at_once_parallel.cpp:120-131, baseline.cpp:48-71; accumulator
 *       include/matrix_impl.hpp:81-95; CSR build :275-315) --------------------------------- */
int spgemm_multiply(spgemm_ctx *ctx, const spgemm_mat *A, const spgemm_mat *B, spgemm_mat **C);
/* Staged form for callers that own the output buffers (device pointers).
 * symbolic: fills d_row_ptr[rows_a + 1] with the structural row pointer and returns its last
 *           entry.  numeric: must follow symbolic on the same (A, B) with no other multiply in
 *           between; writes col_idx/val (capacity nnz_structural) and, when exact zeros were
 *           dropped, compacts in place and rewrites d_row_ptr; returns the final nnz. */
int spgemm_symbolic(spgemm_ctx *ctx, const spgemm_mat *A, const spgemm_mat *B,
                    uint64_t *d_row_ptr, uint64_t *nnz_structural);
int spgemm_numeric(spgemm_ctx *ctx, const spgemm_mat *A, const spgemm_mat *B,
                   uint64_t *d_row_ptr, uint32_t *d_col_idx, double *d_val, uint64_t *nnz_final);
int spgemm_get_stats(const spgemm_ctx *ctx, spgemm_stats *out);

/* Per-launch figures of the last multiply, filled only when the option "profile_launches" is
 * 1 (each symbolic / numeric kernel launch is bracketed by HIP events on the context's stream:
 * `ms` is valid) or 2 (additionally the rows each launch worked on are summed up: nnz_a, nprod
 * and nnz_c are valid; this adds small kernels and host syncs to the multiply).  No counterpart in the reference, whose
 * measure points are per phase (include/measure.hpp:16-30). */
typedef struct spgemm_launch_info {
  uint32_t phase;  /* 0 symbolic, 1 numeric */
  uint32_t path;   /* 0 window (LDS bitmap), 1 hash (LDS table), 2 global scratch, 3 window for row groups,
                      4 big rows (symbolic: bitmap windows per workgroup; numeric: output slices),
                      5 window for row groups where B has runs of rows with one column list (a run is
                        ranked once and summed in registers) */
  uint32_t lanes;  /* 64: one A entry per wavefront step; 0: flat walk (products laid end to end);
                      256 / 1024: threads of the workgroup that shares one row */
  uint32_t rows;   /* rows handled by the launch */
  uint32_t param0; /* window: bitmap words; hash: log2 of the table size */
  uint32_t param1; /* window numeric: value slots */
  float ms;        /* device time of the launch */
  uint32_t lds_bytes; /* dynamic LDS per workgroup */
  uint64_t nnz_a;  /* A entries of those rows */
  uint64_t nprod;  /* products of those rows */
  uint64_t nnz_c;  /* structural output entries of those rows */
} spgemm_launch_info;
int spgemm_get_launches(spgemm_ctx *ctx, spgemm_launch_info *out, int max_out, int *count);

/* ---- section bitmap: replaces bitmap::compute_bitmap (src/bitmap.cpp:12-40).
 *      bit s = min(col / (A.cols / n_sections), n_sections - 1); words_out is a HOST array of
 *      ceil(n_sections / 64) words.  Fails with SPGEMM_ERR_INVALID when A.cols / n_sections
 *      == 0 (the reference asserts). --------------------------------------------------------- */
int spgemm_section_bitmap(spgemm_ctx *ctx, const spgemm_mat *A, uint32_t n_sections,
                          uint64_t *words_out);

/* ---- partitions: replace parts::baseline::partition / balanced_partition
 *      (src/parts/baseline.cpp:10-42, :44-86).  Host-side integer work. ------------------- */
int spgemm_partition_naive(uint32_t height, uint32_t width, int n_parts, spgemm_partition *out);
/* Rows are cut where the running sum of row_weight exceeds sum/n_parts (the reference weighs
 * by nnz of the rows of C_sparsity; any per-row weight can be given, e.g. spgemm_row_nprod). */
int spgemm_partition_balanced(const uint64_t *row_weight, uint32_t height, uint32_t width,
                              int n_parts, spgemm_partition *out);
/* Per-row product counts of A*B (HOST array of A.rows entries): a balance weight that does not
 * need the answer's pattern. */
int spgemm_row_nprod(spgemm_ctx *ctx, const spgemm_mat *A, const spgemm_mat *B,
                     uint64_t *row_nprod_out);

/* ---- shuffle: replaces partition::shuffle (src/partition.cpp:15-37), but seeded so runs are
 *      reproducible. perm_out[i] = original index placed at position i. ------------------- */
int spgemm_shuffle_random(uint32_t n, uint64_t seed, uint32_t *perm_out);

/* Iterative locality shuffle: replaces partition::iterative_shuffle and its helpers
 * calculate_avg_indices_row / _col (reference src/partition.cpp:73-202).  C is a device matrix
 * whose PATTERN is used (the reference reads C_sparsity.mtx; here it is typically a product
 * computed by spgemm_multiply).  Iteration i re-ranks the rows (i even) or the columns (i odd) by
 * the mean position of their entries in the other dimension; `iterations` replaces the reference's
 * wall-clock limit.  Means are exact (integer sum, one division in double) and ties go to the
 * lower index, so the result is reproducible.  row_shuffle[i] / col_shuffle[j] = the original row /
 * column at position i / j (what the reference saves as A_shuffle / B_shuffle); host arrays of
 * C.rows / C.cols entries.  row_variance (optional, (iterations + 1) / 2 doubles) receives the
 * variance of (mean - rank) after every row iteration, the figure the reference logs. */
int spgemm_shuffle_iterative(spgemm_ctx *ctx, const spgemm_mat *C, uint32_t iterations, uint32_t *row_shuffle,
                             uint32_t *col_shuffle, double *row_variance);

/* ---- operand selection: what the reference's loader does with keep_rows / keep_cols
 *      (include/matrix_impl.hpp:163-221, src/main.cpp:213-217), on the device.
 *      keep_rows (HOST, may be NULL = all rows in order): output row r is input row
 *      keep_rows[r].  col_new_id (HOST, may be NULL = unchanged): an entry in column c moves to
 *      column col_new_id[c]; entries whose new id is 0xFFFFFFFF are removed.  Output rows are
 *      sorted by new column id when sort_rows != 0, else keep the input order. -------------- */
int spgemm_mat_select(spgemm_ctx *ctx, const spgemm_mat *M, const uint32_t *keep_rows,
                      uint64_t n_keep_rows, const uint32_t *col_new_id, uint64_t new_cols,
                      int sort_rows, spgemm_mat **out);

/* The matrix M with its columns renamed by d_new_id (device array of M.cols entries, a bijection
 * onto [0, new_cols)) and every row sorted by the new ids.  Used to apply the column shuffle
 * (partition::Shuffle B_shuffle, reference src/main.cpp:131,216-217) to the RESULT of a multiply
 * instead of to B: the entries and their values are bit for bit those of multiplying with a
 * relabelled B (every entry still gets its products in the same order), while B keeps the
 * locality the kernels live on.  *out is owned by the library. */
int spgemm_mat_relabel_columns(spgemm_ctx *ctx, const spgemm_mat *M, const uint32_t *d_new_id,
                               uint64_t new_cols, spgemm_mat **out);

/* ---- B exchange helpers: replace BlockedCSRMatrix::filter + the receive-side rebuild
 *      (include/matrix_impl.hpp:507-534, :480-485) for a B that is cut into n_sections row
 *      sections of height rows/n_sections (last one takes the remainder, :628-639).
 *
 *  section_nnz:  HOST array of n_sections entries = stored entries per row section of M.
 *  pack:         gathers the row lengths / col_idx / val of the sections whose bit is set in
 *                bitmap_words (HOST) into caller device buffers, sections in ascending order;
 *                returns rows and entries written.  The copies are QUEUED on the context's stream
 *                (the counts come from a host copy of the row pointer cached on the handle): the
 *                call does not wait for them, later work on the stream or spgemm_synchronize does.
 *  assemble:     with nnz_out == NULL it is queued likewise.
 *  assemble:     builds, on the receiving side, the row pointer of a B that has all `rows` rows
 *                where only the sections set in bitmap_words are present (others empty), from
 *                the concatenated row lengths of the present sections; col_idx/val of that B
 *                are simply the concatenated payloads. --------------------------------------- */
int spgemm_section_nnz(spgemm_ctx *ctx, const spgemm_mat *M, uint32_t n_sections,
                       uint64_t first_row, uint64_t total_rows, uint64_t *nnz_out);
int spgemm_pack_sections(spgemm_ctx *ctx, const spgemm_mat *M, uint32_t n_sections,
                         uint64_t first_row, uint64_t total_rows, const uint64_t *bitmap_words,
                         uint32_t *d_row_len_out, uint32_t *d_col_idx_out, double *d_val_out,
                         uint64_t *rows_out, uint64_t *nnz_out);
int spgemm_assemble_row_ptr(spgemm_ctx *ctx, uint64_t rows, uint32_t n_sections,
                            const uint64_t *bitmap_words, const uint32_t *d_row_len_concat,
                            uint64_t n_row_len, uint64_t *d_row_ptr_out, uint64_t *nnz_out);
/* spgemm_assemble_row_ptr for callers that cannot allocate (code under HIP graph capture, tight
 * serving loops that call it once per exchange step, harnesses that forbid hipMalloc): same
 * arguments, same result, but no hipMalloc / hipFree or allocating library routine is called.  It
 * has no counterpart in the reference.  Its scratch (row lengths and scan partials, 16 MiB) is a
 * static device array of the library: at most 4,194,304 rows (SPGEMM_ERR_UNSUPPORTED beyond).
 * Every call on a GPU shares that GPU's array.  Calls made through this library in one process are
 * ordered on the device (a call from another stream or context waits for the previous call's
 * kernels); calls from different PROCESSES on one GPU each have their own copy of the array. */
int spgemm_assemble_row_ptr_noalloc(spgemm_ctx *ctx, uint64_t rows, uint32_t n_sections,
                                    const uint64_t *bitmap_words, const uint32_t *d_row_len_concat,
                                    uint64_t n_row_len, uint64_t *d_row_ptr_out, uint64_t *nnz_out);

/* ---- raw device buffers for the exchange: what the reference keeps in std::vector<std::byte>
 *      receive buffers (src/mults/drop_parallel.cpp:65-68) lives in HBM here.  copy moves bytes
 *      between two contexts' devices (peer-to-peer over xGMI when they differ; the analogue of
 *      communication::send / recv, src/communication.cpp:11-22) and is ordered after the work
 *      already queued on the source context and before later work of the destination context. */
int spgemm_buffer_alloc(spgemm_ctx *ctx, size_t bytes, void **d_ptr);
int spgemm_buffer_free(spgemm_ctx *ctx, void *d_ptr);
int spgemm_buffer_copy(spgemm_ctx *dst_ctx, void *d_dst, spgemm_ctx *src_ctx, const void *d_src, size_t bytes);
/* Host bytes into a device buffer of the context (set-up data such as a column permutation;
 * returns when the copy is done). */
int spgemm_buffer_upload(spgemm_ctx *ctx, void *d_dst, const void *h_src, size_t bytes);
/* The receive side of the reference's exchange for a B that stays on the device: the n CSR
 * sections of a received blocked blob (BlockedCSRMatrix(const Data &), include/matrix_impl.hpp:
 * 428-439, 480-485: per section a row pointer of rows[i] + 1 uint32 that starts at 0, nnz[i]
 * columns, nnz[i] values; the values may be aligned to 4 bytes only) go to the device in the layout
 * spgemm_pack_sections produces and spgemm_assemble_row_ptr reads: row lengths, columns and values
 * of the sections one after the other.  All pointers but the d_* ones are HOST pointers; the
 * sections come in ascending order.  Returns when the host arrays may be reused. */
int spgemm_upload_sections(spgemm_ctx *ctx, uint32_t n, const uint32_t *const *h_row_ptr,
                           const uint32_t *const *h_col_idx, const void *const *h_val,
                           const uint32_t *rows, const uint32_t *nnz, uint32_t *d_row_len_out,
                           uint32_t *d_col_idx_out, double *d_val_out);

#ifdef __cplusplus
}
#endif
#endif /* SPGEMM_AMD_H */
