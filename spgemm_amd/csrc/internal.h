// Hooks between the translation units of the library (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/spgemm_amd.h"

extern "C" {
// the stream the context currently runs on (makes the context's device current)
int spgemm_internal_stream(spgemm_ctx *ctx, hipStream_t *stream);
// records the message returned by spgemm_last_error() on this thread and returns `code`
int spgemm_internal_fail(int code, const char *message);
// a new matrix owned by the library, its three arrays taken from the context's buffer pool and
// left uninitialised (freed with spgemm_mat_free)
int spgemm_internal_alloc_mat(spgemm_ctx *ctx, uint64_t rows, uint64_t cols, uint64_t nnz, spgemm_mat **out);
// a block from the context's buffer pool / back to it (the stream must have drained the block's users)
int spgemm_internal_pool_alloc(spgemm_ctx *ctx, size_t bytes, void **p, size_t *got);
void spgemm_internal_pool_release(spgemm_ctx *ctx, void *p, size_t got);
// sorts n pairs of (32-bit key, 12-byte record) by key on `stream`.  tmp == nullptr: only *tmp_bytes
// (the scratch the sort needs) is set.
int spgemm_internal_sort_records12(hipStream_t stream, const uint32_t *key_in, uint32_t *key_out, const void *rec_in,
                                   void *rec_out, size_t n, void *tmp, size_t *tmp_bytes);
// every row of a CSR structure (row pointer m_ptr) sorted by the 32-bit keys of its entries (all below
// key_limit), stable, values moving along: key_in / val_in -> key_out / val_out; drains the stream
int spgemm_internal_sort_rows(spgemm_ctx *ctx, uint64_t rows, uint64_t nnz, uint64_t key_limit, const uint64_t *m_ptr,
                              const uint32_t *key_in, uint32_t *key_out, const double *val_in, double *val_out);
}
