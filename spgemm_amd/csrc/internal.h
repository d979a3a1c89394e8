// Hooks between the translation units of the library (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/spgemm_amd.h"

extern "C" {
// the stream the context currently runs on (makes the context's device current)
int spgemm_internal_stream(spgemm_ctx *ctx, hipStream_t *stream);
// records the message returned by spgemm_last_error() on this thread and returns `code`
int spgemm_internal_fail(int code, const char *message);
}
