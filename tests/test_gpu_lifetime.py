"""Lifetime rules of the handles behind the C ABI (include/spgemm_amd.h): a matrix belongs to the
context that made it; freeing it twice or through another context is refused; a context that is
destroyed takes the matrices still alive on it along (their device memory comes back); the
allocation-free row-pointer call is safe from two contexts of one GPU."""
import ctypes

import numpy as np
import pytest

from spgemm_amd import capi, hostcsr
from spgemm_amd import dist as sd

pytestmark = pytest.mark.gpu


def test_second_free_and_foreign_free_are_refused():
    A = hostcsr.erdos_renyi(2000, 8, seed=3)
    c1, c2 = capi.Context(0), capi.Context(0)
    try:
        dA = c1.from_host(A.rows, A.cols, A.row_ptr, A.col_idx, A.val)
        lib = capi.load()
        raw = ctypes.c_void_p(dA.handle.value)
        assert lib.spgemm_mat_free(c2.handle, raw) == 1          # another context's handle: refused, left alone
        assert "does not belong" in lib.spgemm_last_error().decode()
        dC = c1.multiply(dA, dA)                                   # ... and still usable
        assert dC.nnz > 0
        assert lib.spgemm_mat_free(c1.handle, raw) == 0
        assert lib.spgemm_mat_free(c1.handle, raw) == 1          # freed already
        dA.handle = ctypes.c_void_p(None)
        assert lib.spgemm_mat_free(c1.handle, None) == 0          # NULL is a no-op
        dC.free()
    finally:
        c1.close()
        c2.close()


def test_destroy_releases_the_matrices_left_on_the_context():
    import torch
    A = hostcsr.erdos_renyi(1 << 20, 16, seed=4)                  # 200 MB of arrays on the device
    free0 = None
    for i in range(12):
        ctx = capi.Context(0)
        dA = ctx.from_host(A.rows, A.cols, A.row_ptr, A.col_idx, A.val)
        dC = ctx.multiply(dA, dA)                                 # result, row info, padded records: all left alive
        assert dC.nnz > 0
        ctx.close()                                               # no free of dA / dC before it
        dA.handle = dC.handle = ctypes.c_void_p(None)             # dead handles: never touched again
        torch.cuda.synchronize()
        free_now = torch.cuda.mem_get_info(0)[0]
        if i == 1:
            free0 = free_now
        if i > 1:   # nothing accumulates from one context to the next (a leak would be > 2 GB per context)
            assert free_now > free0 - (256 << 20), f"context {i}: {free0 - free_now} bytes fewer free than after context 1"


def test_noalloc_assemble_from_two_contexts_of_one_gpu():
    """Both contexts use the GPU's one scratch array from streams of their own; the calls are only
    queued (no read-back), so without the library's ordering the second would overwrite the row
    lengths the first is still scanning."""
    import torch
    c1, c2 = capi.Context(0), capi.Context(0)
    try:
        N, total = 64, 3_000_001
        sb = sd.section_bounds(total, N)
        cases = []
        for k, ctx in enumerate((c1, c2, c1, c2, c2, c1)):
            rng = np.random.default_rng(10 + k)
            bits = rng.random(N) < 0.6
            bits[k] = True
            rows = np.concatenate([np.arange(sb[s], sb[s + 1]) for s in np.flatnonzero(bits)])
            lens = rng.integers(0, 50, size=len(rows)).astype(np.int32)
            t_len = torch.from_numpy(lens).cuda()
            out = torch.full((total + 1,), -1, dtype=torch.int64, device="cuda")
            cases.append((ctx, bits, rows, lens, t_len, out))
        torch.cuda.synchronize()
        for ctx, bits, rows, lens, t_len, out in cases:           # all queued back to back
            ctx.assemble_row_ptr(total, N, bits, t_len.data_ptr(), len(lens), out.data_ptr(), noalloc=True, want_nnz=False)
        c1.synchronize()
        c2.synchronize()
        for ctx, bits, rows, lens, t_len, out in cases:
            full = np.zeros(total, dtype=np.int64)
            full[rows] = lens
            assert np.array_equal(out.cpu().numpy(), np.concatenate([[0], np.cumsum(full)]))
    finally:
        c1.close()
        c2.close()
