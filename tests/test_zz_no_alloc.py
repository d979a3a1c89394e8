"""GPU test of the allocation-free entry point spgemm_assemble_row_ptr_noalloc: same row pointer and
nnz as spgemm_assemble_row_ptr (bit for bit: integer work), on two calls in a row, the second with a
smaller input, through the C ABI.  Its scratch is a per-GPU static array that keeps the previous
call's contents, which is what the second call checks."""
import numpy as np
import pytest

from spgemm_amd import capi
from spgemm_amd import dist as sd

pytestmark = pytest.mark.gpu


def _case(ctx, torch, total, N, bits, seed):
    rng = np.random.default_rng(seed)
    sb = sd.section_bounds(total, N)
    rows = np.concatenate([np.arange(sb[s], sb[s + 1]) for s in np.flatnonzero(bits)] or [np.zeros(0, np.int64)])
    lens = rng.integers(0, 40, size=len(rows)).astype(np.int32)
    t_len = torch.from_numpy(lens).cuda()
    want_ptr = torch.full((total + 1,), -1, dtype=torch.int64, device="cuda")
    got_ptr = torch.full((total + 1,), -2, dtype=torch.int64, device="cuda")
    nnz_want = ctx.assemble_row_ptr(total, N, bits, t_len.data_ptr(), len(lens), want_ptr.data_ptr())
    nnz_got = ctx.assemble_row_ptr(total, N, bits, t_len.data_ptr(), len(lens), got_ptr.data_ptr(), noalloc=True)
    full = np.zeros(total, dtype=np.int64)
    full[rows] = lens
    assert nnz_want == int(full.sum())
    assert nnz_got == nnz_want
    assert np.array_equal(got_ptr.cpu().numpy(), want_ptr.cpu().numpy())
    assert np.array_equal(got_ptr.cpu().numpy(), np.concatenate([[0], np.cumsum(full)]))


def test_noalloc_assemble_matches_allocating_entry_point():
    import torch
    ctx = capi.Context(0)
    try:
        N = 64
        bits = np.zeros(N, dtype=bool)
        bits[[0, 1, 2, 17, 40, 41, 63]] = True
        _case(ctx, torch, 300_001, N, bits, seed=1)       # several scan blocks, a ragged last section
        bits2 = np.zeros(N, dtype=bool)
        bits2[[5, 6, 30]] = True
        _case(ctx, torch, 9_000, N, bits2, seed=2)        # smaller: the scratch still holds the first call's data
        _case(ctx, torch, 64, N, np.ones(N, dtype=bool), seed=3)
        # sizes beyond the scratch array are refused, and so are inconsistent lengths
        t = torch.zeros(8, dtype=torch.int32, device="cuda")
        p = torch.zeros(8, dtype=torch.int64, device="cuda")
        with pytest.raises(capi.SpgemmError) as ei:
            ctx.assemble_row_ptr((1 << 22) + 64, N, bits2, t.data_ptr(), 8, p.data_ptr(), noalloc=True)
        assert ei.value.code == 5 and "at most" in str(ei.value)
        with pytest.raises(capi.SpgemmError):
            ctx.assemble_row_ptr(9_000, N, bits2, t.data_ptr(), 7, p.data_ptr(), noalloc=True)
    finally:
        ctx.close()
