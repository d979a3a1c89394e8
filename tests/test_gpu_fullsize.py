"""Full-size checks (the sizes bench.py runs): every invariant of the result is checked on the
device, and a sample of rows - random ones, the rows with the most products, the last rows (beyond
entry 2^32 of the R-MAT result) - is compared bit for bit with the CPU oracle run on those rows."""
import numpy as np
import pytest

from oracle import oracle as orc
from spgemm_amd import capi, hostcsr

pytestmark = pytest.mark.gpu


def _verify(A, n_random=1000, n_heavy=3, n_tail=40, seed=5, expect_u64=False):
    ctx = capi.Context(0)
    try:
        ctx.set_option("poison_output", 1)  # an entry no kernel writes: column 2^32 - 1, caught by check() below
        dA = ctx.from_host(A.rows, A.cols, A.row_ptr, A.col_idx, A.val)
        dC = ctx.multiply(dA, dA)
        st = ctx.stats()
        rows, cols, nnz = dC.info()
        assert (rows, cols) == (A.rows, A.cols) and nnz == st["nnz_c"]
        # device-side invariants: monotone row pointer ending at nnz, columns in range, rows strictly
        # increasing (sorted, duplicate free), no stored zero
        assert dC.check() == (0, 0)
        # products: the engine's count equals the sum over A entries of the B row lengths
        lens = np.diff(A.row_ptr.astype(np.int64))
        per_row = np.add.reduceat(np.concatenate([lens[A.col_idx.astype(np.int64)], [0]]),
                                  np.minimum(A.row_ptr[:-1].astype(np.int64), A.nnz))
        per_row[lens == 0] = 0
        assert st["nprod"] == int(per_row.sum())
        if expect_u64:
            assert nnz > 2 ** 32
        rng = np.random.default_rng(seed)
        nonempty = np.flatnonzero(lens > 0)
        sample = set(rng.choice(nonempty, size=min(n_random, len(nonempty)), replace=False).tolist())
        sample.update(np.argsort(per_row)[-n_heavy:].tolist())
        sample.update(nonempty[-n_tail:].tolist())
        sample = np.array(sorted(sample), dtype=np.int64)
        full = orc.OCsr(A.rows, A.cols, A.row_ptr, A.col_idx, A.val)
        E = orc.spgemm(full.select_rows(sample), full)
        e_ptr = E.row_ptr.astype(np.int64)
        for i, r in enumerate(sample):
            rp, ci, v = dC.rows_to_host(int(r), 1)
            lo, hi = e_ptr[i], e_ptr[i + 1]
            assert int(rp[1]) == hi - lo, f"row {r}: {int(rp[1])} entries, oracle {hi - lo}"
            assert np.array_equal(ci, E.col_idx[lo:hi]), f"row {r}: columns differ"
            assert np.array_equal(v.view(np.uint64), E.val[lo:hi].view(np.uint64)), f"row {r}: values differ"
        if expect_u64:
            full_ptr = np.empty(2, dtype=np.uint64)
            capi._check(capi.load().spgemm_mat_rows_to_host(ctx.handle, dC.handle, int(sample[-1]), 1,
                                                            capi._np_ptr(full_ptr), None, None, 2 ** 63))
            assert int(full_ptr[0]) > 2 ** 32, "the sample must reach beyond entry 2^32"
        dC.free()
        dA.free()
        return st
    finally:
        ctx.close()


def test_full_size_shell():
    st = _verify(hostcsr.shell_like())
    assert st["rows_a"] == 1508065


def test_full_size_diel():
    _verify(hostcsr.diel_like(), n_random=500)


def test_full_size_erdos_renyi_2_22():
    st = _verify(hostcsr.erdos_renyi(1 << 22, 16, seed=42))
    assert st["nnz_c"] > 10 ** 9


def test_full_size_rmat_scale_22_u64_offsets():
    """nnz(C) = 8.27e9: row pointer entries beyond 2^32; the sampled tail rows live there."""
    _verify(hostcsr.rmat(22, 4, seed=1), n_random=600, n_heavy=2, expect_u64=True)


def test_full_size_erdos_renyi_2_22_shuffle_variant():
    """The `shuffle` variant (bitmap-filtered exchange + random row and column shuffle, B relabelled
    on the device before the exchange) at n = 2^22, a world of one rank: the rank's rows of C are the
    oracle's rows of A * A, in shuffled row order and shuffled column ids, on a sample of rows."""
    import os
    import socket
    import torch
    import torch.distributed as dist
    from spgemm_amd import dist as sd
    A = hostcsr.erdos_renyi(1 << 22, 16, seed=42)
    started = not dist.is_initialized()
    if started:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        op = sd.DistSpgemm(A, None, impl="shuffle", n_sections=512, seed=7, device_index=0)
        try:
            assert not op.late_relabel                      # a B without column locality is relabelled up front
            C = op.gemm()
            rows, cols, nnz = C.info()
            assert (rows, cols) == (A.rows, A.cols) and nnz > 10 ** 9
            assert C.check() == (0, 0)                      # sorted rows, columns in range, no stored zero
            inv = np.empty(A.cols, dtype=np.int64)
            inv[op.B_shuffle] = np.arange(A.cols, dtype=np.int64)
            rng = np.random.default_rng(3)
            sample = np.sort(rng.choice(A.rows, size=600, replace=False))
            full = orc.OCsr(A.rows, A.cols, A.row_ptr, A.col_idx, A.val)
            E = orc.spgemm(full.select_rows(op.A_shuffle[sample]), full)
            e_ptr = E.row_ptr.astype(np.int64)
            for i, r in enumerate(sample):
                rp, ci, v = C.rows_to_host(int(r), 1)
                lo, hi = e_ptr[i], e_ptr[i + 1]
                want_c = inv[E.col_idx[lo:hi].astype(np.int64)]
                order = np.argsort(want_c, kind="stable")
                assert int(rp[1]) == hi - lo, f"row {r}: {int(rp[1])} entries, oracle {hi - lo}"
                assert np.array_equal(ci.astype(np.int64), want_c[order]), f"row {r}: columns differ"
                assert np.array_equal(v.view(np.uint64), E.val[lo:hi][order].view(np.uint64)), f"row {r}: values differ"
        finally:
            op.close()
    finally:
        if started:
            dist.destroy_process_group()
