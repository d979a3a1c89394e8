"""GPU tests of the distributed operator (world size 1 over RCCL: the exchange is a self-send, the
whole gemm path runs) and of the C-ABI helpers around the multiply."""
import os
import socket

import numpy as np
import pytest

from oracle import oracle as orc
from spgemm_amd import capi, hostcsr
from spgemm_amd import dist as sd

pytestmark = pytest.mark.gpu


def to_o(h):
    return orc.OCsr(h.rows, h.cols, h.row_ptr, h.col_idx, h.val)


@pytest.fixture(scope="module")
def pg():
    import torch
    import torch.distributed as dist
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield dist
    dist.destroy_process_group()


@pytest.mark.parametrize("impl", ["baseline", "drop", "shuffle", "shuffle:b", "shuffle:result"])
def test_dist_operator_world1(pg, impl):
    # shuffle:b / shuffle:result force where the column shuffle is applied (to B before the exchange,
    # or to the result of the multiply); left alone the operator decides from the operands
    impl, _, relabel = impl.partition(":")
    A = hostcsr.banded_random(6000, 8, 50, seed=21)
    op = sd.DistSpgemm(A, None, impl=impl, n_sections=512, seed=5, device_index=0, relabel=relabel or None)
    if relabel:
        assert op.late_relabel == (relabel == "result")
    op.gemm()
    op.gemm()  # a second run replaces the first result (reset + gemm)
    C = op.result()
    M = sd.merge_unshuffle([C], op.partitions, op.A_shuffle, op.B_shuffle, A.rows, A.cols)
    orc.assert_same(to_o(M), orc.spgemm(to_o(A), to_o(A)), rtol=0.0)
    if impl == "drop":
        assert np.array_equal(op.bitmap, orc.section_bitmap(to_o(op.ops.A_part), 512))
    op.close()


def test_section_helpers_match_numpy():
    ctx = capi.Context(0)
    import torch
    A = hostcsr.erdos_renyi(5000, 6, seed=8)
    N, total = 16, 5000
    first, last = 1200, 3900  # this block holds rows [1200, 3900) of a 5000-row matrix
    blk = sd.select_rows(A, np.arange(first, last))
    d = ctx.from_host(blk.rows, blk.cols, blk.row_ptr, blk.col_idx, blk.val)
    sb = sd.section_bounds(total, N)
    rp = blk.row_ptr.astype(np.int64)
    lo = np.clip(sb[:-1], first, last) - first
    hi = np.clip(sb[1:], first, last) - first
    assert np.array_equal(ctx.section_nnz(d, N, first, total).astype(np.int64), rp[hi] - rp[lo])
    bits = np.zeros(N, dtype=bool)
    bits[[3, 4, 5, 9, 12]] = True
    rows_want = int(sum(hi[s] - lo[s] for s in np.flatnonzero(bits)))
    nnz_want = int(sum(rp[hi[s]] - rp[lo[s]] for s in np.flatnonzero(bits)))
    t_len = torch.empty(rows_want, dtype=torch.int32, device="cuda")
    t_col = torch.empty(nnz_want, dtype=torch.int32, device="cuda")
    t_val = torch.empty(nnz_want, dtype=torch.float64, device="cuda")
    r, z = ctx.pack_sections(d, N, first, total, bits, t_len.data_ptr(), t_col.data_ptr(), t_val.data_ptr())
    assert (r, z) == (rows_want, nnz_want)
    ctx.synchronize()  # pack_sections only queues its copies on the context's stream
    keep_rows = np.concatenate([np.arange(lo[s], hi[s]) for s in np.flatnonzero(bits)])
    sel = sd.select_rows(blk, keep_rows)
    assert np.array_equal(t_len.cpu().numpy(), np.diff(sel.row_ptr.astype(np.int64)))
    assert np.array_equal(t_col.cpu().numpy().view(np.uint32), sel.col_idx)
    assert np.array_equal(t_val.cpu().numpy(), sel.val)
    # assemble: a 5000-row B where only sections 3,4,5 (fully inside the block) are present
    bits2 = np.zeros(N, dtype=bool)
    bits2[[4, 5, 6]] = True
    rows2 = np.concatenate([np.arange(sb[s], sb[s + 1]) for s in (4, 5, 6)])
    lens2 = np.diff(A.row_ptr.astype(np.int64))[rows2].astype(np.int32)
    t_len2 = torch.from_numpy(lens2).cuda()
    t_ptr = torch.empty(total + 1, dtype=torch.int64, device="cuda")
    nnz = ctx.assemble_row_ptr(total, N, bits2, t_len2.data_ptr(), len(lens2), t_ptr.data_ptr())
    full = np.zeros(total, dtype=np.int64)
    full[rows2] = lens2
    assert nnz == int(full.sum())
    assert np.array_equal(t_ptr.cpu().numpy(), np.concatenate([[0], np.cumsum(full)]))
    with pytest.raises(capi.SpgemmError):
        ctx.assemble_row_ptr(total, N, bits2, t_len2.data_ptr(), len(lens2) - 1, t_ptr.data_ptr())
    d.free()
    ctx.close()


def test_select_and_row_nprod():
    ctx = capi.Context(0)
    A = hostcsr.rmat(11, 6, seed=4)
    d = ctx.from_host(A.rows, A.cols, A.row_ptr, A.col_idx, A.val)
    # products per row
    lens = np.diff(A.row_ptr.astype(np.int64))
    want = np.add.reduceat(np.concatenate([lens[A.col_idx.astype(np.int64)], [0]]),
                           np.minimum(A.row_ptr[:-1].astype(np.int64), A.nnz))
    want[lens == 0] = 0
    assert np.array_equal(ctx.row_nprod(d, d).astype(np.int64), want)
    # keep_rows + column relabel (a random permutation), rows re-sorted
    perm = capi.shuffle_random(A.rows, 3)
    inv = np.empty(A.cols, dtype=np.uint32)
    inv[capi.shuffle_random(A.cols, 4)] = np.arange(A.cols, dtype=np.uint32)
    s = ctx.select(d, keep_rows=perm[:700], col_new_id=inv, new_cols=A.cols, sort_rows=True)
    rp, ci, v = s.to_host()
    want_m = sd.relabel_cols(sd.select_rows(A, perm[:700].astype(np.int64)), inv.astype(np.int64))
    assert np.array_equal(rp, want_m.row_ptr) and np.array_equal(ci, want_m.col_idx) and np.array_equal(v, want_m.val)
    # dropping columns: keep the first half only
    half = np.where(np.arange(A.cols) < A.cols // 2, np.arange(A.cols), 0xFFFFFFFF).astype(np.uint32)
    s2 = ctx.select(d, col_new_id=half, new_cols=A.cols // 2, sort_rows=False)
    rp2, ci2, _ = s2.to_host()
    keep = A.col_idx < A.cols // 2
    assert np.array_equal(ci2, A.col_idx[keep])
    assert rp2[-1] == keep.sum()
    for h in (s, s2, d):
        h.free()
    ctx.close()


def test_staged_api_with_torch_buffers():
    import torch
    ctx = capi.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    A = hostcsr.banded_random(4000, 7, 35, seed=9)
    d = ctx.from_host(A.rows, A.cols, A.row_ptr, A.col_idx, A.val)
    row_ptr = torch.empty(A.rows + 1, dtype=torch.int64, device="cuda")
    nnz = ctx.symbolic(d, d, row_ptr.data_ptr())
    col = torch.empty(nnz, dtype=torch.int32, device="cuda")
    val = torch.empty(nnz, dtype=torch.float64, device="cuda")
    nnz_f = ctx.numeric(d, d, row_ptr.data_ptr(), col.data_ptr(), val.data_ptr())
    E = orc.spgemm(to_o(A), to_o(A))
    assert nnz_f == E.nnz
    C = orc.OCsr(A.rows, A.cols, row_ptr.cpu().numpy().astype(np.uint64), col.cpu().numpy().view(np.uint32)[:nnz_f],
                 val.cpu().numpy()[:nnz_f])
    orc.assert_same(C, E, rtol=0.0)
    with pytest.raises(capi.SpgemmError):  # numeric without a fresh symbolic
        ctx.numeric(d, d, row_ptr.data_ptr(), col.data_ptr(), val.data_ptr())
    d.free()
    ctx.close()


@pytest.mark.parametrize("world", [2, 4])
def test_dist_operator_ranks_share_one_gpu(world):
    """The whole N > 1 operator path (device bitmap, exchange, HIP multiply per rank, merge) with
    `world` processes on this one GPU; collectives over gloo, staged through host memory."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "2"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(root, "tests", "_dist_gpu_worker.py")]
    p = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert p.stdout.count("DIST_GPU_OK") == 4, p.stdout[-2000:]


def _shuffled_band(n, seed):
    """A banded matrix under ONE random permutation of rows and columns (P A P^T)."""
    A = hostcsr.banded_random(n, 6, 30, seed=seed)
    perm = np.random.default_rng(seed).permutation(n)
    inv = np.empty(n, dtype=np.int64)
    inv[perm] = np.arange(n)
    return sd.relabel_cols(sd.select_rows(A, perm), inv)


def _spread(C, rs, cs):
    """mean |row position - column position| over the entries of C under the two shuffles"""
    pos_r = np.empty(C.rows, dtype=np.int64)
    pos_r[rs] = np.arange(C.rows)
    pos_c = np.empty(C.cols, dtype=np.int64)
    pos_c[cs] = np.arange(C.cols)
    return float(np.abs(pos_r[C.row_ids()] - pos_c[C.col_idx.astype(np.int64)]).mean())


@pytest.mark.parametrize("case", ["band", "rmat", "rect"])
def test_iterative_shuffle_equals_oracle(case):
    ctx = capi.Context(0)
    if case == "band":
        A = B = _shuffled_band(4000, 5)
    elif case == "rmat":
        A = B = hostcsr.rmat(11, 6, seed=2)
    else:  # rectangular product with empty rows and columns in C
        A = hostcsr.HostCsr.from_coo(700, 500, *np.random.default_rng(3).integers(0, [[700], [500]], (2, 1500)),
                                     np.ones(1500), sum_duplicates=True).sorted_rows()
        B = hostcsr.HostCsr.from_coo(500, 900, *np.random.default_rng(4).integers(0, [[500], [900]], (2, 1200)),
                                     np.ones(1200), sum_duplicates=True).sorted_rows()
    dA = ctx.from_host(A.rows, A.cols, A.row_ptr, A.col_idx, A.val)
    dB = dA if B is A else ctx.from_host(B.rows, B.cols, B.row_ptr, B.col_idx, B.val)
    dC = ctx.multiply(dA, dB)
    E = orc.spgemm(to_o(A), to_o(B))
    for iters in (0, 1, 2, 7):
        rs, cs, var = ctx.shuffle_iterative(dC, iters)
        ers, ecs, evar = orc.iterative_shuffle(E, iters)
        assert np.array_equal(rs, ers) and np.array_equal(cs, ecs), (case, iters)
        assert np.allclose(var, evar, rtol=1e-12, atol=0.0)
        assert np.array_equal(np.sort(rs), np.arange(A.rows)) and np.array_equal(np.sort(cs), np.arange(B.cols))
    if case == "band":  # the shuffle pulls C back towards a band
        Ch = hostcsr.HostCsr(E.rows, E.cols, E.row_ptr, E.col_idx, E.val)
        rs, cs, var = ctx.shuffle_iterative(dC, 40)
        ident = np.arange(A.rows)
        assert _spread(Ch, rs, cs) < 0.25 * _spread(Ch, ident, ident)
        assert var[-1] < var[0]
    for h in (dC, dA) + (() if dB is dA else (dB,)):
        h.free()
    ctx.close()


def test_dist_operator_with_iterative_shuffling(pg):
    A = _shuffled_band(5000, 8)
    op = sd.DistSpgemm(A, None, impl="drop", n_sections=64, seed=5, device_index=0, shuffling="iterative",
                       shuffle_iterations=10)
    op.gemm()
    M = sd.merge_unshuffle([op.result()], op.partitions, op.A_shuffle, op.B_shuffle, A.rows, A.cols)
    E = orc.spgemm(to_o(A), to_o(A))
    orc.assert_same(to_o(M), E, rtol=0.0)
    ers, ecs, _ = orc.iterative_shuffle(E, 10)
    assert np.array_equal(op.A_shuffle, ers) and np.array_equal(op.B_shuffle, ecs)
    op.close()


def test_relabel_columns_matches_numpy():
    """Columns renamed by a permutation and rows re-sorted on the device = the same on the host."""
    import torch
    ctx = capi.Context(0)
    A = hostcsr.rmat(11, 6, seed=9)
    d = ctx.from_host(A.rows, A.cols, A.row_ptr, A.col_idx, A.val)
    perm = capi.shuffle_random(A.cols, 12)                       # position -> original column
    new_id = np.empty(A.cols, dtype=np.int64)
    new_id[perm] = np.arange(A.cols)                             # original column -> position
    t = torch.from_numpy(new_id.astype(np.int32)).cuda()
    r = ctx.relabel_columns(d, t.data_ptr(), A.cols)
    rp, ci, v = r.to_host()
    want = sd.relabel_cols(A, new_id)
    assert np.array_equal(rp, want.row_ptr) and np.array_equal(ci, want.col_idx) and np.array_equal(v, want.val)
    E = hostcsr.HostCsr.from_coo(5, 7, [], [], [])               # no entries at all
    de = ctx.from_host(E.rows, E.cols, E.row_ptr, E.col_idx, E.val)
    t7 = torch.arange(7, dtype=torch.int32).cuda()
    re_ = ctx.relabel_columns(de, t7.data_ptr(), 7)
    assert re_.info() == (5, 7, 0)
    for h in (r, d, re_, de):
        h.free()
    ctx.close()
