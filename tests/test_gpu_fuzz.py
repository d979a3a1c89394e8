"""Randomised GPU parity sweep: small matrices of varied structure, multiplied under randomly chosen
row-binning options so that every kernel family (window, row groups, LDS hash, workgroup hash,
big-row windows and slices, unsorted-B fallback) sees them; each result is compared bit for bit
with the CPU oracle.  Seeds are fixed: a failure names its case."""
import os

import numpy as np
import pytest

from oracle import oracle as orc
from spgemm_amd import capi, hostcsr

pytestmark = pytest.mark.gpu

DEFAULTS = {"window_max_span": 65536, "window_ratio": 128, "hash_max_nprod": 8192, "big_win_words": 32768,
            "slice_blocks": 0, "hash_num_slice": 1024, "row_groups": 1, "flat_walk": 1, "big_rows": 1,
            "g_thresh2": 24, "lane_order_adds": 1, "slice_queues": 1, "slice_packed": 0, "short_packed": 2, "grp_compact": 2, "short_key32": 1, "analyze_lanes": 0, "walk_tail": 2, "grp_runs": 1, "big_sym_bounds": 1, "sym_rpw": 8, "grp_rpw": 8, "slice_sampled": 1, "rpw_fill": 16,
            "slice_sorted": 3, "big_sym_sorted": 1, "slice_run_min": 64, "short_rpw": 16, "short_num_rpw": 8,
            "poison_output": 1}  # (always: the second multiply of a case gets the first one's arrays back from the pool)


def to_o(h):
    return orc.OCsr(h.rows, h.cols, h.row_ptr, h.col_idx, h.val)


def random_matrix(rng, rows, cols):
    kind = rng.integers(0, 5)
    if kind == 0:      # uniform random
        nnz = int(rng.integers(1, 12) * rows)
        r, c = rng.integers(0, rows, nnz), rng.integers(0, cols, nnz)
    elif kind == 1:    # skewed: a few hub rows and hub columns
        nnz = int(rng.integers(2, 10) * rows)
        r = (rows * rng.random(nnz) ** 3).astype(np.int64)
        c = (cols * rng.random(nnz) ** 3).astype(np.int64)
    elif kind == 2:    # banded
        nnz = int(rng.integers(2, 9) * rows)
        r = rng.integers(0, rows, nnz)
        c = np.clip((r * cols) // rows + rng.integers(-20, 21, nnz), 0, cols - 1)
    elif kind == 3:    # blocks of identical rows (row groups), some long rows
        d = int(rng.integers(2, 7))
        base = rng.integers(0, cols, (max(rows // d, 1), int(rng.integers(2, 30))))
        r = np.repeat(np.arange(base.shape[0] * d), base.shape[1]) % rows
        c = np.repeat(base, d, axis=0).ravel()
    else:              # one dense-ish row and column on top of a sparse background
        nnz = int(3 * rows)
        r, c = rng.integers(0, rows, nnz), rng.integers(0, cols, nnz)
        hub = int(rng.integers(0, rows))
        r = np.concatenate([r, np.full(cols // 2, hub)])
        c = np.concatenate([c, rng.choice(cols, cols // 2, replace=False)])
    v = rng.integers(-3, 4, len(r)).astype(np.float64)  # small integers: exact cancellations do happen
    v[v == 0] = 0.5
    M = hostcsr.HostCsr.from_coo(rows, cols, r, c, v, sum_duplicates=True).sorted_rows()
    keep = M.val != 0.0  # duplicates that summed to 0.0 do not stay as explicit zeros
    return hostcsr.HostCsr.from_coo(rows, cols, M.row_ids()[keep], M.col_idx[keep].astype(np.int64), M.val[keep]).sorted_rows()


def make_case(seed):
    """(A, B, engine options) of fuzz case `seed`."""
    rng = np.random.default_rng(1000 + seed)
    m, k, n = (int(x) for x in rng.integers(40, 2500, 3))
    if seed % 3 == 0:
        m = k = n = int(rng.integers(300, 3000))
    A, B = random_matrix(rng, m, k), random_matrix(rng, k, n)
    # an entry whose sum of duplicates is 0.0 must not reach the operands as an explicit zero
    for M in (A, B):
        assert np.all(M.val != 0.0)
    opts = dict(DEFAULTS)
    opts["window_max_span"] = int(rng.choice([0, 512, 65536]))
    opts["hash_max_nprod"] = int(rng.choice([0, 64, 512, 4096]))
    opts["big_win_words"] = int(rng.choice([1024, 32768]))
    opts["slice_blocks"] = int(rng.choice([0, 1, 7]))
    opts["hash_num_slice"] = int(rng.choice([64, 1024, 4096]))
    opts["row_groups"] = int(rng.integers(0, 2))
    opts["flat_walk"] = int(rng.integers(0, 2))
    opts["big_rows"] = int(rng.integers(0, 4) != 0)
    opts["g_thresh2"] = int(rng.choice([24, 100000]))
    opts["lane_order_adds"] = int(rng.integers(0, 3) != 0)  # 0: claim rounds instead of one ordered LDS add
    opts["slice_queues"] = int(rng.choice([1, 3, 8]))
    opts["slice_packed"] = int(rng.integers(0, 2))
    opts["short_packed"] = int(rng.integers(0, 3))
    opts["grp_compact"] = int(rng.integers(0, 3))
    opts["short_key32"] = int(rng.integers(0, 2))
    opts["analyze_lanes"] = int(rng.choice([0, 8, 16, 32]))
    opts["walk_tail"] = int(rng.integers(0, 3))
    opts["grp_runs"] = int(rng.integers(0, 4) != 0)
    opts["big_sym_bounds"] = int(rng.integers(0, 3) != 0)
    opts["sym_rpw"] = int(rng.choice([1, 3, 8]))
    opts["grp_rpw"] = int(rng.choice([1, 2, 8]))
    opts["slice_sampled"] = int(rng.integers(0, 3) != 0)
    opts["rpw_fill"] = int(rng.choice([0, 0, 16]))  # 0: several rows / groups per wavefront on these small inputs too
    opts["slice_sorted"] = int(rng.integers(0, 4))   # row order, by first column, the dearest first, by magnitude then column
    opts["big_sym_sorted"] = int(rng.integers(0, 2))
    opts["slice_run_min"] = int(rng.choice([4, 64]))
    opts["short_rpw"] = int(rng.choice([2, 16]))
    opts["short_num_rpw"] = int(rng.choice([1, 3, 8]))
    unsorted_b = seed % 5 == 4
    if unsorted_b:  # rows of B in reverse column order: the general (unsorted) path
        rp = B.row_ptr.astype(np.int64)
        order = np.concatenate([np.arange(rp[i + 1] - 1, rp[i] - 1, -1) for i in range(B.rows)]) if B.nnz else np.zeros(0, np.int64)
        B = hostcsr.HostCsr(B.rows, B.cols, B.row_ptr, B.col_idx[order], B.val[order])
    return A, B, opts


# SPGEMM_FUZZ_CASES: more seeds for a one-off sweep (the suite runs the first 96)
N_CASES = int(os.environ.get("SPGEMM_FUZZ_CASES", "96"))


@pytest.mark.parametrize("seed", range(N_CASES))
def test_random_structure_and_options(seed):
    A, B, opts = make_case(seed)
    ctx = capi.Context(0)
    try:
        for name, value in opts.items():
            ctx.set_option(name, value)
        dA = ctx.from_host(A.rows, A.cols, A.row_ptr, A.col_idx, A.val)
        dB = ctx.from_host(B.rows, B.cols, B.row_ptr, B.col_idx, B.val)
        for it in range(2):  # the second multiply reuses the context's scratch
            dC = ctx.multiply(dA, dB)
            rp, ci, v = dC.to_host()
            dC.free()
            E = orc.spgemm(to_o(A), to_o(B))
            try:
                orc.assert_same(orc.OCsr(A.rows, B.cols, rp, ci, v), E, rtol=0.0)
            except AssertionError as err:  # say where: which multiply, how many entries, which rows, unwritten (poisoned) ones
                detail = f"multiply {it}: nnz {len(ci)} (oracle {E.nnz})"
                if len(ci) == E.nnz and np.array_equal(rp, E.row_ptr):
                    Ec = E.canonical()
                    bad = np.flatnonzero((v != Ec.val) | (ci != Ec.col_idx))
                    rows_of = np.searchsorted(rp, bad, side="right") - 1
                    detail += (f"; {len(bad)} entries differ in rows {sorted(set(rows_of.tolist()))[:8]} (lengths "
                               f"{[int(rp[r + 1] - rp[r]) for r in sorted(set(rows_of.tolist()))[:8]]}, A rows "
                               f"{[int(A.row_ptr[r + 1] - A.row_ptr[r]) for r in sorted(set(rows_of.tolist()))[:8]]}), at "
                               f"{[int(b - rp[r]) for b, r in zip(bad[:8], rows_of[:8])]}; NaN values {int(np.isnan(v).sum())}, "
                               f"poisoned columns {int((ci == 0xFFFFFFFF).sum())}; got {v[bad[:4]].tolist()} want {Ec.val[bad[:4]].tolist()}")
                raise AssertionError(f"{err} [{detail}]") from err
        dA.free()
        dB.free()
    finally:
        ctx.close()  # (frees whatever a failed case left alive on the context)
