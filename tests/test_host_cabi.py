"""CPU tests of the C-ABI library: it loads, exports every declared symbol, and its host-only
entry points (partitions, shuffle) agree with the oracle's restatement.  No GPU compute here."""
import os
import re

import numpy as np
import pytest

from oracle import oracle as orc
from spgemm_amd import capi, hostcsr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    lib = capi.load()
    header = open(os.path.join(ROOT, "include", "spgemm_amd.h")).read()
    declared = set(re.findall(r"\b(spgemm_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations found"
    for name in declared:
        assert hasattr(lib, name), f"{name} is declared in include/spgemm_amd.h but not exported"
    assert declared == set(capi.EXPORTED_SYMBOLS)
    assert lib.spgemm_version() >= 100


def test_no_cpu_fallback_without_device():
    if capi.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(capi.SpgemmError):
        capi.Context(0)


@pytest.mark.parametrize("h,w,P", [(165, 165, 2), (165, 165, 4), (10, 7, 3), (5, 5, 8), (1508065, 1508065, 8)])
def test_partition_naive_matches_oracle(h, w, P):
    assert capi.partition_naive(h, w, P) == orc.partition_naive(h, w, P)


def test_partition_balanced_matches_oracle():
    rng = np.random.default_rng(1)
    for P in (2, 3, 8):
        w = rng.integers(0, 50, size=400)
        got = capi.partition_balanced(w, 400, P)
        assert got == orc.partition_balanced(w, 400, P)
        # contiguous cover of all rows
        assert got[0][0] == 0 and got[-1][1] == 400
        assert all(got[i][1] == got[i + 1][0] for i in range(P - 1))
    # one dominant row: the cut lands right after it crosses the share
    w = np.array([1, 1, 100, 1, 1, 1])
    assert capi.partition_balanced(w, 6, 2) == orc.partition_balanced(w, 6, 2)


def test_shuffle_is_a_seeded_permutation():
    a = capi.shuffle_random(1000, 42)
    b = capi.shuffle_random(1000, 42)
    c = capi.shuffle_random(1000, 43)
    assert np.array_equal(a, b) and not np.array_equal(a, c)
    assert np.array_equal(np.sort(a), np.arange(1000))
    assert capi.shuffle_random(0, 1).size == 0


def test_bitmap_word_packing_roundtrip():
    rng = np.random.default_rng(2)
    for n in (4, 64, 65, 512):
        bits = rng.random(n) < 0.3
        assert np.array_equal(capi.bitmap_bits(capi.bitmap_words(bits), n), bits)


def test_generators_are_deterministic_and_well_formed():
    A = hostcsr.erdos_renyi(500, 16, seed=42)
    B = hostcsr.erdos_renyi(500, 16, seed=42)
    assert np.array_equal(A.col_idx, B.col_idx) and np.array_equal(A.val, B.val)
    assert A.nnz <= 500 * 16 and A.val.min() >= 0.5
    S = hostcsr.shell_like(n_nodes=400, grid_w=20)
    assert S.rows == 2000 and S.nnz <= 35 * S.rows
    lens = np.diff(S.row_ptr)
    assert lens.max() == 35 and lens.min() >= 15
    # symmetric pattern
    o = orc.OCsr(S.rows, S.cols, S.row_ptr, S.col_idx, S.val)
    t = o.transpose().canonical()
    assert np.array_equal(t.row_ptr, o.row_ptr) and np.array_equal(t.col_idx, o.canonical().col_idx)
    R = hostcsr.rmat(8, 4, seed=1)
    assert R.rows == 256 and R.nnz <= 4 * 256
    for m in (A, S, R):  # rows sorted, no duplicates
        rid = m.row_ids()
        key = rid * m.cols + m.col_idx
        assert np.all(np.diff(key) > 0)


def test_mtx_roundtrip(tmp_path, golden_dir):
    A = hostcsr.read_mtx(f"{golden_dir}/ref_first/A.mtx")
    p = tmp_path / "a.mtx"
    hostcsr.write_mtx(str(p), A)
    B = hostcsr.read_mtx(str(p))
    s = A.sorted_rows()
    assert np.array_equal(B.row_ptr, s.row_ptr) and np.array_equal(B.col_idx, s.col_idx)
    assert np.array_equal(B.val, s.val)  # 17 significant digits round-trip exactly


def test_row_products_and_locality_rule():
    """Host helpers of the distributed operator: products per row, and the rule that decides where
    a column shuffle is applied."""
    from oracle import oracle as orc
    from spgemm_amd import dist as sd
    A = hostcsr.rmat(9, 6, seed=4)
    O = orc.OCsr(A.rows, A.cols, A.row_ptr, A.col_idx, A.val)
    _, st = orc.spgemm(O, O, with_stats=True)
    w = sd.row_products(A, A)
    assert int(w.sum()) == st["nprod"] and len(w) == A.rows
    band = hostcsr.banded_random(200_000, 4, 50, seed=1)
    assert sd.product_rows_are_narrow(band, band)                    # a band keeps its columns together
    er = hostcsr.erdos_renyi(200_000, 4, seed=1)
    assert not sd.product_rows_are_narrow(er, er)                    # random columns: nothing to lose
    empty = hostcsr.HostCsr.from_coo(5, 5, [], [], [])
    assert not sd.product_rows_are_narrow(empty, empty)
