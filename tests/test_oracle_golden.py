"""CPU tests: the oracle against the reference's own fixtures and recorded observations.

Fixtures under tests/golden/ref_test and ref_first are data files copied from the reference
(matrices/test, matrices/first).  The numbers in APPENDIX were recorded from the reference binary
during the survey of the reference (its run directories: partitions.csv, C_<rank>.mtx headers, the
`bytes` rows of measurements_<rank>.csv, the "Bitmap drop percentage" lines) for matrices/first
with N_SECTIONS=4.
"""
import numpy as np
import pytest

from oracle import oracle as orc


def test_fixture_test_a_times_b(golden_dir):
    A = orc.read_mtx(f"{golden_dir}/ref_test/A.mtx")
    B = orc.read_mtx(f"{golden_dir}/ref_test/B.mtx")
    E = orc.read_mtx(f"{golden_dir}/ref_test/C_expected.mtx")
    orc.assert_same(orc.spgemm(A, B), E, rtol=0.0)
    # the five expected entries, spelled out (matrices/test/C_expected.mtx)
    want = {(0, 1): 12.0, (0, 3): 16.0, (1, 0): 58.0, (1, 2): 78.0, (1, 3): 72.0}
    C = orc.spgemm(A, B).canonical()
    rows = np.repeat(np.arange(C.rows), np.diff(C.row_ptr).astype(int))
    got = {(int(r), int(c)): float(v) for r, c, v in zip(rows, C.col_idx, C.val)}
    assert got == want


def test_fixture_first_a_times_a_bitwise(golden_dir):
    A = orc.read_mtx(f"{golden_dir}/ref_first/A.mtx")
    E = orc.read_mtx(f"{golden_dir}/ref_first/C_expected.mtx")
    C, st = orc.spgemm(A, A, with_stats=True)
    assert C.nnz == 3067 and st["nprod"] == 5860
    orc.assert_same(C, E, rtol=0.0)  # 17-digit file values reproduced bit for bit


def test_baseline_inner_product_equals_gustavson(golden_dir):
    A = orc.read_mtx(f"{golden_dir}/ref_first/A.mtx").canonical()
    E = orc.read_mtx(f"{golden_dir}/ref_first/C_expected.mtx")
    orc.assert_same(orc.baseline_inner(A, A.transpose()), E, rtol=0.0)


APPENDIX = {
    2: dict(parts=[(0, 83, 0, 83), (83, 165, 83, 165)], nnz=[782, 2285], drop=[0.5, 0.0],
            baseline=[10048, 4264], outer=[20752, 9184], at_once=[10520, 448]),
    4: dict(parts=[(0, 42, 0, 42), (42, 83, 42, 83), (83, 124, 83, 124), (124, 165, 124, 165)],
            nnz=[182, 600, 759, 1526], drop=[0.75, 0.5, 0.25, 0.0],
            baseline=[17496, 12720, 7584, 5280], outer=[25280, 18944, 12096, 9024],
            at_once=[14696, 8360, 3872, 1344]),
}


@pytest.mark.parametrize("P", [2, 4])
def test_distributed_flow_matches_recorded_reference_runs(golden_dir, P):
    A = orc.read_mtx(f"{golden_dir}/ref_first/A.mtx")
    E = orc.read_mtx(f"{golden_dir}/ref_first/C_expected.mtx")
    want = APPENDIX[P]
    parts = orc.partition_naive(165, 165, P)
    assert parts == want["parts"]
    ident = np.arange(165)
    res = orc.distributed_reference(A, A, P, parts, ident, ident, n_sections=4)
    assert [r.C.nnz for r in res] == want["nnz"]
    # the reference prints the drop percentages unordered; compare as multisets
    assert sorted(1.0 - r.bitmap.mean() for r in res) == sorted(want["drop"])
    # bytes sent per gemm: Baseline sends its transposed block P-1 times, Outer P times,
    # DropAtOnceParallel the filtered blocked CSR once to every other rank
    assert [(P - 1) * orc.csr_blob_size(r.block_B.cols, r.block_B.nnz) for r in res] == want["baseline"]
    assert [P * orc.csr_blob_size(r.block_B.rows, r.block_B.nnz) for r in res] == want["outer"]
    assert [sum(orc.filter_size(res[s].block_B, 4, res[d].bitmap) for d in range(P) if d != s)
            for s in range(P)] == want["at_once"]
    merged = orc.merge_unshuffle([r.C for r in res], parts, ident, ident, 165, 165)
    orc.assert_same(merged, E, rtol=0.0)


def test_shuffled_flow_unshuffles_to_expected(golden_dir):
    A = orc.read_mtx(f"{golden_dir}/ref_first/A.mtx")
    E = orc.read_mtx(f"{golden_dir}/ref_first/C_expected.mtx")
    a_sh = np.loadtxt(f"{golden_dir}/ref_first/A_shuffle", dtype=np.int64)
    b_sh = np.loadtxt(f"{golden_dir}/ref_first/B_shuffle", dtype=np.int64)
    assert sorted(a_sh) == list(range(165)) and sorted(b_sh) == list(range(165))
    parts = orc.partition_naive(165, 165, 4)
    res = orc.distributed_reference(A, A, 4, parts, a_sh, b_sh, n_sections=4)
    merged = orc.merge_unshuffle([r.C for r in res], parts, a_sh, b_sh, 165, 165)
    orc.assert_same(merged, E, rtol=0.0)


def test_cells_add_erases_exact_zero():
    # 1*1 + 1*(-1) == 0 exactly: the entry is erased; re-adding later re-creates it
    A = orc.OCsr.from_coo(1, 3, [0, 0, 0], [0, 1, 2], [1.0, 1.0, 1.0])
    B = orc.OCsr.from_coo(3, 2, [0, 1, 2, 0], [0, 0, 0, 1], [1.0, -1.0, 4.0, 2.0])
    C = orc.spgemm(A, B)
    assert C.col_idx.tolist() == [0, 1] and C.val.tolist() == [4.0, 2.0]
    B2 = orc.OCsr.from_coo(3, 2, [0, 1, 0], [0, 0, 1], [1.0, -1.0, 2.0])
    C2 = orc.spgemm(A, B2)
    assert C2.col_idx.tolist() == [1] and C2.val.tolist() == [2.0]


def test_insertion_order_output():
    A = orc.OCsr.from_coo(1, 2, [0, 0], [1, 0], [1.0, 1.0])
    B = orc.OCsr.from_coo(2, 4, [0, 0, 1, 1], [0, 3, 2, 3], [1.0, 1.0, 1.0, 1.0])
    C = orc.spgemm(A, B, sort_cols=False)  # A row visits k=1 first: columns 2, 3, then 0
    assert C.col_idx.tolist() == [2, 3, 0] and C.val.tolist() == [1.0, 2.0, 1.0]


def test_section_bitmap_reference_rule():
    # width 10, 4 sections -> section width 2; columns 8 and 9 clamp into section 3
    A = orc.OCsr.from_coo(2, 10, [0, 0, 1], [0, 9, 5], [1.0, 1.0, 1.0])
    assert orc.section_bitmap(A, 4).tolist() == [True, False, True, True]
    with pytest.raises(ValueError):
        orc.section_bitmap(A, 512)  # width / n_sections == 0: the reference asserts


def test_against_scipy_on_random_inputs():
    import scipy.sparse as sp
    rng = np.random.default_rng(0)
    for n, d in ((50, 0.1), (300, 0.02), (1000, 0.004)):
        M = sp.random(n, n, density=d, random_state=rng, format="csr", dtype=np.float64)
        M.sort_indices()
        E = (M @ M).tocsr()
        E.sort_indices()
        C = orc.spgemm(orc.OCsr.from_scipy(M), orc.OCsr.from_scipy(M))
        orc.assert_same(C, orc.OCsr.from_scipy(E), rtol=0.0)


def test_iterative_shuffle_restated():
    """The oracle's restatement of partition::iterative_shuffle (reference src/partition.cpp:123-202):
    valid permutations, identity for zero iterations, a first row iteration that sorts the rows by
    the mean of their column indices, and a shuffled band that is pulled back together."""
    from spgemm_amd import hostcsr
    from spgemm_amd import dist as sd
    n = 1200
    A = hostcsr.banded_random(n, 5, 20, seed=2)
    perm = np.random.default_rng(0).permutation(n)
    inv = np.empty(n, dtype=np.int64)
    inv[perm] = np.arange(n)
    S = sd.relabel_cols(sd.select_rows(A, perm), inv)
    O = orc.OCsr(S.rows, S.cols, S.row_ptr, S.col_idx, S.val)
    rs, cs, var = orc.iterative_shuffle(O, 0)
    assert np.array_equal(rs, np.arange(n)) and np.array_equal(cs, np.arange(n)) and len(var) == 0
    rs, cs, var = orc.iterative_shuffle(O, 1)
    rp = O.row_ptr.astype(np.int64)
    means = np.array([O.col_idx[rp[r]:rp[r + 1]].astype(np.float64).mean() if rp[r + 1] > rp[r] else n for r in range(n)])
    assert np.array_equal(rs, np.argsort(means, kind="stable")) and np.array_equal(cs, np.arange(n))
    rs, cs, var = orc.iterative_shuffle(O, 30)
    assert np.array_equal(np.sort(rs), np.arange(n)) and np.array_equal(np.sort(cs), np.arange(n))
    pos_r = np.empty(n, dtype=np.int64); pos_r[rs] = np.arange(n)
    pos_c = np.empty(n, dtype=np.int64); pos_c[cs] = np.arange(n)
    rid = np.repeat(np.arange(n), np.diff(rp))
    after = np.abs(pos_r[rid] - pos_c[O.col_idx.astype(np.int64)]).mean()
    before = np.abs(rid - O.col_idx.astype(np.int64)).mean()
    assert after < 0.2 * before and var[-1] < var[0]
