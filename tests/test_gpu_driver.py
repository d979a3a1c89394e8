"""GPU tests of the C++ host driver (spgemm_amd/dphpc_amd): the reference's command line, run
directory layout and per-rank outputs, checked against the reference's fixtures."""
import csv
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle as orc
from spgemm_amd import hostcsr
from spgemm_amd import dist as sd

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "spgemm_amd", "dphpc_amd")


def driver_cmd(tmp_path, algo, matrix, shuffle="none", part="naive", extra=()):
    run = tmp_path / f"run_{algo}_{matrix}_{shuffle}_{'_'.join(extra)}"
    return run, [DRIVER, algo, matrix, str(run), "2", "1", shuffle, part, "false", "true",
                 os.path.join(ROOT, "tests", "golden"), *extra]


def run_driver(tmp_path, algo, matrix, shuffle="none", part="naive", extra=()):
    run, cmd = driver_cmd(tmp_path, algo, matrix, shuffle, part, extra)
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    return run, p.stdout


def merged(run, n_ranks, n_rows, n_cols):
    parts = []
    with open(run / "partitions.csv") as f:
        rows = list(csv.DictReader(f))
    parts = [(int(r["start_row"]), int(r["end_row"]), int(r["start_col"]), int(r["end_col"])) for r in rows]
    a_sh = np.loadtxt(run / "A_shuffle", dtype=np.int64, ndmin=1)
    b_sh = np.loadtxt(run / "B_shuffle", dtype=np.int64, ndmin=1)
    Cs = [hostcsr.read_mtx(str(run / f"C_{r}.mtx")) for r in range(n_ranks)]
    M = sd.merge_unshuffle(Cs, parts, a_sh, b_sh, n_rows, n_cols)
    return M, parts, Cs


def to_o(h):
    return orc.OCsr(h.rows, h.cols, h.row_ptr, h.col_idx, h.val)


def test_driver_exists():
    assert os.path.exists(DRIVER), "build it with make -C spgemm_amd/host"


def test_config1_test_fixture_two_ranks_baseline(tmp_path, golden_dir):
    # matrices/test: A (2x3) * B (3x4), --impl baseline, 2 ranks, against C_expected.mtx
    run, _ = run_driver(tmp_path, "baseline", "ref_test", extra=("--ranks", "2", "--impl", "baseline", "--sections", "3"))
    M, parts, _ = merged(run, 2, 2, 4)
    assert parts == [(0, 1, 0, 2), (1, 2, 2, 4)]
    orc.assert_same(to_o(M), orc.read_mtx(f"{golden_dir}/ref_test/C_expected.mtx"), rtol=0.0)
    assert (run / "algo").read_text().strip() == "baseline"
    assert (run / "matrix").read_text().strip() == "ref_test"


@pytest.mark.parametrize("ranks,algo,shuffle", [(1, "drop_parallel", "none"), (2, "drop_parallel", "none"),
                                                (4, "drop_at_once_parallel", "none"), (4, "baseline", "none"),
                                                (4, "drop", "random"), (3, "outer", "random"),
                                                (3, "drop", "iterative")])
def test_first_fixture_all_variants(tmp_path, golden_dir, ranks, algo, shuffle):
    run, out = run_driver(tmp_path, algo, "ref_first", shuffle=shuffle, extra=("--ranks", str(ranks), "--sections", "4"))
    M, parts, Cs = merged(run, ranks, 165, 165)
    E = orc.read_mtx(f"{golden_dir}/ref_first/C_expected.mtx")
    orc.assert_same(to_o(M), E, rtol=0.0)
    assert parts == orc.partition_naive(165, 165, ranks)
    if shuffle == "iterative":  # the permutations the driver saved are the oracle's (20 iterations by default)
        rs, cs, _ = orc.iterative_shuffle(E, 20)
        assert np.array_equal(np.loadtxt(run / "A_shuffle", dtype=np.int64), rs)
        assert np.array_equal(np.loadtxt(run / "B_shuffle", dtype=np.int64), cs)
    if shuffle == "none" and ranks in (2, 4):
        want_nnz = {2: [782, 2285], 4: [182, 600, 759, 1526]}[ranks]      # recorded reference runs
        assert [c.nnz for c in Cs] == want_nnz
        if algo.startswith("drop"):
            want_drop = {2: [0.5, 0.0], 4: [0.75, 0.5, 0.25, 0.0]}[ranks]
            got = sorted(float(l.split(":")[1]) for l in out.splitlines() if l.startswith("Bitmap drop percentage"))
            assert got == sorted(want_drop)
    for r in range(ranks):
        with open(run / f"measurements_{r}.csv") as f:
            rows = list(csv.reader(f))
        assert rows[0] == ["func", "duration", "bytes"]
        assert [x[0] for x in rows[1:3]] == ["bytes", "bytes"]            # two timed runs
        funcs = {x[0] for x in rows[3:]}
        assert {"gemm", "mult", "wait_all"} <= funcs


def test_balanced_partition_and_shuffle_impl(tmp_path, golden_dir):
    run, _ = run_driver(tmp_path, "drop_parallel", "ref_first", part="balanced",
                        extra=("--ranks", "4", "--sections", "4", "--impl", "shuffle", "--seed", "9"))
    M, parts, _ = merged(run, 4, 165, 165)
    orc.assert_same(to_o(M), orc.read_mtx(f"{golden_dir}/ref_first/C_expected.mtx"), rtol=0.0)
    assert parts[0][0] == 0 and parts[-1][1] == 165
    assert sorted(np.loadtxt(run / "A_shuffle", dtype=np.int64)) == list(range(165))


def test_driver_rejects_bad_arguments(tmp_path):
    p = subprocess.run([DRIVER, "nope"], capture_output=True, text=True)
    assert p.returncode == 1 and "Did not get enough arguments" in p.stderr
    p = subprocess.run([DRIVER, "comb", "ref_first", str(tmp_path / "x"), "1", "0", "none", "naive", "false", "true",
                        os.path.join(ROOT, "tests", "golden")], capture_output=True, text=True)
    assert p.returncode == 1 and "Unknown algorithm type" in p.stderr


@pytest.mark.parametrize("where", ["result", "b"])
def test_column_shuffle_applied_to_result_or_to_b(tmp_path, golden_dir, where):
    """Both places to apply the column shuffle give the reference's C, entry for entry."""
    run, _ = run_driver(tmp_path, "drop_parallel", "ref_first", shuffle="random",
                        extra=("--ranks", "3", "--sections", "4", "--relabel", where))
    M, _, _ = merged(run, 3, 165, 165)
    orc.assert_same(to_o(M), orc.read_mtx(f"{golden_dir}/ref_first/C_expected.mtx"), rtol=0.0)


@pytest.mark.parametrize("ranks,stages,algo,shuffle", [(4, 3, "drop_parallel", "none"), (2, 5, "baseline", "none"),
                                                       (3, 2, "drop", "random")])
def test_staged_schedule_gives_the_same_result(tmp_path, golden_dir, ranks, stages, algo, shuffle):
    """--stages S: every rank cuts its rows into S blocks and exchanges the sections of block s + 1
    while block s is multiplied; the merged result is the reference's C, entry for entry."""
    run, out = run_driver(tmp_path, algo, "ref_first", shuffle=shuffle,
                          extra=("--ranks", str(ranks), "--sections", "4", "--stages", str(stages)))
    M, parts, Cs = merged(run, ranks, 165, 165)
    orc.assert_same(to_o(M), orc.read_mtx(f"{golden_dir}/ref_first/C_expected.mtx"), rtol=0.0)
    if shuffle == "none" and ranks == 4:
        assert [c.nnz for c in Cs] == [182, 600, 759, 1526]               # as the unstaged run
        # a rank's bitmap is the union of its blocks': the same sections as the unstaged run drops
        got = sorted(float(l.split(":")[1]) for l in out.splitlines() if l.startswith("Bitmap drop percentage"))
        assert got == sorted([0.75, 0.5, 0.25, 0.0])
    for r in range(ranks):
        with open(run / f"measurements_{r}.csv") as f:
            funcs = {x[0] for x in csv.reader(f)}
        assert {"gemm", "mult", "wait_all", "filter", "deserialize"} <= funcs


@pytest.mark.parametrize("stages", [1, 2])
def test_a_failing_rank_ends_the_run_with_an_error(tmp_path, stages):
    """A rank that gives up inside a run does not leave the others waiting: the process ends with
    exit status 1 and the rank's message, and writes no result."""
    run, cmd = driver_cmd(tmp_path, "drop_parallel", "ref_first", extra=("--ranks", "3", "--sections", "4", "--stages", str(stages)))
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=120, env=dict(os.environ, DPHPC_AMD_FAIL_RANK="1"))
    assert p.returncode == 1, p.stdout[-2000:] + p.stderr[-2000:]
    assert "rank 1: failure requested through DPHPC_AMD_FAIL_RANK" in p.stderr
    assert "another rank failed" in p.stderr            # the two others were told
    assert not (run / "C_0.mtx").exists()
