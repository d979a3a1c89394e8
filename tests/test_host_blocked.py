"""Host mirror of the reference's BlockedCSRMatrix (include/matrix_impl.hpp:388-538): section cut,
filter(bitmap), the receiver's view over the filtered bytes and row() - built with plain g++ (no
GPU) and checked against the oracle's restatement of the blob sizes and against the bytes the
reference binary sent (drop_at_once_parallel, P = 2, N_SECTIONS = 4 on `first`)."""
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle as orc
from spgemm_amd import hostcsr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "spgemm_amd", "host")


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("blocked") / "blocked_check")
    subprocess.check_call(["g++", "-std=c++20", "-O1", "-Wall", "-o", out, os.path.join(HOST, "blocked_check.cpp")])
    return out


def run(checker, path, n_sections, c0, c1, bits):
    res = subprocess.run([checker, path, str(n_sections), str(c0), str(c1), "".join("1" if b else "0" for b in bits)],
                         capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    return tuple(int(x) for x in res.stdout.split())


def test_blocked_filter_sizes_match_oracle_and_reference_bytes(checker, golden_dir):
    path = f"{golden_dir}/ref_first/A.mtx"
    A = hostcsr.read_mtx(path)
    Ao = orc.OCsr(A.rows, A.cols, A.row_ptr, A.col_idx, A.val)
    N = 4
    sb = orc.section_bounds(A.rows, N)
    for P in (2, 4):
        parts = orc.partition_naive(A.rows, A.cols, P)
        bitmaps = [orc.section_bitmap(Ao.select_rows(np.arange(p[0], p[1])), N) for p in parts]
        sent = [0] * P
        for r, p in enumerate(parts):
            blk = Ao.select_cols(np.arange(p[2], p[3]))
            for d in range(P):
                whole, filtered, checked = run(checker, path, N, p[2], p[3], bitmaps[d])
                assert whole == orc.blocked_fields_size(N) + int(orc.blocked_section_sizes(blk, N).sum())
                assert filtered == orc.filter_size(blk, N, bitmaps[d])
                rows_kept = sum(int(sb[s + 1] - sb[s]) for s in range(N) if bitmaps[d][s])
                assert checked == rows_kept
                if d != r:
                    sent[r] += filtered
        # bytes the reference binary's ranks sent in one gemm (`bytes` rows of its measurement files)
        want = {2: [10520, 448], 4: [14696, 8360, 3872, 1344]}[P]
        assert sent == want, (P, sent)


def test_blocked_rejects_more_sections_than_rows(checker, golden_dir):
    res = subprocess.run([checker, f"{golden_dir}/ref_test/A.mtx", "4", "0", "3", "1111"], capture_output=True, text=True)
    assert res.returncode == 1 and "fewer rows than sections" in res.stderr
