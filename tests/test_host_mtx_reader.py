"""The host layer's Matrix Market reader (spgemm_amd/host/matrix.hpp: mapped file, one piece per
thread, std::from_chars) against a plain Python parse of the same files: entries, their ORDER (rows
keep the order in which the file lists their entries, as in the reference's loader) and the bits of
the values; symmetric and pattern files, comments and blank lines between entries, CRLF, a leading
plus sign, and the two ways a file can disagree with its size line."""
import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "spgemm_amd", "host")
MASK = (1 << 64) - 1


@pytest.fixture(scope="module")
def reader(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("mtx") / "mtx_check")
    subprocess.check_call(["g++", "-std=c++20", "-O2", "-Wall", "-Wextra", "-pthread", "-o", out, os.path.join(HOST, "mtx_check.cpp")])
    return out


def checksum(rows, cols, vals):
    h = 0
    for i, (r, c, v) in enumerate(zip(rows, cols, vals)):
        bits = struct.unpack("<Q", struct.pack("<d", v))[0]
        s = i % 63
        h = (h + (i + 1) * (r * 1000003 + c)) & MASK
        h ^= ((bits << s) | (bits >> ((64 - s) % 64))) & MASK
    return h


def python_parse(text):
    lines = text.replace("\r\n", "\n").split("\n")
    header = lines[0].lower()
    symmetric, pattern = "symmetric" in header, "pattern" in header
    body = [l for l in lines[1:] if l.strip() and not l.lstrip().startswith("%")]
    nr, nc, nnz = (int(x) for x in body[0].split())
    rows, cols, vals = [], [], []
    for l in body[1:]:
        f = l.split()
        r, c = int(f[0]), int(f[1])
        v = 1.0 if pattern else float(f[2])
        rows.append(r - 1); cols.append(c - 1); vals.append(v)
        if symmetric and r != c:
            rows.append(c - 1); cols.append(r - 1); vals.append(v)
    return nr, nc, nnz, rows, cols, vals


def run(reader, path, threads):
    return subprocess.run([reader, path], capture_output=True, text=True, env=dict(os.environ, SPGEMM_IO_THREADS=str(threads)))


@pytest.mark.parametrize("kind", ["general", "symmetric", "pattern"])
def test_reader_matches_python_parse_in_order(reader, tmp_path, kind):
    rng = np.random.default_rng({"general": 1, "symmetric": 2, "pattern": 3}[kind])
    n = 120_000                                   # about 3 MB of entry lines: several pieces
    r, c = rng.integers(1, 5001, n), rng.integers(1, 7001, n)
    if kind == "symmetric":
        c = np.minimum(c, 5000)
    v = rng.standard_normal(n) * 10.0 ** rng.integers(-30, 30, n)
    out = [f"%%MatrixMarket matrix coordinate {'pattern' if kind == 'pattern' else 'real'} {'symmetric' if kind == 'symmetric' else 'general'}",
           "% a comment", "", f"5000 {5000 if kind == 'symmetric' else 7000} {n}"]
    for i in range(n):
        if i % 9973 == 0:
            out.append("% a comment between the entries")
            out.append("")
        if kind == "pattern":
            out.append(f"{r[i]} {c[i]}")
        elif i % 5 == 0:
            out.append(f"  {r[i]}\t{c[i]}   +{float(abs(v[i]))!r}")   # blanks, a tab, a plus sign
        else:
            out.append(f"{r[i]} {c[i]} {v[i]:.17e}")
    text = ("\r\n" if kind == "general" else "\n").join(out) + "\n"
    path = tmp_path / f"{kind}.mtx"
    path.write_bytes(text.encode())
    nr, nc, nnz, rows, cols, vals = python_parse(text)
    want = f"{nr} {nc} {len(vals)} {checksum(rows, cols, vals)}"
    for threads in (1, 3, 16):
        res = run(reader, str(path), threads)
        assert res.returncode == 0, res.stderr
        assert res.stdout.strip() == want, f"{threads} thread(s)"


def test_reader_reads_the_reference_fixtures(reader, golden_dir):
    for name in ("ref_first/A.mtx", "ref_first/C_expected.mtx", "ref_test/A.mtx", "ref_test/B.mtx"):
        text = open(f"{golden_dir}/{name}").read()
        nr, nc, nnz, rows, cols, vals = python_parse(text)
        res = run(reader, f"{golden_dir}/{name}", 4)
        assert res.returncode == 0, res.stderr
        assert res.stdout.strip() == f"{nr} {nc} {len(vals)} {checksum(rows, cols, vals)}"


def test_reader_rejects_files_that_disagree_with_their_size_line(reader, tmp_path):
    head = "%%MatrixMarket matrix coordinate real general\n3 3 4\n"
    (tmp_path / "short.mtx").write_text(head + "1 1 1.0\n2 2 2.0\n3 3 3.0\n")
    (tmp_path / "long.mtx").write_text(head + "1 1 1.0\n2 2 2.0\n3 3 3.0\n1 2 4.0\n1 3 5.0\n")
    (tmp_path / "bad.mtx").write_text(head + "1 1 1.0\n2 x 2.0\n3 3 3.0\n1 2 4.0\n")
    (tmp_path / "nofinalnewline.mtx").write_text(head + "1 1 1.0\n2 2 2.0\n3 3 3.0\n1 2 4.5")
    for name, word in (("short", "truncated"), ("long", "more entry lines"), ("bad", "malformed")):
        res = run(reader, str(tmp_path / f"{name}.mtx"), 2)
        assert res.returncode == 3 and word in res.stderr, (name, res.stderr)
    res = run(reader, str(tmp_path / "nofinalnewline.mtx"), 2)
    assert res.returncode == 0 and res.stdout.split()[2] == "4"
