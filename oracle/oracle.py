"""Python face of the CPU oracle.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py may
import this module; the product package (spgemm_amd) never does.

The arithmetic lives in oracle/spgemm_oracle.c (plain C restatement of the
reference's Gustavson loop + Cells::add, of its Baseline merge-join loop and of
its section bitmap).  This file adds, in numpy, the restatement of the integer
set-up around the multiply:

  * partitions          reference src/parts/baseline.cpp:10-42 and :44-86
  * serialized sizes    reference include/matrix_impl.hpp:238-241 (CSR blob),
                        include/matrix.hpp:63-68 (BlockedFields),
                        include/matrix_impl.hpp:507-534 (filter)
  * the per-rank flow   reference src/main.cpp:213-217 (keep_rows / keep_cols),
                        src/mults/drop_parallel.cpp:43-59,72-118 (blocks visited
                        rank, rank-1, ... each adding partitions[s].start_col)
  * merge + un-shuffle  reference scripts/postprocessing.py:44-79
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile liboracle.so with gcc (make -C oracle)."""
    if force or not os.path.exists(_LIB_PATH) or (
        os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "spgemm_oracle.c"))
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s", "liboracle.so"])
    return _LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        vp = ctypes.c_void_p
        L.oracle_spgemm.restype = vp
        L.oracle_spgemm.argtypes = [ctypes.c_uint64, vp, vp, vp, vp, vp, vp, ctypes.c_int, ctypes.c_int]
        L.oracle_spgemm_spa.restype = vp
        L.oracle_spgemm_spa.argtypes = [ctypes.c_uint64, vp, vp, vp, ctypes.c_uint64, vp, vp, vp, ctypes.c_int]
        L.oracle_spgemm_blocks.restype = vp
        L.oracle_spgemm_blocks.argtypes = [ctypes.c_uint64, vp, vp, vp, ctypes.c_int, vp, vp, vp, vp,
                                           ctypes.c_int, ctypes.c_int]
        L.oracle_baseline_inner.restype = vp
        L.oracle_baseline_inner.argtypes = [ctypes.c_uint64, vp, vp, vp, ctypes.c_uint64, vp, vp, vp,
                                            ctypes.c_uint32]
        L.oracle_section_bitmap.restype = ctypes.c_int
        L.oracle_section_bitmap.argtypes = [ctypes.c_uint64, vp, ctypes.c_uint32, ctypes.c_uint32, vp]
        for f in ("rows", "nnz", "nprod"):
            getattr(L, "oracle_result_" + f).restype = ctypes.c_uint64
            getattr(L, "oracle_result_" + f).argtypes = [vp]
        L.oracle_result_seconds.restype = ctypes.c_double
        L.oracle_result_seconds.argtypes = [vp]
        L.oracle_result_fetch.restype = None
        L.oracle_result_fetch.argtypes = [vp, vp, vp, vp]
        L.oracle_result_free.restype = None
        L.oracle_result_free.argtypes = [vp]
        L.oracle_max_threads.restype = ctypes.c_int
        _lib = L
    return _lib


@dataclass
class OCsr:
    """Plain CSR holder: row_ptr uint64[rows+1], col_idx uint32[nnz], val float64[nnz]."""
    rows: int
    cols: int
    row_ptr: np.ndarray
    col_idx: np.ndarray
    val: np.ndarray

    @property
    def nnz(self) -> int:
        return int(self.row_ptr[-1])

    def canonical(self) -> "OCsr":
        """Rows sorted by column (stable)."""
        rows_of = np.repeat(np.arange(self.rows, dtype=np.int64), np.diff(self.row_ptr).astype(np.int64))
        order = np.lexsort((self.col_idx, rows_of))
        return OCsr(self.rows, self.cols, self.row_ptr.copy(), self.col_idx[order], self.val[order])

    def to_scipy(self):
        import scipy.sparse as sp
        return sp.csr_matrix((self.val, self.col_idx.astype(np.int64), self.row_ptr.astype(np.int64)),
                             shape=(self.rows, self.cols))

    @staticmethod
    def from_scipy(m) -> "OCsr":
        m = m.tocsr()
        return OCsr(m.shape[0], m.shape[1], m.indptr.astype(np.uint64), m.indices.astype(np.uint32),
                    m.data.astype(np.float64))

    @staticmethod
    def from_coo(rows: int, cols: int, r: np.ndarray, c: np.ndarray, v: np.ndarray) -> "OCsr":
        """CSR whose within-row order is the order of appearance in (r, c, v): the order the
        reference's single-threaded loader inserts in (include/matrix_impl.hpp:198-221)."""
        r = np.asarray(r, dtype=np.int64)
        order = np.argsort(r, kind="stable")
        counts = np.bincount(r, minlength=rows)
        row_ptr = np.zeros(rows + 1, dtype=np.uint64)
        row_ptr[1:] = np.cumsum(counts)
        return OCsr(rows, cols, row_ptr, np.asarray(c, dtype=np.uint32)[order],
                    np.asarray(v, dtype=np.float64)[order])

    def transpose(self) -> "OCsr":
        rows_of = np.repeat(np.arange(self.rows, dtype=np.int64), np.diff(self.row_ptr).astype(np.int64))
        return OCsr.from_coo(self.cols, self.rows, self.col_idx.astype(np.int64), rows_of, self.val)

    def select_rows(self, keep: np.ndarray) -> "OCsr":
        keep = np.asarray(keep, dtype=np.int64)
        lens = (self.row_ptr[keep + 1] - self.row_ptr[keep]).astype(np.int64)
        row_ptr = np.zeros(len(keep) + 1, dtype=np.uint64)
        row_ptr[1:] = np.cumsum(lens)
        idx = _ranges(self.row_ptr[keep].astype(np.int64), lens)
        return OCsr(len(keep), self.cols, row_ptr, self.col_idx[idx], self.val[idx])

    def select_cols(self, keep: np.ndarray) -> "OCsr":
        """Keep the listed original columns; a kept column gets its index in `keep` as new id
        (reference include/matrix_impl.hpp:212-217)."""
        keep = np.asarray(keep, dtype=np.int64)
        remap = np.full(self.cols, -1, dtype=np.int64)
        remap[keep] = np.arange(len(keep))
        newc = remap[self.col_idx.astype(np.int64)]
        m = newc >= 0
        rows_of = np.repeat(np.arange(self.rows, dtype=np.int64), np.diff(self.row_ptr).astype(np.int64))
        return OCsr.from_coo(self.rows, len(keep), rows_of[m], newc[m], self.val[m])


def _ranges(starts: np.ndarray, lens: np.ndarray) -> np.ndarray:
    total = int(lens.sum())
    if total == 0:
        return np.zeros(0, dtype=np.int64)
    out = np.ones(total, dtype=np.int64)
    nz = lens > 0
    s, l = starts[nz], lens[nz]
    ends = np.cumsum(l)
    out[0] = s[0]
    out[ends[:-1]] = s[1:] - (s[:-1] + l[:-1] - 1)
    return np.cumsum(out)


def _p(a: np.ndarray):
    return a.ctypes.data_as(ctypes.c_void_p)


def _take(handle, cols: int) -> Tuple[OCsr, dict]:
    L = lib()
    rows = L.oracle_result_rows(handle)
    nnz = L.oracle_result_nnz(handle)
    row_ptr = np.empty(rows + 1, dtype=np.uint64)
    col_idx = np.empty(nnz, dtype=np.uint32)
    val = np.empty(nnz, dtype=np.float64)
    L.oracle_result_fetch(handle, _p(row_ptr), _p(col_idx), _p(val))
    stats = {"nprod": int(L.oracle_result_nprod(handle)), "seconds": float(L.oracle_result_seconds(handle))}
    L.oracle_result_free(handle)
    return OCsr(int(rows), cols, row_ptr, col_idx, val), stats


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def spgemm(A: OCsr, B: OCsr, sort_cols: bool = True, nthreads: int = 0, with_stats: bool = False):
    """C = A * B by the reference's Gustavson loop + Cells::add."""
    assert A.cols == B.rows
    ap, ac, av = _c(A.row_ptr, np.uint64), _c(A.col_idx, np.uint32), _c(A.val, np.float64)
    bp, bc, bv = _c(B.row_ptr, np.uint64), _c(B.col_idx, np.uint32), _c(B.val, np.float64)
    h = lib().oracle_spgemm(A.rows, _p(ap), _p(ac), _p(av), _p(bp), _p(bc), _p(bv), int(sort_cols), nthreads)
    C, st = _take(h, B.cols)
    return (C, st) if with_stats else C


def spgemm_spa(A: OCsr, B: OCsr, nthreads: int = 0, with_stats: bool = False):
    """C = A * B by a tuned CPU Gustavson with a dense accumulator per thread (not the reference's
    accumulator: the second, fairer line of the CPU baseline).  Keeps an entry iff its value != 0."""
    assert A.cols == B.rows
    ap, ac, av = _c(A.row_ptr, np.uint64), _c(A.col_idx, np.uint32), _c(A.val, np.float64)
    bp, bc, bv = _c(B.row_ptr, np.uint64), _c(B.col_idx, np.uint32), _c(B.val, np.float64)
    h = lib().oracle_spgemm_spa(A.rows, _p(ap), _p(ac), _p(av), B.cols, _p(bp), _p(bc), _p(bv), nthreads)
    C, st = _take(h, B.cols)
    return (C, st) if with_stats else C


def spgemm_blocks(A: OCsr, blocks: Sequence[OCsr], col_offsets: Sequence[int], total_cols: int,
                  sort_cols: bool = True, nthreads: int = 0) -> OCsr:
    """The multi-block accumulate of one rank: every block adds its own column offset."""
    n = len(blocks)
    ap, ac, av = _c(A.row_ptr, np.uint64), _c(A.col_idx, np.uint32), _c(A.val, np.float64)
    keep = [(_c(b.row_ptr, np.uint64), _c(b.col_idx, np.uint32), _c(b.val, np.float64)) for b in blocks]
    arr = ctypes.c_void_p * n
    P = arr(*[k[0].ctypes.data for k in keep])
    Cc = arr(*[k[1].ctypes.data for k in keep])
    V = arr(*[k[2].ctypes.data for k in keep])
    off = _c(np.asarray(col_offsets), np.uint32)
    h = lib().oracle_spgemm_blocks(A.rows, _p(ap), _p(ac), _p(av), n, ctypes.cast(P, ctypes.c_void_p),
                                   ctypes.cast(Cc, ctypes.c_void_p), ctypes.cast(V, ctypes.c_void_p), _p(off),
                                   int(sort_cols), nthreads)
    C, _ = _take(h, total_cols)
    return C


def baseline_inner(A: OCsr, Bt: OCsr, col_offset: int = 0, total_cols: Optional[int] = None) -> OCsr:
    """Baseline operator's merge-join multiply; Bt is the transposed B block (sorted rows)."""
    ap, ac, av = _c(A.row_ptr, np.uint64), _c(A.col_idx, np.uint32), _c(A.val, np.float64)
    bp, bc, bv = _c(Bt.row_ptr, np.uint64), _c(Bt.col_idx, np.uint32), _c(Bt.val, np.float64)
    h = lib().oracle_baseline_inner(A.rows, _p(ap), _p(ac), _p(av), Bt.rows, _p(bp), _p(bc), _p(bv), col_offset)
    C, _ = _take(h, total_cols if total_cols is not None else col_offset + Bt.rows)
    return C


def section_bitmap(A: OCsr, n_sections: int) -> np.ndarray:
    """bool[n_sections]; reference src/bitmap.cpp:12-40."""
    nwords = (n_sections + 63) // 64
    words = np.zeros(nwords, dtype=np.uint64)
    ac = _c(A.col_idx, np.uint32)
    rc = lib().oracle_section_bitmap(A.nnz, _p(ac), A.cols, n_sections, _p(words))
    if rc != 0:
        raise ValueError("width / n_sections == 0 (the reference asserts section_width > 0)")
    bits = np.unpackbits(words.view(np.uint8), bitorder="little")[:n_sections]
    return bits.astype(bool)


# ---------------------------------------------------------------------------------------------
# integer set-up around the multiply (numpy restatement)
# ---------------------------------------------------------------------------------------------

Partition = Tuple[int, int, int, int]  # start_row, end_row, start_col, end_col


def partition_naive(height: int, width: int, P: int) -> List[Partition]:
    """reference src/parts/baseline.cpp:10-42."""
    out = []
    rpp, er = height // P, height % P
    cpp, ec = width // P, width % P
    cr = cc = 0
    for i in range(P):
        tr = rpp + (1 if i < er else 0)
        tc = cpp + (1 if i < ec else 0)
        out.append((cr, cr + tr, cc, cc + tc))
        cr += tr
        cc += tc
    return out


def partition_balanced(row_lens: np.ndarray, width: int, P: int) -> List[Partition]:
    """reference src/parts/baseline.cpp:44-86: cut rows when the running count of the rows'
    nnz(C_sparsity) exceeds nnz/P; columns are split naively.  The reference writes one
    partition past the end when the last cut lands on node P-1 (:62-66); here the loop stops
    before that write, which is what its `break` intends."""
    height = len(row_lens)
    nnz = int(np.sum(row_lens))
    per = nnz // P
    starts = [0] * P
    ends = [0] * P
    node = 0
    cnt = 0
    for row in range(height):
        cnt += int(row_lens[row])
        if cnt > per:
            cnt = 0
            ends[node] = row
            if node + 1 == P:
                node += 1
                break
            starts[node + 1] = row
            node += 1
    if node < P:
        ends[node] = height
        for j in range(node + 1, P):
            starts[j] = ends[j] = height
    else:
        ends[P - 1] = height
    cols = partition_naive(0, width, P)
    return [(starts[i], ends[i], cols[i][2], cols[i][3]) for i in range(P)]


def round8(x: int) -> int:
    return (x + 7) & ~7


def csr_blob_size(height: int, nnz: int) -> int:
    """reference include/matrix_impl.hpp:238-241: ROUND8(16 + 4(h+1) + 4 nnz + 8 nnz)."""
    return round8(16 + 4 * (height + 1) + 12 * nnz)


def blocked_fields_size(n_sections: int) -> int:
    """reference include/matrix.hpp:63-68: two size_t + two size_t[N] arrays."""
    return 16 + 16 * n_sections


def section_bounds(height: int, n_sections: int) -> np.ndarray:
    """Row bounds of the N row-sections of a B block, reference include/matrix_impl.hpp:628-639:
    section i = [i*(h/N), (i+1)*(h/N)), the last one runs to h."""
    sh = height // n_sections
    b = np.arange(n_sections + 1, dtype=np.int64) * sh
    b[-1] = height
    return b


def blocked_section_sizes(Bblk: OCsr, n_sections: int) -> np.ndarray:
    b = section_bounds(Bblk.rows, n_sections)
    rp = Bblk.row_ptr.astype(np.int64)
    return np.array([csr_blob_size(int(b[i + 1] - b[i]), int(rp[b[i + 1]] - rp[b[i]]))
                     for i in range(n_sections)], dtype=np.int64)


def filter_size(Bblk: OCsr, n_sections: int, bitmap: np.ndarray) -> int:
    """Byte size of BlockedCSRMatrix::filter(bitmap), reference include/matrix_impl.hpp:507-534."""
    sizes = blocked_section_sizes(Bblk, n_sections)
    return blocked_fields_size(n_sections) + int(sizes[np.asarray(bitmap, dtype=bool)].sum())


@dataclass
class RankResult:
    C: OCsr                 # local rows, shuffled-global columns
    bitmap: Optional[np.ndarray]
    part_A: OCsr
    block_B: OCsr           # this rank's B column block (all rows x its columns, relabelled)


def distributed_reference(A: OCsr, B: OCsr, P: int, partitions: Sequence[Partition],
                          A_shuffle: np.ndarray, B_shuffle: np.ndarray,
                          n_sections: Optional[int] = None, nthreads: int = 0) -> List[RankResult]:
    """What the P ranks of the reference compute (Gustavson operators).

    rank r: keep_rows = A_shuffle[start_row:end_row], keep_cols = B_shuffle[start_col:end_col]
    (src/main.cpp:213-217); it multiplies its A rows with every rank's B column block, visiting
    blocks in the order r, r-1, ... (src/mults/drop_parallel.cpp:72-118), block s adding
    partitions[s].start_col to its column ids.
    """
    A_shuffle = np.asarray(A_shuffle, dtype=np.int64)
    B_shuffle = np.asarray(B_shuffle, dtype=np.int64)
    blocks = [B.select_cols(B_shuffle[p[2]:p[3]]) for p in partitions]
    total_cols = partitions[-1][3]
    out = []
    for r in range(P):
        p = partitions[r]
        part_A = A.select_rows(A_shuffle[p[0]:p[1]])
        order = [(r - i) % P for i in range(P)]
        C = spgemm_blocks(part_A, [blocks[s] for s in order], [partitions[s][2] for s in order], total_cols,
                          sort_cols=True, nthreads=nthreads)
        bm = section_bitmap(part_A, n_sections) if n_sections else None
        out.append(RankResult(C, bm, part_A, blocks[r]))
    return out


def merge_unshuffle(parts: Sequence[OCsr], partitions: Sequence[Partition], A_shuffle: np.ndarray,
                    B_shuffle: np.ndarray, n_rows: int, n_cols: int) -> OCsr:
    """reference scripts/postprocessing.py:44-79: row += start_row; row -> A_shuffle[row];
    col -> B_shuffle[col]; concatenate; canonical CSR."""
    A_shuffle = np.asarray(A_shuffle, dtype=np.int64)
    B_shuffle = np.asarray(B_shuffle, dtype=np.int64)
    rs, cs, vs = [], [], []
    for C, p in zip(parts, partitions):
        rows_of = np.repeat(np.arange(C.rows, dtype=np.int64), np.diff(C.row_ptr).astype(np.int64))
        rs.append(A_shuffle[rows_of + p[0]])
        cs.append(B_shuffle[C.col_idx.astype(np.int64)])
        vs.append(C.val)
    r = np.concatenate(rs) if rs else np.zeros(0, dtype=np.int64)
    c = np.concatenate(cs) if cs else np.zeros(0, dtype=np.int64)
    v = np.concatenate(vs) if vs else np.zeros(0)
    return OCsr.from_coo(n_rows, n_cols, r, c, v).canonical()


# ---------------------------------------------------------------------------------------------
# Matrix Market (coordinate real general), enough for the fixtures
# ---------------------------------------------------------------------------------------------

def iterative_shuffle(C: OCsr, iterations: int):
    """Iterative locality shuffle (reference src/partition.cpp:123-202 with its helpers :73-121),
    restated with a fixed number of iterations instead of the reference's wall-clock limit
    (:142-146), exact means (integer sum / count in float64; the reference adds float32 in storage
    order) and ties broken by the lower index (the reference's parallel sort is not stable).
    Even iterations re-rank the rows by the mean position of their columns, odd iterations the
    columns by the mean position of their rows; an empty row or column gets the mean `cols`
    (:83-84, :108-109).  Returns (row_shuffle, col_shuffle, row_variance): shuffle[i] = original
    index at position i (:187-199); row_variance[i] = variance of (mean - rank) after row
    iteration i (:169-178, with a true n / (n - 1) factor).  Parity of the permutation against the
    reference binary is unpinned (the reference's own result depends on run time and thread
    schedule); the GPU path is checked against this function."""
    rows, cols = C.rows, C.cols
    rp = C.row_ptr.astype(np.int64)
    ci = C.col_idx.astype(np.int64)
    rid = np.repeat(np.arange(rows, dtype=np.int64), np.diff(rp))
    pos_r = np.arange(rows, dtype=np.int64)
    pos_c = np.arange(cols, dtype=np.int64)
    len_r = np.diff(rp)
    len_c = np.bincount(ci, minlength=cols).astype(np.int64)
    variances = []
    for it in range(iterations):
        if it % 2 == 0:
            n, seg, other, lens = rows, rid, pos_c[ci], len_r
        else:
            n, seg, other, lens = cols, ci, pos_r[rid], len_c
        if n == 0:
            continue
        sums = np.zeros(n, dtype=np.int64)
        np.add.at(sums, seg, other)
        mean = np.full(n, float(cols), dtype=np.float64)
        nz = lens > 0
        mean[nz] = sums[nz].astype(np.float64) / lens[nz].astype(np.float64)
        order = np.argsort(mean, kind="stable")
        rank = np.empty(n, dtype=np.int64)
        rank[order] = np.arange(n, dtype=np.int64)
        if it % 2 == 0:
            pos_r = rank
            x = mean[order] - np.arange(n, dtype=np.float64)
            sx, sx2 = 0.0, 0.0
            for v in x:  # the same order of additions as the host loop of the library
                sx += v / n
                sx2 += v * v / n
            variances.append((sx2 - sx * sx) * (n / (n - 1)) if n > 1 else 0.0)
        else:
            pos_c = rank
    row_shuffle = np.empty(rows, dtype=np.uint32)
    row_shuffle[pos_r] = np.arange(rows, dtype=np.uint32)
    col_shuffle = np.empty(cols, dtype=np.uint32)
    col_shuffle[pos_c] = np.arange(cols, dtype=np.uint32)
    return row_shuffle, col_shuffle, np.array(variances)


def read_mtx(path: str) -> OCsr:
    with open(path, "r") as f:
        header = f.readline()
        if not header.lower().startswith("%%matrixmarket matrix coordinate"):
            raise ValueError("unsupported Matrix Market header: " + header.strip())
        symmetric = "symmetric" in header.lower()
        pattern = "pattern" in header.lower()
        line = f.readline()
        while line.startswith("%") or not line.strip():
            line = f.readline()
        rows, cols, nnz = (int(x) for x in line.split()[:3])
        data = np.loadtxt(f, ndmin=2) if nnz else np.zeros((0, 3))
    r = data[:, 0].astype(np.int64) - 1
    c = data[:, 1].astype(np.int64) - 1
    v = np.ones(len(r)) if pattern else data[:, 2].astype(np.float64)
    if symmetric:
        off = r != c
        r, c, v = np.concatenate([r, c[off]]), np.concatenate([c, r[off]]), np.concatenate([v, v[off]])
    return OCsr.from_coo(rows, cols, r, c, v)


def assert_same(C: OCsr, E: OCsr, rtol: float = 0.0):
    """Pattern bit-exact (row_ptr, col_idx after sort); values exact (rtol=0) or relative."""
    C, E = C.canonical(), E.canonical()
    assert (C.rows, C.cols) == (E.rows, E.cols), ((C.rows, C.cols), (E.rows, E.cols))
    assert np.array_equal(C.row_ptr.astype(np.uint64), E.row_ptr.astype(np.uint64)), "row_ptr differs"
    assert np.array_equal(C.col_idx.astype(np.uint32), E.col_idx.astype(np.uint32)), "col_idx differs"
    if rtol == 0.0:
        assert np.array_equal(C.val, E.val), "values differ (bitwise)"
    else:
        denom = np.maximum(np.abs(E.val), np.finfo(np.float64).tiny)
        err = np.abs(C.val - E.val) / denom
        assert err.size == 0 or err.max() <= rtol, f"max relative error {err.max()} > {rtol}"
