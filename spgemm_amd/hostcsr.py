"""Host-side CSR holder, Matrix Market I/O and the synthetic matrix generators used by the
tests and the benchmark.  Nothing here multiplies matrices.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Tuple

import numpy as np


@dataclass
class HostCsr:
    """row_ptr uint64[rows+1], col_idx uint32[nnz], val float64[nnz]; rows in storage order."""
    rows: int
    cols: int
    row_ptr: np.ndarray
    col_idx: np.ndarray
    val: np.ndarray

    @property
    def nnz(self) -> int:
        return int(self.row_ptr[-1])

    @staticmethod
    def from_coo(rows: int, cols: int, r, c, v, sum_duplicates: bool = False) -> "HostCsr":
        """Within a row the entries keep their order of appearance (the order the reference's
        loader inserts them in, include/matrix_impl.hpp:198-221)."""
        r = np.asarray(r, dtype=np.int64)
        c = np.asarray(c, dtype=np.int64)
        v = np.asarray(v, dtype=np.float64)
        if sum_duplicates and len(r):
            key = r * np.int64(cols) + c
            order = np.argsort(key, kind="stable")
            key, v = key[order], v[order]
            first = np.concatenate(([True], key[1:] != key[:-1]))
            seg = np.cumsum(first) - 1
            v = np.bincount(seg, weights=v)  # sums in storage order within a duplicate run
            key = key[first]
            r, c = key // cols, key % cols
        order = np.argsort(r, kind="stable")
        counts = np.bincount(r, minlength=rows)
        row_ptr = np.zeros(rows + 1, dtype=np.uint64)
        row_ptr[1:] = np.cumsum(counts)
        return HostCsr(rows, cols, row_ptr, c[order].astype(np.uint32), v[order].astype(np.float64))

    def row_ids(self) -> np.ndarray:
        return np.repeat(np.arange(self.rows, dtype=np.int64), np.diff(self.row_ptr).astype(np.int64))

    def sorted_rows(self) -> "HostCsr":
        order = np.lexsort((self.col_idx, self.row_ids()))
        return HostCsr(self.rows, self.cols, self.row_ptr.copy(), self.col_idx[order], self.val[order])


# ---- Matrix Market (coordinate) -----------------------------------------------------------------------

def read_mtx(path: str, eliminate_zeros: bool = False) -> HostCsr:
    """Reads `coordinate real|integer|pattern general|symmetric`.  Symmetric files are expanded
    (what the reference's fetch script does before it stores A.mtx,
    scripts/fetch_matrices.py:116-137)."""
    with open(path, "r") as f:
        header = f.readline().lower()
        if not header.startswith("%%matrixmarket matrix coordinate"):
            raise ValueError("unsupported Matrix Market header: " + header.strip())
        symmetric = "symmetric" in header
        pattern = "pattern" in header
        pos = f.tell()
        line = f.readline()
        while line.startswith("%") or not line.strip():
            pos = f.tell()
            line = f.readline()
        rows, cols, nnz = (int(x) for x in line.split()[:3])
        try:
            import pandas as pd
            # round_trip: the default fast float parser is off by an ulp now and then
            df = pd.read_csv(f, sep=r"\s+", header=None, comment="%", engine="c",
                             dtype={0: np.int64, 1: np.int64}, float_precision="round_trip")
            data = df.to_numpy(dtype=np.float64) if len(df) else np.zeros((0, 3))
        except ImportError:
            data = np.loadtxt(f, ndmin=2) if nnz else np.zeros((0, 3))
    r = data[:, 0].astype(np.int64) - 1
    c = data[:, 1].astype(np.int64) - 1
    v = np.ones(len(r)) if pattern else data[:, 2].astype(np.float64)
    if len(r) != nnz:
        raise ValueError(f"{path}: header says {nnz} entries, file holds {len(r)}")
    if symmetric:
        off = r != c
        r, c, v = np.concatenate([r, c[off]]), np.concatenate([c, r[off]]), np.concatenate([v, v[off]])
    if eliminate_zeros:
        keep = v != 0
        r, c, v = r[keep], c[keep], v[keep]
    return HostCsr.from_coo(rows, cols, r, c, v)


def write_mtx(path: str, m: HostCsr, one_based_sorted: bool = True):
    """`coordinate real general`, entries sorted by (row, col) as the reference writes them
    (src/matrix.cpp:46-64), but with 17 significant digits instead of the stream default of 6."""
    s = m.sorted_rows() if one_based_sorted else m
    with open(path, "w") as f:
        f.write("%%MatrixMarket matrix coordinate real general\n")
        f.write(f"{s.rows} {s.cols} {s.nnz}\n")
        rid = s.row_ids() + 1
        cid = s.col_idx.astype(np.int64) + 1
        for i in range(0, s.nnz, 1 << 20):
            j = min(s.nnz, i + (1 << 20))
            f.write("".join(f"{a} {b} {x:.17g}\n" for a, b, x in zip(rid[i:j], cid[i:j], s.val[i:j])))


# ---- synthetic inputs --------------------------------------------------------------------------------------

def erdos_renyi(n: int, per_row: int = 16, seed: int = 42) -> HostCsr:
    """n x n, exactly `per_row` draws per row, columns uniform in [0, n), duplicates summed,
    values uniform in [0.5, 1.5); rows sorted by column."""
    rng = np.random.Generator(np.random.PCG64(seed))
    cols = rng.integers(0, n, size=(n, per_row), dtype=np.int64)
    vals = rng.random((n, per_row)) + 0.5
    order = np.argsort(cols, axis=1, kind="stable")
    cols = np.take_along_axis(cols, order, axis=1)
    vals = np.take_along_axis(vals, order, axis=1)
    first = np.ones((n, per_row), dtype=bool)
    first[:, 1:] = cols[:, 1:] != cols[:, :-1]
    counts = first.sum(axis=1)
    row_ptr = np.zeros(n + 1, dtype=np.uint64)
    row_ptr[1:] = np.cumsum(counts)
    seg = np.cumsum(first.ravel()) - 1
    v = np.bincount(seg, weights=vals.ravel())
    c = cols.ravel()[first.ravel()]
    return HostCsr(n, n, row_ptr, c.astype(np.uint32), v.astype(np.float64))


def rmat(scale: int, edge_factor: int = 4, a: float = 0.57, b: float = 0.19, c: float = 0.19,
         seed: int = 1, unit_values: bool = False) -> HostCsr:
    """R-MAT (Graph500 parameters by default), n = 2^scale, edge_factor * n edge draws, duplicates
    merged (their values are summed); values 1.0 or uniform in [0.5, 1.5); rows sorted by column."""
    n = 1 << scale
    m = edge_factor * n
    rng = np.random.Generator(np.random.PCG64(seed))
    r = np.zeros(m, dtype=np.int64)
    cc = np.zeros(m, dtype=np.int64)
    ab, abc = a + b, a + b + c
    for _ in range(scale):
        u = rng.random(m)
        right = ((u >= a) & (u < ab)) | (u >= abc)
        down = u >= ab
        r = (r << 1) | down
        cc = (cc << 1) | right
    v = np.ones(m) if unit_values else rng.random(m) + 0.5
    return HostCsr.from_coo(n, n, r, cc, v, sum_duplicates=True)


def shell_like(n_nodes: int = 301_613, grid_w: int = 549, dofs: int = 5, seed: int = 7) -> HostCsr:
    """Stand-in for SuiteSparse Schenk_AFE/af_shell10 (not downloadable here): the same order
    (5 * 301,613 = 1,508,065) and about the same entries per row (~35).  Nodes of a structured
    triangulated sheet, `dofs` unknowns per node, every node coupled to itself and its up to six
    mesh neighbours by dense dofs x dofs blocks; values uniform in [-1, 1) with a dominant
    diagonal; rows sorted by column."""
    N, W = n_nodes, grid_w
    p = np.arange(N, dtype=np.int64)
    x, y = p % W, p // W
    offs = [(-1, -1), (0, -1), (-1, 0), (0, 0), (1, 0), (0, 1), (1, 1)]  # ascending node index
    nb = np.full((N, len(offs)), -1, dtype=np.int64)
    for t, (dx, dy) in enumerate(offs):
        qx, qy = x + dx, y + dy
        q = qy * W + qx
        ok = (qx >= 0) & (qx < W) & (qy >= 0) & (q < N)
        nb[ok, t] = q[ok]
    return _fem_matrix(nb, dofs, seed)


def _fem_matrix(nb: np.ndarray, dofs: int, seed: int) -> HostCsr:
    """nb[p, t] = t-th neighbour of node p (ascending node index, -1 where there is none, the node
    itself included): every node pair is coupled by a dense dofs x dofs block; values uniform in
    [-1, 1) with a dominant diagonal; rows sorted by column."""
    N = nb.shape[0]
    valid = nb >= 0
    cnt = valid.sum(axis=1)                                  # neighbours per node
    row_len_node = cnt * dofs
    n = N * dofs
    row_len = np.repeat(row_len_node, dofs)
    row_ptr = np.zeros(n + 1, dtype=np.uint64)
    row_ptr[1:] = np.cumsum(row_len)
    # columns of one node row: for every valid neighbour q (ascending) the dofs columns of q
    nb_flat = nb[valid]                                      # node-major, ascending inside a node
    cols_node = (nb_flat[:, None] * dofs + np.arange(dofs)[None, :]).ravel()  # per node: cnt*dofs
    node_ptr = np.zeros(N + 1, dtype=np.int64)
    node_ptr[1:] = np.cumsum(row_len_node)
    # repeat each node's column list `dofs` times (one per dof row)
    starts = np.repeat(node_ptr[:-1], dofs)
    lens = np.repeat(row_len_node, dofs)
    total = int(lens.sum())
    idx = np.ones(total, dtype=np.int64)
    ends = np.cumsum(lens)
    idx[0] = starts[0]
    idx[ends[:-1]] = starts[1:] - (starts[:-1] + lens[:-1] - 1)
    idx = np.cumsum(idx)
    col_idx = cols_node[idx].astype(np.uint32)
    rng = np.random.Generator(np.random.PCG64(seed))
    val = rng.random(total) * 2.0 - 1.0
    rows_of = np.repeat(np.arange(n, dtype=np.int64), lens)
    val[col_idx == rows_of] += 10.0
    return HostCsr(n, n, row_ptr, col_idx, val)


def diel_like(nx: int = 71, ny: int = 72, nz: int = 72, dofs: int = 3, seed: int = 11) -> HostCsr:
    """Stand-in for SuiteSparse Dziekonski/dielFilterV3real (not downloadable here): about the same
    order (1,102,824; here 3 * 71 * 72 * 72 = 1,104,192) and entries per row (~81).  Nodes of a
    structured 3-D grid, `dofs` unknowns per node, every node coupled to the 27 nodes of its
    3 x 3 x 3 neighbourhood by dense dofs x dofs blocks."""
    N = nx * ny * nz
    p = np.arange(N, dtype=np.int64)
    x, y, z = p % nx, (p // nx) % ny, p // (nx * ny)
    offs = [(dx, dy, dz) for dz in (-1, 0, 1) for dy in (-1, 0, 1) for dx in (-1, 0, 1)]  # ascending node index
    nb = np.full((N, len(offs)), -1, dtype=np.int64)
    for t, (dx, dy, dz) in enumerate(offs):
        qx, qy, qz = x + dx, y + dy, z + dz
        ok = (qx >= 0) & (qx < nx) & (qy >= 0) & (qy < ny) & (qz >= 0) & (qz < nz)
        nb[ok, t] = (qx + nx * (qy + ny * qz))[ok]
    return _fem_matrix(nb, dofs, seed)


def banded_random(n: int, per_row: int, half_band: int, seed: int = 3, signed: bool = True) -> HostCsr:
    """Small test generator: `per_row` draws per row inside [i - half_band, i + half_band]."""
    rng = np.random.Generator(np.random.PCG64(seed))
    r = np.repeat(np.arange(n, dtype=np.int64), per_row)
    c = np.clip(r + rng.integers(-half_band, half_band + 1, size=len(r)), 0, n - 1)
    v = rng.random(len(r)) * (2.0 if signed else 1.0) - (1.0 if signed else -0.5)
    return HostCsr.from_coo(n, n, r, c, v, sum_duplicates=True)
