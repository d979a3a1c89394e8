"""Row-block distributed C = A * B on one node: one process per GPU, torch.distributed for the
exchange (backend "nccl" = RCCL over xGMI on GPUs; "gloo" on CPU for the plumbing tests).

What is kept from the reference (src/main.cpp:213-217, src/mults/drop_at_once_parallel.cpp):
  * rank r owns the rows A_shuffle[start_row:end_row] of A, in that order, and produces the same
    rows of C with local row ids and shuffled-global column ids (column p stands for original
    column B_shuffle[p]); the inner dimension k is never permuted;
  * the 512-bit section bitmap of src/bitmap.cpp decides which row sections of B a rank needs
    (`drop`); without it every section is sent (`baseline`); `shuffle` applies a seeded random
    row/column permutation first.

What is different, on purpose: B is distributed by ROW SECTIONS (section s = rows
[s*(n/N), (s+1)*(n/N)), last one to n), not by column blocks.  Rank r initially owns the sections
[r*N/P, (r+1)*N/P).  One all-to-all moves, to every rank, exactly the sections its bitmap marks;
the receiver's buffers ARE the col_idx/val arrays of its local B (sections arrive in ascending
order), so it runs ONE full-width Gustavson multiply instead of P partial ones.
"""
from __future__ import annotations

import json
import os
import time
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

from .hostcsr import HostCsr

Partition = Tuple[int, int, int, int]


# ---- host-side set-up (integer work, numpy) ---------------------------------------------------------

def section_bounds(n: int, n_sections: int) -> np.ndarray:
    """Row bounds of the row sections of a matrix with n rows (reference
    include/matrix_impl.hpp:628-639): all sections have n // N rows, the last one the rest."""
    sh = n // n_sections
    if sh == 0:
        raise ValueError("fewer rows than sections (the reference asserts section_width > 0)")
    b = np.arange(n_sections + 1, dtype=np.int64) * sh
    b[-1] = n
    return b


def section_owner_bounds(n_sections: int, world: int) -> np.ndarray:
    """Rank r owns sections [b[r], b[r+1])."""
    return (np.arange(world + 1, dtype=np.int64) * n_sections) // world


def make_shuffles(n_rows: int, n_cols: int, impl: str, seed: int) -> Tuple[np.ndarray, np.ndarray]:
    """Row / column permutations of a run: `shuffle` = seeded random (reference src/partition.cpp:15-37,
    which is unseeded), anything else the identity."""
    from . import capi
    if impl == "shuffle":
        return capi.shuffle_random(n_rows, seed).astype(np.int64), capi.shuffle_random(n_cols, seed + 1).astype(np.int64)
    return np.arange(n_rows, dtype=np.int64), np.arange(n_cols, dtype=np.int64)


def row_products(A: HostCsr, B: HostCsr) -> np.ndarray:
    """Products per row of A * B: the sum of the lengths of the B rows its entries select."""
    len_b = np.diff(B.row_ptr.astype(np.int64))
    w = np.zeros(A.rows, dtype=np.int64)
    np.add.at(w, A.row_ids(), len_b[A.col_idx.astype(np.int64)])
    return w.astype(np.uint64)


def product_rows_are_narrow(A: HostCsr, B: HostCsr, max_span: int = 65536) -> bool:
    """True when most rows of A * B span at most max_span columns (column hull of the B rows a row
    of A selects), i.e. when B has column locality that a relabelling of B would destroy.  Decides
    where a column shuffle is applied: to the result (B untouched) or to B."""
    if A.nnz == 0 or B.nnz == 0:
        return False
    rp = B.row_ptr.astype(np.int64)
    has = rp[1:] > rp[:-1]
    bmin = np.full(B.rows, np.iinfo(np.int64).max, dtype=np.int64)
    bmax = np.full(B.rows, -1, dtype=np.int64)
    ci = B.col_idx.astype(np.int64)
    bmin[has] = np.minimum.reduceat(ci, rp[:-1][has])
    bmax[has] = np.maximum.reduceat(ci, rp[:-1][has])
    k = A.col_idx.astype(np.int64)
    arp = A.row_ptr.astype(np.int64)
    ahas = arp[1:] > arp[:-1]
    lo = np.minimum.reduceat(bmin[k], arp[:-1][ahas])
    hi = np.maximum.reduceat(bmax[k], arp[:-1][ahas])
    span = hi - lo + 1
    live = hi >= 0
    return bool(live.any()) and float(np.mean(span[live] <= max_span)) >= 0.5


def iterative_shuffles(ctx, A: HostCsr, B: HostCsr, iterations: int = 20) -> Tuple[np.ndarray, np.ndarray]:
    """The reference's `iterative` shuffling (src/main.cpp:151-152, src/partition.cpp:123-202): the
    pattern of C is computed here on the device (the reference reads C_sparsity.mtx)."""
    dA = ctx.from_host(A.rows, A.cols, A.row_ptr, A.col_idx, A.val)
    dB = dA if B is A else ctx.from_host(B.rows, B.cols, B.row_ptr, B.col_idx, B.val)
    C = ctx.multiply(dA, dB)
    try:
        rs, cs, _ = ctx.shuffle_iterative(C, iterations)
    finally:
        C.free()
        if dB is not dA:
            dB.free()
        dA.free()
    return rs.astype(np.int64), cs.astype(np.int64)


def select_rows(M: HostCsr, keep: np.ndarray) -> HostCsr:
    keep = np.asarray(keep, dtype=np.int64)
    starts = M.row_ptr[keep].astype(np.int64)
    lens = (M.row_ptr[keep + 1] - M.row_ptr[keep]).astype(np.int64)
    row_ptr = np.zeros(len(keep) + 1, dtype=np.uint64)
    row_ptr[1:] = np.cumsum(lens)
    total = int(lens.sum())
    if total == 0:
        return HostCsr(len(keep), M.cols, row_ptr, np.zeros(0, np.uint32), np.zeros(0))
    idx = np.ones(total, dtype=np.int64)
    nz = lens > 0
    s, l = starts[nz], lens[nz]
    ends = np.cumsum(l)
    idx[0] = s[0]
    idx[ends[:-1]] = s[1:] - (s[:-1] + l[:-1] - 1)
    idx = np.cumsum(idx)
    return HostCsr(len(keep), M.cols, row_ptr, M.col_idx[idx], M.val[idx])


def relabel_cols(M: HostCsr, new_id: np.ndarray, ctx=None) -> HostCsr:
    """Column c becomes new_id[c]; rows are re-sorted by the new ids.  With a device context the
    renaming and the per-row sort run on the GPU (spgemm_mat_select: a segmented radix sort, seconds
    instead of a minute at 67 M entries); the result is the same."""
    if ctx is not None and M.nnz:
        d = ctx.from_host(M.rows, M.cols, M.row_ptr, M.col_idx, M.val)
        try:
            s = ctx.select(d, col_new_id=np.asarray(new_id).astype(np.uint32), new_cols=M.cols, sort_rows=True)
            try:
                rp, ci, v = s.to_host()
            finally:
                s.free()
        finally:
            d.free()
        return HostCsr(M.rows, M.cols, rp, ci, v)
    c = new_id[M.col_idx.astype(np.int64)]
    order = np.lexsort((c, M.row_ids()))
    return HostCsr(M.rows, M.cols, M.row_ptr.copy(), c[order].astype(np.uint32), M.val[order])


@dataclass
class RankOperands:
    A_part: HostCsr       # my rows of A (shuffled order), original column ids (= k)
    B_block: HostCsr      # my row sections of B, columns relabelled to shuffled-global ids
    first_row: int        # first global row (k) of B_block
    n: int                # rows of B / inner dimension
    n_cols: int           # columns of C


def build_rank_operands(A: HostCsr, B: HostCsr, rank: int, world: int, partitions: Sequence[Partition],
                        A_shuffle: np.ndarray, B_shuffle: np.ndarray, n_sections: int,
                        relabel_b: bool = True, ctx=None) -> RankOperands:
    """relabel_b=False leaves the columns of the B block as they are: the caller renames the columns
    of the RESULT instead (same entries, same values; B keeps its locality)."""
    p = partitions[rank]
    A_part = select_rows(A, A_shuffle[p[0]:p[1]])
    sb = section_bounds(B.rows, n_sections)
    ob = section_owner_bounds(n_sections, world)
    r0, r1 = int(sb[ob[rank]]), int(sb[ob[rank + 1]])
    blk = select_rows(B, np.arange(r0, r1, dtype=np.int64))
    inv = np.empty(B.cols, dtype=np.int64)
    inv[B_shuffle] = np.arange(B.cols, dtype=np.int64)
    if relabel_b and not np.array_equal(B_shuffle, np.arange(B.cols)):
        blk = relabel_cols(blk, inv, ctx=ctx)
    return RankOperands(A_part, blk, r0, B.rows, B.cols)


# ---- the exchange (torch tensors, any device) -----------------------------------------------------------

def _staged(device, group=None) -> bool:
    """True when the collectives must go through host memory: the process group is gloo (which
    has no device all-to-all) while the data lives on a GPU.  This is the rehearsal mode that runs
    several ranks on one GPU; with the "nccl" backend (RCCL) nothing is staged."""
    import torch
    import torch.distributed as dist
    return torch.device(device).type == "cuda" and dist.get_backend(group) == "gloo"


def _broadcast0(t_cpu, device, group=None):
    """Broadcast of a host tensor from rank 0 (through the device when the backend needs it)."""
    import torch.distributed as dist
    if dist.get_backend(group) == "gloo":
        dist.broadcast(t_cpu, src=0, group=group)
        return t_cpu
    d = t_cpu.to(device)
    dist.broadcast(d, src=0, group=group)
    return d.cpu()


def _all_reduce(t, op=None, group=None):
    """all_reduce of a small tensor on whatever device the backend can reach; returns the result
    on t's device."""
    import torch.distributed as dist
    kw = {} if op is None else {"op": op}
    if _staged(t.device, group):
        h = t.cpu()
        dist.all_reduce(h, group=group, **kw)
        return h.to(t.device)
    dist.all_reduce(t, group=group, **kw)
    return t


class SectionExchange:
    """All-to-all of B row sections.  Holds this rank's B block and, from the gathered bitmaps and
    the global per-section entry counts, the split sizes of the collective.
    `run()` returns (row_ptr int64[n+1], col int32[nnz], val float64[nnz]) of the local B.

    With a context (`ctx`, a GPU) the sections are packed by spgemm_pack_sections and the row pointer
    of the received B is built by spgemm_assemble_row_ptr (the allocation-free variant when B has at
    most 2^22 rows): the same device helpers the C++ driver uses, all queued on the context's stream,
    in buffers allocated once.  Without one (the CPU plumbing tests over gloo) torch does both."""

    def __init__(self, ops: RankOperands, n_sections: int, bitmaps: np.ndarray, device, group=None, share=None,
                 ctx=None):
        """share: another SectionExchange over the same B block, whose device copies of the block
        are used instead of making new ones (several exchanges with different bitmaps)."""
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.n, self.N, self.device = ops.n, n_sections, device
        self.first_row = ops.first_row
        self.ctx = ctx
        self.bitmaps = np.asarray(bitmaps, dtype=bool)            # [world, N]
        assert self.bitmaps.shape == (self.world, n_sections)
        self.sb = section_bounds(ops.n, n_sections)
        self.ob = section_owner_bounds(n_sections, self.world)
        blk = ops.B_block
        if share is not None:
            self.row_len, self.col, self.val, self.dB = share.row_len, share.col, share.val, share.dB
        else:
            self.row_len = torch.from_numpy(np.diff(blk.row_ptr.astype(np.int64)).astype(np.int32)).to(device)
            self.col = torch.from_numpy(blk.col_idx.view(np.int32).copy()).to(device)
            self.val = torch.from_numpy(blk.val.copy()).to(device)
            self.dB = None
            if ctx is not None:
                self.blk_ptr = torch.from_numpy(blk.row_ptr.astype(np.int64)).to(device)
                torch.cuda.synchronize(device)
                self.dB = ctx.wrap_device(blk.rows, blk.cols, blk.nnz, self.blk_ptr.data_ptr(), self.col.data_ptr(),
                                          self.val.data_ptr(), validate=False,
                                          keepalive=(self.blk_ptr, self.col, self.val))
        # entries per section, globally: every section is owned by exactly one rank
        my_lo, my_hi = int(self.ob[self.rank]), int(self.ob[self.rank + 1])
        rp = blk.row_ptr.astype(np.int64)
        sec_nnz = np.zeros(n_sections, dtype=np.int64)
        loc = self.sb[my_lo:my_hi + 1] - ops.first_row
        sec_nnz[my_lo:my_hi] = rp[loc[1:]] - rp[loc[:-1]]
        self.staged = _staged(device, group)
        t = _all_reduce(torch.from_numpy(sec_nnz).to(device), group=group)
        self.sec_nnz = t.cpu().numpy()
        self.sec_rows = np.diff(self.sb)
        self.local_nnz_start = rp[loc[:-1]]                       # start of my sections in my arrays
        self.local_row_start = loc[:-1]
        # what I send to each destination: runs of my sections that the destination marks
        self.send_runs: List[List[Tuple[int, int, int, int]]] = []  # per dest: (row0,row1,nnz0,nnz1)
        self.send_rows, self.send_nnz = [], []
        for d in range(self.world):
            runs, rows, nnz = [], 0, 0
            s = my_lo
            while s < my_hi:
                if not self.bitmaps[d, s]:
                    s += 1
                    continue
                e = s
                while e + 1 < my_hi and self.bitmaps[d, e + 1]:
                    e += 1
                r0, r1 = int(loc[s - my_lo]), int(loc[e - my_lo + 1])
                z0, z1 = int(rp[r0]), int(rp[r1])
                runs.append((r0, r1, z0, z1))
                rows += r1 - r0
                nnz += z1 - z0
                s = e + 1
            self.send_runs.append(runs)
            self.send_rows.append(rows)
            self.send_nnz.append(nnz)
        # what I receive from each source: my marked sections among the source's
        mine = self.bitmaps[self.rank]
        self.recv_rows = [int(self.sec_rows[self.ob[s]:self.ob[s + 1]][mine[self.ob[s]:self.ob[s + 1]]].sum())
                          for s in range(self.world)]
        self.recv_nnz = [int(self.sec_nnz[self.ob[s]:self.ob[s + 1]][mine[self.ob[s]:self.ob[s + 1]]].sum())
                         for s in range(self.world)]
        # rows of the full-height B that are present locally (ascending)
        present = np.zeros(self.n, dtype=bool)
        for s in np.flatnonzero(mine):
            present[self.sb[s]:self.sb[s + 1]] = True
        self.present_idx = torch.from_numpy(np.flatnonzero(present)).to(device)
        self.bytes_sent_per_run = sum(4 * r + 12 * z for d, (r, z) in enumerate(zip(self.send_rows, self.send_nnz))
                                      if d != self.rank)
        # what this rank would send if every peer took all of its sections (no bitmap)
        self.bytes_sent_without_drop = (self.world - 1) * (4 * int(blk.rows) + 12 * int(blk.nnz))
        if self.dB is not None:  # send / receive buffers of the device path, allocated once
            self.buf_send = (torch.empty(sum(self.send_rows), dtype=torch.int32, device=device),
                             torch.empty(sum(self.send_nnz), dtype=torch.int32, device=device),
                             torch.empty(sum(self.send_nnz), dtype=torch.float64, device=device))
            self.buf_recv = (torch.empty(sum(self.recv_rows), dtype=torch.int32, device=device),
                             torch.empty(sum(self.recv_nnz), dtype=torch.int32, device=device),
                             torch.empty(sum(self.recv_nnz), dtype=torch.float64, device=device),
                             torch.empty(self.n + 1, dtype=torch.int64, device=device))

    def _pack(self, src, runs_key: int):
        torch = self.torch
        parts = []
        for d in range(self.world):
            for (r0, r1, z0, z1) in self.send_runs[d]:
                parts.append(src[r0:r1] if runs_key == 0 else src[z0:z1])
        if not parts:
            return src[:0]
        return torch.cat(parts) if len(parts) > 1 else parts[0].contiguous()

    def _run_device(self):
        """Pack with spgemm_pack_sections, exchange, rebuild the row pointer with
        spgemm_assemble_row_ptr; everything is queued on the context's stream (which must be
        torch's current stream: the collectives are ordered on it).  The result lives in this object's
        receive buffers until its next run."""
        torch, dist = self.torch, self.dist
        send_len, send_col, send_val = self.buf_send
        ro = zo = 0
        for d in range(self.world):
            if self.send_rows[d]:
                r, z = self.ctx.pack_sections(self.dB, self.N, self.first_row, self.n, self.bitmaps[d],
                                              send_len.data_ptr() + 4 * ro, send_col.data_ptr() + 4 * zo,
                                              send_val.data_ptr() + 8 * zo)
                assert (r, z) == (self.send_rows[d], self.send_nnz[d])
                ro += r
                zo += z
        recv_len, recv_col, recv_val, row_ptr = self.buf_recv
        if self.staged:
            h = [x.cpu() for x in (send_len, send_col, send_val)]
            o = [torch.empty_like(x, device="cpu") for x in (recv_len, recv_col, recv_val)]
            dist.all_to_all_single(o[0], h[0], self.recv_rows, self.send_rows, group=self.group)
            dist.all_to_all_single(o[1], h[1], self.recv_nnz, self.send_nnz, group=self.group)
            dist.all_to_all_single(o[2], h[2], self.recv_nnz, self.send_nnz, group=self.group)
            recv_len.copy_(o[0]); recv_col.copy_(o[1]); recv_val.copy_(o[2])
        else:
            dist.all_to_all_single(recv_len, send_len, self.recv_rows, self.send_rows, group=self.group)
            dist.all_to_all_single(recv_col, send_col, self.recv_nnz, self.send_nnz, group=self.group)
            dist.all_to_all_single(recv_val, send_val, self.recv_nnz, self.send_nnz, group=self.group)
        self.ctx.assemble_row_ptr(self.n, self.N, self.bitmaps[self.rank], recv_len.data_ptr(), int(recv_len.numel()),
                                  row_ptr.data_ptr(), noalloc=self.n <= (1 << 22), want_nnz=False)
        return row_ptr, recv_col, recv_val

    def run(self):
        if self.dB is not None:
            return self._run_device()
        torch, dist = self.torch, self.dist
        send_len = self._pack(self.row_len, 0)
        send_col = self._pack(self.col, 1)
        send_val = self._pack(self.val, 1)
        cdev = "cpu" if self.staged else self.device
        if self.staged:
            send_len, send_col, send_val = send_len.cpu(), send_col.cpu(), send_val.cpu()
        recv_len = torch.empty(sum(self.recv_rows), dtype=torch.int32, device=cdev)
        recv_col = torch.empty(sum(self.recv_nnz), dtype=torch.int32, device=cdev)
        recv_val = torch.empty(sum(self.recv_nnz), dtype=torch.float64, device=cdev)
        dist.all_to_all_single(recv_len, send_len, self.recv_rows, self.send_rows, group=self.group)
        dist.all_to_all_single(recv_col, send_col, self.recv_nnz, self.send_nnz, group=self.group)
        dist.all_to_all_single(recv_val, send_val, self.recv_nnz, self.send_nnz, group=self.group)
        if self.staged:
            recv_len, recv_col, recv_val = (x.to(self.device) for x in (recv_len, recv_col, recv_val))
        # full-height row pointer: absent rows are empty
        full_len = torch.zeros(self.n + 1, dtype=torch.int64, device=self.device)
        full_len[self.present_idx + 1] = recv_len.to(torch.int64)
        row_ptr = torch.cumsum(full_len, dim=0)
        return row_ptr, recv_col, recv_val


def gather_bitmaps(my_bits: np.ndarray, device, group=None) -> np.ndarray:
    """All-gather of the per-rank section bitmaps (reference src/main.cpp:238-240)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    cdev = "cpu" if _staged(device, group) else device
    mine = torch.from_numpy(np.asarray(my_bits, dtype=np.uint8)).to(cdev)
    out = torch.empty(world * len(my_bits), dtype=torch.uint8, device=cdev)
    dist.all_gather_into_tensor(out, mine, group=group)
    return out.cpu().numpy().reshape(world, len(my_bits)).astype(bool)


# ---- the distributed operator (GPU) ------------------------------------------------------------------------

class DistSpgemm:
    """Per-rank operator: mirrors the surface of the reference's mults classes
    (include/mults.hpp:17-33): construct once, `gemm()` per run (replaces reset() + gemm()),
    `result()` for the rank's rows of C."""

    def __init__(self, A: HostCsr, B: Optional[HostCsr], impl: str = "drop", n_sections: int = 512, seed: int = 1,
                 partition: str = "naive", device_index: Optional[int] = None, group=None,
                 shuffling: Optional[str] = None, shuffle_iterations: int = 20, stages: int = 1,
                 relabel: Optional[str] = None):
        """stages > 1: the pipelined schedule.  The rank's rows are cut into `stages` blocks, each with
        its own section bitmap and exchange; the exchange of block s + 1 runs on a second stream
        while block s is multiplied (the overlap the reference's ring gets from Isend/Irecv around
        mult_step, src/mults/drop_parallel.cpp:79-114).  Every row of C is still computed by one
        full-width multiply, so the result is the same, bit for bit."""
        import torch
        import torch.distributed as dist
        from . import capi
        if impl not in ("baseline", "drop", "shuffle"):
            raise ValueError("impl must be baseline, drop or shuffle")
        self.torch, self.dist, self.group = torch, dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.impl, self.N = impl, n_sections
        B = A if B is None else B
        if device_index is None:
            device_index = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(device_index)
        self.device = torch.device("cuda", device_index)
        self.ctx = capi.Context(device_index)
        # One stream of this operator's own carries the exchange (the collectives are ordered after
        # it, their results before what follows on it), the tensor work that turns the received
        # sections into the local B, and the multiply's kernels.
        self.stream = torch.cuda.Stream(device=self.device)
        self.ctx.set_stream(self.stream.cuda_stream)
        # shuffling: None = what the impl implies (`shuffle` -> random), or the reference's
        # shuffling argument none | random | iterative (src/main.cpp:143-152)
        if shuffling not in (None, "none", "random", "iterative"):
            raise ValueError("shuffling must be none, random or iterative")
        if shuffling == "iterative":
            # rank 0 computes the permutations, everybody receives them (src/main.cpp:202-205)
            if self.rank == 0:
                a_sh, b_sh = iterative_shuffles(self.ctx, A, B, shuffle_iterations)
                both = torch.from_numpy(np.concatenate([a_sh, b_sh]))
            else:
                both = torch.empty(A.rows + B.cols, dtype=torch.int64)
            if self.world > 1:
                both = _broadcast0(both, self.device, group)
            both = both.numpy()
            self.A_shuffle, self.B_shuffle = both[:A.rows].copy(), both[A.rows:].copy()
        else:
            kind = impl if shuffling is None else ("shuffle" if shuffling == "random" else "none")
            self.A_shuffle, self.B_shuffle = make_shuffles(A.rows, B.cols, kind, seed)
        if partition == "naive":
            self.partitions = capi.partition_naive(A.rows, B.cols, self.world)
        elif partition == "balanced":
            # reference src/parts/baseline.cpp:44-86 cuts the (shuffled) rows by their share of
            # nnz(C_sparsity); here the weight of a row is its number of products, which needs no
            # pattern of the answer
            self.partitions = capi.partition_balanced(row_products(A, B)[self.A_shuffle], B.cols, self.world)
        else:
            raise ValueError("partition must be naive or balanced")
        # A column shuffle is applied to the result of the multiply, not to B (a permutation of the
        # columns commutes with the product, bit for bit): B keeps the column locality the window
        # kernels live on.  new_id[c] = position of original column c.
        # That pays when B has column locality to lose (rows of the product a few ten thousand
        # columns wide: the window path); for a B without it (Erdos-Renyi, R-MAT) sorting the result
        # costs more than it saves, and B is relabelled up front.  The rule reads the whole
        # operands, so every rank decides the same (the exchanged sections carry the column ids).
        self.col_new_id = None
        shuffled_cols = not np.array_equal(self.B_shuffle, np.arange(B.cols))
        if relabel not in (None, "result", "b"):
            raise ValueError("relabel must be None (decide from the operands), 'result' or 'b'")
        self.late_relabel = shuffled_cols and (product_rows_are_narrow(A, B) if relabel is None else relabel == "result")
        if self.late_relabel:
            inv = np.empty(B.cols, dtype=np.int32)
            inv[self.B_shuffle] = np.arange(B.cols, dtype=np.int32)
            self.col_new_id = torch.from_numpy(inv).to(self.device)
        self.ops = build_rank_operands(A, B, self.rank, self.world, self.partitions, self.A_shuffle, self.B_shuffle,
                                       n_sections, relabel_b=not self.late_relabel, ctx=self.ctx)
        ap = self.ops.A_part
        if stages < 1:
            raise ValueError("stages must be at least 1")
        self.n_stages = int(stages)
        self.comm_stream = torch.cuda.Stream(device=self.device) if self.n_stages > 1 else None
        # the exchange's device helpers run on a context of their own when the exchange has its own
        # stream (pipelined schedule); else on the multiply's context and stream
        self.xctx = self.ctx
        if self.comm_stream is not None:
            self.xctx = capi.Context(device_index)
            self.xctx.set_stream(self.comm_stream.cuda_stream)
        # every rank cuts its rows the same way (equal shares), so that all ranks issue the same
        # sequence of collectives
        cuts = [(ap.rows * s) // self.n_stages for s in range(self.n_stages + 1)]
        self.stage_rows = list(zip(cuts[:-1], cuts[1:]))
        self.dA_stage, self.bitmap_stage, self.exchange_stage = [], [], []
        for r0, r1 in self.stage_rows:
            sub = ap if self.n_stages == 1 else select_rows(ap, np.arange(r0, r1, dtype=np.int64))
            dA = self.ctx.from_host(sub.rows, sub.cols, sub.row_ptr, sub.col_idx, sub.val)
            if impl == "baseline":
                bits = np.ones(n_sections, dtype=bool)
            else:
                bits = self.ctx.section_bitmap(dA, n_sections)   # device-built, reference rule
            bitmaps = gather_bitmaps(bits, self.device, group)
            share = self.exchange_stage[0] if self.exchange_stage else None
            self.dA_stage.append(dA)
            self.bitmap_stage.append(bits)
            self.exchange_stage.append(SectionExchange(self.ops, n_sections, bitmaps, self.device, group, share=share,
                                                       ctx=self.xctx))
        self.dA = self.dA_stage[0]
        self.bitmap = np.logical_or.reduce(self.bitmap_stage)      # sections this rank needs at all
        self.bitmaps = self.exchange_stage[0].bitmaps
        self.exchange = self.exchange_stage[0]
        self.bytes_sent_per_run = sum(e.bytes_sent_per_run for e in self.exchange_stage)
        self.bytes_sent_without_drop = sum(e.bytes_sent_without_drop for e in self.exchange_stage)
        self.C_stage = []
        self.C = None

    def _multiply_stage(self, s, received):
        row_ptr, col, val = received
        dB = self.ctx.wrap_device(self.ops.n, self.ops.n_cols, int(col.numel()), row_ptr.data_ptr(), col.data_ptr(),
                                  val.data_ptr(), validate=False, keepalive=(row_ptr, col, val))
        C = self.ctx.multiply(self.dA_stage[s], dB)   # returns after the stream has drained
        st = self.ctx.stats()
        dB.free()
        if self.col_new_id is not None:
            shuffled = self.ctx.relabel_columns(C, self.col_new_id.data_ptr(), self.ops.n_cols)
            C.free()
            C = shuffled
        return C, st

    def gemm(self):
        """One timed run: exchange of the needed B sections + the local multiply (per stage)."""
        torch = self.torch
        for C in self.C_stage:
            C.free()
        self.C_stage, self.C = [], None
        stats = []
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        if self.n_stages == 1:
            with torch.cuda.stream(self.stream):
                C, st = self._multiply_stage(0, self.exchange_stage[0].run())
            self.C_stage, stats = [C], [st]
        else:
            self.comm_stream.wait_stream(self.stream)
            with torch.cuda.stream(self.comm_stream):
                nxt = self.exchange_stage[0].run()
                ready = torch.cuda.Event()
                ready.record(self.comm_stream)
            for s in range(self.n_stages):
                cur, cur_ready = nxt, ready
                if s + 1 < self.n_stages:  # queued before the multiply below blocks the host
                    with torch.cuda.stream(self.comm_stream):
                        nxt = self.exchange_stage[s + 1].run()
                        ready = torch.cuda.Event()
                        ready.record(self.comm_stream)
                self.stream.wait_event(cur_ready)
                with torch.cuda.stream(self.stream):
                    C, st = self._multiply_stage(s, cur)
                self.C_stage.append(C)
                stats.append(st)
                del cur
            self.comm_stream.synchronize()
        self.C = self.C_stage[0]
        self.last_stats = dict(stats[0])
        for st in stats[1:]:
            for k, v in st.items():
                if k in ("nnz_b", "temp_bytes"):  # not additive over the stages
                    self.last_stats[k] = max(self.last_stats[k], v)
                elif isinstance(v, (int, float)):
                    self.last_stats[k] = self.last_stats[k] + v
        return self.C

    def result(self) -> HostCsr:
        parts = [C.to_host() for C in self.C_stage]
        if len(parts) == 1:
            rp, ci, v = parts[0]
        else:
            offs = np.cumsum([0] + [int(p[0][-1]) for p in parts]).astype(np.uint64)
            rp = np.concatenate([parts[0][0]] + [p[0][1:] + offs[i + 1] for i, p in enumerate(parts[1:])])
            ci = np.concatenate([p[1] for p in parts])
            v = np.concatenate([p[2] for p in parts])
        return HostCsr(self.ops.A_part.rows, self.ops.n_cols, rp, ci, v)

    def close(self):
        for C in self.C_stage:
            C.free()
        for dA in self.dA_stage:
            dA.free()
        if self.exchange_stage and self.exchange_stage[0].dB is not None:
            self.exchange_stage[0].dB.free()
        if self.xctx is not self.ctx:
            self.xctx.close()
        self.ctx.close()


def merge_unshuffle(parts: Sequence[HostCsr], partitions: Sequence[Partition], A_shuffle: np.ndarray,
                    B_shuffle: np.ndarray, n_rows: int, n_cols: int) -> HostCsr:
    """Merge the per-rank results into the unshuffled C (what the reference's
    scripts/postprocessing.py:44-79 does with C_<rank>.mtx, A_shuffle, B_shuffle, partitions.csv)."""
    rs, cs, vs = [], [], []
    for C, p in zip(parts, partitions):
        rs.append(A_shuffle[C.row_ids() + p[0]])
        cs.append(B_shuffle[C.col_idx.astype(np.int64)])
        vs.append(C.val)
    return HostCsr.from_coo(n_rows, n_cols, np.concatenate(rs), np.concatenate(cs), np.concatenate(vs)).sorted_rows()


# ---- bench entry for N > 1 -----------------------------------------------------------------------------------

def bench_main(args):
    import torch
    import torch.distributed as dist
    import bench as bench_mod
    if "RANK" not in os.environ:  # started without a launcher: a world of one (bench.py --dist)
        import socket
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    # SPGEMM_DIST_BACKEND=gloo with SPGEMM_DIST_DEVICE=0 rehearses several ranks on ONE GPU (collectives
    # staged through host memory); the measured configuration is always nccl, one GPU per rank.
    backend = os.environ.get("SPGEMM_DIST_BACKEND", "nccl")
    local = int(os.environ.get("SPGEMM_DIST_DEVICE", local))
    torch.cuda.set_device(local)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(backend)
    A, name = bench_mod.make_workload(args)
    op = DistSpgemm(A, None, impl=args.impl, n_sections=512, seed=1, device_index=local,
                    partition=getattr(args, "partition", "naive"), stages=getattr(args, "stages", 1))
    for o in getattr(args, "opt", []):
        opt_name, opt_value = o.split("=")
        op.ctx.set_option(opt_name, int(opt_value))
    for _ in range(args.warmup):
        op.gemm()
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        op.gemm()
    op.ctx.synchronize()
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    st = op.last_stats
    # The timed steps above carry no instrumentation (HIP events around each of the ~50 launches of a
    # step and the host round trip that reads them are a tenth of a 1-ms step).  The roofline figure of
    # rank 0's dominant numeric launch comes from the same number of steps run again with events on
    # the operator's stream, then one more step that also sums the rows of every launch (bytes).
    op.ctx.set_option("profile_launches", 1)
    per_launch_ms = None
    for _ in range(args.steps):
        op.gemm()
        ms = np.array([l["ms"] for l in op.ctx.launches()])
        per_launch_ms = ms if per_launch_ms is None or len(ms) != len(per_launch_ms) else per_launch_ms + ms
    op.ctx.set_option("profile_launches", 2)
    op.gemm()
    roofline = None
    launches = op.ctx.launches()
    if rank == 0 and per_launch_ms is not None and len(launches) == len(per_launch_ms):
        for l, ms in zip(launches, per_launch_ms / args.steps):
            l["ms"] = float(ms)
        num = [l for l in launches if l["phase"] == "numeric"]
        if num:
            dom = max(num, key=lambda l: l["ms"])
            b = bench_mod.algorithmic_bytes(dom["nnz_a"], dom["nprod"], dom["nnz_c"], dom["rows"])
            bc = bench_mod.compulsory_bytes(dom["nnz_a"], dom["nnz_c"], dom["rows"])
            achieved = bc / (dom["ms"] * 1e-3) / 1e9
            kname = bench_mod.kernel_name(dom)
            # the same two figures as the single-GPU line: `frac` from the bytes no kernel can avoid
            # (a bound, <= 1), the traffic model without reuse of B rows beside it
            roofline = {"bound": "hbm", "achieved": achieved, "peak": bench_mod.HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": achieved / bench_mod.HBM_PEAK_GBS, "traffic": None, "kernel": kname,
                        "kernel_ms": dom["ms"], "kernel_rows": dom["rows"], "kernel_compulsory_bytes": bc,
                        "kernel_algorithmic_bytes": b,
                        "frac_model_no_reuse": b / (dom["ms"] * 1e-3) / 1e9 / bench_mod.HBM_PEAK_GBS,
                        "scope": "rank 0's launch, one GPU's peak; HIP events over the same number of steps, repeated after the timed region"}
    t = torch.tensor([t1 - t0], dtype=torch.float64, device="cuda")
    t = _all_reduce(t, op=dist.ReduceOp.MAX)
    sums = torch.tensor([st["nprod"], st["nnz_c"], st["nnz_a"], st["nnz_b"], op.bytes_sent_per_run,
                         st["bytes_algorithmic"], op.bytes_sent_without_drop], dtype=torch.float64, device="cuda")
    sums = _all_reduce(sums)
    drop = torch.tensor([1.0 - float(np.mean(op.bitmap))], dtype=torch.float64, device="cuda")
    drop = _all_reduce(drop)
    mult_ms = torch.tensor([st["ms_total"]], dtype=torch.float64, device="cuda")
    mult_ms = _all_reduce(mult_ms, op=dist.ReduceOp.MAX)
    if rank == 0:
        ms_per_step = float(t.item()) * 1e3 / args.steps
        nprod, nnz_c, nnz_a, nnz_b_local, sent, b_alg, sent_all = (float(x) for x in sums.tolist())
        out = {
            "metric": "SpGEMM GFLOP/s (useful flops, 2*nprod/t) C=A*A fp64",
            "value": 2.0 * nprod / (ms_per_step * 1e-3) / 1e9, "unit": "GFLOP/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic" if args.workload != "mtx" else "file",
            "config": {"workload": name, "rows": A.rows, "nnz_a": A.nnz, "nprod": int(nprod), "nnz_c": int(nnz_c),
                       "impl": args.impl, "n_sections": 512, "partition": getattr(args, "partition", "naive"),
                       "stages": getattr(args, "stages", 1),
                       "parallelism": f"{world} GPUs, A by row blocks, B row sections by all-to-all (RCCL)"},
            "exchange": {"bytes_sent_all_ranks_per_step": int(sent), "bytes_if_every_section_were_sent": int(sent_all), "b_entries_resident_all_ranks": int(nnz_b_local),
                         "mean_section_drop_fraction": float(drop.item()) / world},
            "whole_multiply": {"algorithmic_bytes_all_ranks": int(b_alg),
                               "model_no_reuse_GBs": b_alg / (ms_per_step * 1e-3) / 1e9,
                               "frac_model_no_reuse_all_gpus": b_alg / (ms_per_step * 1e-3) / 1e9 / (8000.0 * world),
                               "compulsory_frac_of_hbm_peak_all_gpus": bench_mod.compulsory_bytes(nnz_a, nnz_c, A.rows) / (ms_per_step * 1e-3) / 1e9 / (8000.0 * world),
                               "slowest_rank_multiply_ms": float(mult_ms.item())},
            "roofline": roofline, "cpu_baseline": None,
        }
        print(json.dumps(out))
    op.close()
    dist.destroy_process_group()
