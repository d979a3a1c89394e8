"""Worker of the multi-process CPU test (gloo): runs the section exchange of spgemm_amd.dist on CPU
tensors and checks, with the ORACLE as the checker, that every rank ends up with exactly the B it
needs and that its rows of C equal what the reference's ranks compute."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import oracle as orc  # noqa: E402
from spgemm_amd import capi, hostcsr  # noqa: E402
from spgemm_amd import dist as sd  # noqa: E402


def to_o(h):
    return orc.OCsr(h.rows, h.cols, h.row_ptr, h.col_idx, h.val)


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    golden = os.path.join(ROOT, "tests", "golden")
    cases = [("first", hostcsr.read_mtx(f"{golden}/ref_first/A.mtx"), 4),
             ("er", hostcsr.erdos_renyi(700, 5, seed=3), 16),
             ("band", hostcsr.banded_random(900, 6, 30, seed=4), 64)]
    for name, A, N in cases:
        E = orc.spgemm(to_o(A), to_o(A))
        sent = {}
        for impl in ("baseline", "drop", "shuffle", "drop+balanced"):
            balanced = impl.endswith("+balanced")
            impl = impl.split("+")[0]
            a_sh, b_sh = sd.make_shuffles(A.rows, A.cols, impl, seed=7)
            if balanced:  # rows cut by their share of the products (needs no pattern of C)
                w = sd.row_products(A, A)
                parts = capi.partition_balanced(w[a_sh], A.cols, world)
                assert parts == orc.partition_balanced(w[a_sh], A.cols, world)
                assert parts[0][0] == 0 and parts[-1][1] == A.rows
                assert all(parts[r][1] == parts[r + 1][0] for r in range(world - 1))
            else:
                parts = capi.partition_naive(A.rows, A.cols, world)
            ops = sd.build_rank_operands(A, A, rank, world, parts, a_sh, b_sh, N)
            bits = np.ones(N, dtype=bool) if impl == "baseline" else orc.section_bitmap(to_o(ops.A_part), N)
            bitmaps = sd.gather_bitmaps(bits, torch.device("cpu"))
            ex = sd.SectionExchange(ops, N, bitmaps, torch.device("cpu"))
            row_ptr, col, val = ex.run()
            Bl = orc.OCsr(A.rows, A.cols, row_ptr.numpy().astype(np.uint64), col.numpy().view(np.uint32), val.numpy())
            # 1. the local B holds exactly the marked sections of the (column-relabelled) B
            inv = np.empty(A.cols, dtype=np.int64)
            inv[b_sh] = np.arange(A.cols)
            Bfull = sd.relabel_cols(A, inv) if impl == "shuffle" else A
            sb = sd.section_bounds(A.rows, N)
            want_len = np.diff(Bfull.row_ptr.astype(np.int64))
            mask = np.zeros(A.rows, dtype=bool)
            for s in np.flatnonzero(bits):
                mask[sb[s]:sb[s + 1]] = True
            want_len = np.where(mask, want_len, 0)
            assert np.array_equal(np.diff(Bl.row_ptr.astype(np.int64)), want_len), (name, impl, rank)
            keep = np.repeat(mask, np.diff(Bfull.row_ptr.astype(np.int64)))
            assert np.array_equal(Bl.col_idx, Bfull.col_idx[keep]) and np.array_equal(Bl.val, Bfull.val[keep])
            # 2. my rows of C = what the reference's rank computes (oracle restatement of its flow)
            C_mine = orc.spgemm(to_o(ops.A_part), Bl)
            ref = orc.distributed_reference(to_o(A), to_o(A), world, parts, a_sh, b_sh, n_sections=N)
            orc.assert_same(C_mine, ref[rank].C, rtol=0.0)
            if impl != "baseline":
                assert np.array_equal(bits, ref[rank].bitmap)
            # 3. merged and un-shuffled, the ranks' results are C = A*A
            gathered = [None] * world
            dist.all_gather_object(gathered, (C_mine.row_ptr, C_mine.col_idx, C_mine.val))
            if rank == 0:
                Cs = [hostcsr.HostCsr(parts[r][1] - parts[r][0], A.cols, *gathered[r]) for r in range(world)]
                M = sd.merge_unshuffle(Cs, parts, a_sh, b_sh, A.rows, A.cols)
                orc.assert_same(to_o(M), E, rtol=0.0)
            tot = torch.tensor([ex.bytes_sent_per_run], dtype=torch.int64)
            dist.all_reduce(tot)
            sent[impl + ("+balanced" if balanced else "")] = int(tot.item())
        assert sent["drop"] <= sent["baseline"], sent
        if name == "band":
            assert sent["drop"] < sent["baseline"] // 2, sent  # banded: most sections are dropped
        if rank == 0:
            print(f"DIST_OK {name} world={world} bytes={sent}", flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
