"""Worker of the multi-rank GPU rehearsal: every rank runs the whole per-rank operator
(spgemm_amd.dist.DistSpgemm: device-built bitmap, exchange of B row sections, HIP multiply) on the
SAME GPU, with a gloo process group whose collectives are staged through host memory.  Rank 0
merges the ranks' rows of C and checks them against the oracle."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import oracle as orc  # noqa: E402
from spgemm_amd import hostcsr  # noqa: E402
from spgemm_amd import dist as sd  # noqa: E402


def to_o(h):
    return orc.OCsr(h.rows, h.cols, h.row_ptr, h.col_idx, h.val)


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    golden = os.path.join(ROOT, "tests", "golden")
    cases = [("first", hostcsr.read_mtx(f"{golden}/ref_first/A.mtx"), 4),
             ("band", hostcsr.banded_random(9000, 7, 40, seed=4), 512),
             ("shell", hostcsr.shell_like(n_nodes=1200, grid_w=40), 512),
             ("rmat", hostcsr.rmat(11, 5, seed=6), 128)]
    for name, A, N in cases:
        E = orc.spgemm(to_o(A), to_o(A))
        sent = {}
        for impl in ("baseline", "drop", "shuffle", "drop+balanced", "drop+stages", "shuffle+stages"):
            partition = "balanced" if impl.endswith("+balanced") else "naive"
            stages = 3 if impl.endswith("+stages") else 1   # pipelined schedule: 3 blocks of rows per rank
            variant = impl
            impl = impl.split("+")[0]
            op = sd.DistSpgemm(A, None, impl=impl, n_sections=N, seed=7, device_index=0, partition=partition,
                               stages=stages)
            op.gemm()
            op.gemm()
            C = op.result()
            ref = orc.distributed_reference(to_o(A), to_o(A), world, op.partitions, op.A_shuffle, op.B_shuffle,
                                            n_sections=N)
            orc.assert_same(to_o(C), ref[rank].C, rtol=0.0)
            if impl != "baseline":  # with stages: the union of the blocks' bitmaps
                assert np.array_equal(op.bitmap, ref[rank].bitmap), (name, impl, rank)
            gathered = [None] * world
            dist.all_gather_object(gathered, (C.row_ptr, C.col_idx, C.val))
            if rank == 0:
                Cs = [hostcsr.HostCsr(op.partitions[r][1] - op.partitions[r][0], A.cols, *gathered[r])
                      for r in range(world)]
                M = sd.merge_unshuffle(Cs, op.partitions, op.A_shuffle, op.B_shuffle, A.rows, A.cols)
                orc.assert_same(to_o(M), E, rtol=0.0)
            tot = torch.tensor([op.bytes_sent_per_run], dtype=torch.int64)
            dist.all_reduce(tot)
            sent[variant] = int(tot.item())
            op.close()
        assert sent["drop"] <= sent["baseline"], sent
        if rank == 0:
            print(f"DIST_GPU_OK {name} world={world} bytes={sent}", flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
