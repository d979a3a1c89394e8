"""Per-rank cost of one distributed step when the matrix is split P ways, emulated at world size 1:
the operator gets the first 1/P of the rows of A and the whole B (it then moves only the sections
its bitmap marks)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
import numpy as np
from spgemm_amd import hostcsr, dist as sd

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("LOCAL_RANK", "0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
A = hostcsr.shell_like()
for P in (1, 2, 4, 8):
    rows = A.rows // P
    Ap = sd.select_rows(A, np.arange(rows, dtype=np.int64))
    op = sd.DistSpgemm(Ap, A, impl="drop", n_sections=512, seed=1, device_index=0)
    for _ in range(3):
        op.gemm()
    torch.cuda.synchronize()
    n = 20
    t0 = time.perf_counter()
    for _ in range(n):
        op.gemm()
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / n * 1e3
    with torch.cuda.stream(op.stream):
        t0 = time.perf_counter()
        for _ in range(n):
            r = op.exchange.run()
        torch.cuda.synchronize()
        t_ex = (time.perf_counter() - t0) / n * 1e3
    st = op.last_stats
    print(f"1/{P} of the rows: step {t_all:.3f} ms  exchange {t_ex:.3f} ms  multiply {st['ms_total']:.3f} ms "
          f"(analyze {st['ms_analyze']:.3f} symbolic {st['ms_symbolic']:.3f} scan {st['ms_scan']:.3f} numeric {st['ms_numeric']:.3f})  sections {int(op.bitmap.sum())}", flush=True)
    op.close()
dist.destroy_process_group()
