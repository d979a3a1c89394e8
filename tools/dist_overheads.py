"""Times the pieces of DistSpgemm.gemm() at world size 1 over RCCL (what is fixed cost per step)."""
import os, sys, time
sys.path.insert(0, ".")
import torch, torch.distributed as dist
import numpy as np
from spgemm_amd import hostcsr, dist as sd

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("LOCAL_RANK", "0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
A = hostcsr.shell_like()
for impl in ("drop", "baseline"):
    op = sd.DistSpgemm(A, None, impl=impl, n_sections=512, seed=1, device_index=0)
    for _ in range(3):
        op.gemm()
    torch.cuda.synchronize()
    n = 10
    t0 = time.perf_counter()
    for _ in range(n):
        op.gemm()
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / n * 1e3
    # exchange only
    with torch.cuda.stream(op.stream):
        t0 = time.perf_counter()
        for _ in range(n):
            r = op.exchange.run()
        torch.cuda.synchronize()
        t_ex = (time.perf_counter() - t0) / n * 1e3
        # pack only
        t0 = time.perf_counter()
        for _ in range(n):
            a = op.exchange._pack(op.exchange.row_len, 0); b = op.exchange._pack(op.exchange.col, 1); c = op.exchange._pack(op.exchange.val, 1)
        torch.cuda.synchronize()
        t_pack = (time.perf_counter() - t0) / n * 1e3
    print(f"{impl}: gemm {t_all:.2f} ms  exchange {t_ex:.2f} ms  (pack {t_pack:.2f})  multiply {op.last_stats['ms_total']:.2f} ms", flush=True)
    op.close()
dist.destroy_process_group()
