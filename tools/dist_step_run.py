"""A few distributed steps at world size 1 on the first 1/P of the shell rows (to be traced)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
import numpy as np
from spgemm_amd import hostcsr, dist as sd

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29536")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("LOCAL_RANK", "0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
A = hostcsr.shell_like()
P = int(sys.argv[1]) if len(sys.argv) > 1 else 8
Ap = sd.select_rows(A, np.arange(A.rows // P, dtype=np.int64))
op = sd.DistSpgemm(Ap, A, impl="drop", n_sections=512, seed=1, device_index=0)
for _ in range(10):
    op.gemm()
op.ctx.synchronize()
torch.cuda.synchronize()
op.close()
dist.destroy_process_group()
