"""Host-timed pieces of one distributed step at world size 1 for the first 1/P of the shell rows."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
import numpy as np
from spgemm_amd import hostcsr, dist as sd

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29535")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("LOCAL_RANK", "0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
A = hostcsr.shell_like()
P = int(sys.argv[1]) if len(sys.argv) > 1 else 8
Ap = sd.select_rows(A, np.arange(A.rows // P, dtype=np.int64))
op = sd.DistSpgemm(Ap, A, impl="drop", n_sections=512, seed=1, device_index=0)
ex = op.exchange
for _ in range(3):
    op.gemm()
def timed(f, n=50):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        r = f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, r
with torch.cuda.stream(op.stream):
    t_pack, packed = timed(lambda: (ex._pack(ex.row_len, 0), ex._pack(ex.col, 1), ex._pack(ex.val, 1)))
    send_len, send_col, send_val = packed
    def a2a():
        rl = torch.empty(sum(ex.recv_rows), dtype=torch.int32, device=ex.device)
        rc = torch.empty(sum(ex.recv_nnz), dtype=torch.int32, device=ex.device)
        rv = torch.empty(sum(ex.recv_nnz), dtype=torch.float64, device=ex.device)
        dist.all_to_all_single(rl, send_len, ex.recv_rows, ex.send_rows)
        dist.all_to_all_single(rc, send_col, ex.recv_nnz, ex.send_nnz)
        dist.all_to_all_single(rv, send_val, ex.recv_nnz, ex.send_nnz)
        return rl, rc, rv
    t_a2a, (rl, rc, rv) = timed(a2a)
    def assemble():
        full_len = torch.zeros(ex.n + 1, dtype=torch.int64, device=ex.device)
        full_len[ex.present_idx + 1] = rl.to(torch.int64)
        return torch.cumsum(full_len, dim=0)
    t_asm, row_ptr = timed(assemble)
    def wrap_mul():
        dB = op.ctx.wrap_device(op.ops.n, op.ops.n_cols, int(rc.numel()), row_ptr.data_ptr(), rc.data_ptr(), rv.data_ptr(),
                                validate=False, keepalive=(row_ptr, rc, rv))
        C = op.ctx.multiply(op.dA, dB)
        st = op.ctx.stats()
        dB.free(); C.free()
        return st
    t_mul, st = timed(wrap_mul)
t_all, _ = timed(op.gemm)
print(f"1/{P}: step {t_all:.3f} | pack {t_pack:.3f} all_to_all x3 {t_a2a:.3f} assemble {t_asm:.3f} wrap+multiply+free {t_mul:.3f} "
      f"(device: analyze {st['ms_analyze']:.3f} symbolic {st['ms_symbolic']:.3f} scan {st['ms_scan']:.3f} numeric {st['ms_numeric']:.3f} total {st['ms_total']:.3f})")
op.close()
dist.destroy_process_group()
