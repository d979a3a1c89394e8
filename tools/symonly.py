import sys, os, json
sys.path.insert(0, os.getcwd())
import torch
from spgemm_amd import capi, hostcsr
w = sys.argv[1]
A = hostcsr.shell_like() if w == "shell" else hostcsr.diel_like()
ctx = capi.Context(0)
ctx.set_option("profile_launches", 1)
dA = ctx.from_host(A.rows, A.cols, A.row_ptr, A.col_idx, A.val)
rp = torch.empty(A.rows + 1, dtype=torch.int64, device="cuda")
import time
ctx.set_option("profile_launches", 0)
for i in range(3):
    ctx.symbolic(dA, dA, rp.data_ptr())
ctx.synchronize()
t0 = time.perf_counter()
for i in range(10):
    ctx.symbolic(dA, dA, rp.data_ptr())
ctx.synchronize()
print(w, os.environ.get("SPGEMM_AMD_LIB", "base").split("/")[-1], "analysis + symbolic + scan: %.3f ms" % ((time.perf_counter() - t0) / 10 * 1e3))
