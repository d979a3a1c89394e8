"""GPU box: one context, a large workload and then a smaller one, each with two copies of its matrix: wall
time per multiply against the device time of its phases (a pool that evicts the wrong blocks shows as
milliseconds of hipMalloc / hipFree per multiply of the second workload)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (its HIP runtime first)
from spgemm_amd import capi, hostcsr

ctx = capi.Context(0)
for name, M in (("rmat20", hostcsr.rmat(20, 4, seed=1)), ("diel", hostcsr.diel_like()), ("er21", hostcsr.erdos_renyi(1 << 21, 16, seed=42))):
    dA = ctx.from_host(M.rows, M.cols, M.row_ptr, M.col_idx, M.val)
    dB = ctx.from_host(M.rows, M.cols, M.row_ptr, M.col_idx, M.val)
    ctx.multiply(dA, dB).free()
    ctx.synchronize()
    t0 = time.perf_counter()
    dev = 0.0
    for _ in range(5):
        ctx.multiply(dA, dB).free()
        dev += ctx.stats()["ms_total"]
    ctx.synchronize()
    wall = (time.perf_counter() - t0) * 1e3 / 5
    print(f"{name}: wall {wall:.2f} ms per multiply, device {dev / 5:.2f} ms", flush=True)
    dB.free(); dA.free()
ctx.close()
