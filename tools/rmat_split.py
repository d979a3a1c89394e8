"""Times the multiply A' * A on R-MAT where A' keeps only the rows of A with len in (lo, hi]."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from spgemm_amd import capi, hostcsr, dist as sd

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 22
A = hostcsr.rmat(scale, 4, seed=1)
lens = np.diff(A.row_ptr.astype(np.int64))
ctx = capi.Context(0)
ctx.set_option("profile_launches", 1)
for o in sys.argv[2:]:
    k, v = o.split("=")
    ctx.set_option(k, int(v))
dB = ctx.from_host(A.rows, A.cols, A.row_ptr, A.col_idx, A.val)
for lo, hi in ((0, 64), (64, 1024), (1024, 1 << 30), (0, 1 << 30)):
    keep = np.flatnonzero((lens > lo) & (lens <= hi))
    Ap = sd.select_rows(A, keep)
    dA = ctx.from_host(Ap.rows, Ap.cols, Ap.row_ptr, Ap.col_idx, Ap.val)
    for rep in range(2):
        C = ctx.multiply(dA, dB)
        st = ctx.stats()
        L = ctx.launches()
        C.free()
    print(f"lenA in ({lo},{hi}]: rows {len(keep)} nprod {st['nprod']:.3e} nnzC {st['nnz_c']:.3e} total {st['ms_total']:.1f} ms", flush=True)
    for l in L:
        if l["ms"] > 1.0:
            print("   ", {k: l[k] for k in ("phase", "path", "rows", "ms")}, flush=True)
    dA.free()
