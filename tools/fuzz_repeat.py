"""GPU box, repo root: python tools/fuzz_repeat.py <seed> <reps> - one case of tests/test_gpu_fuzz.py repeated in fresh
contexts (two multiplies each) under a few option overrides; prints how often row pointer, entry count and the
oracle's count disagree.  For failures that come and go (a first multiply writes into fresh memory)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib.util
import numpy as np
from oracle import oracle as orc
from spgemm_amd import capi
spec = importlib.util.spec_from_file_location('fz', 'tests/test_gpu_fuzz.py'); fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
seed = int(sys.argv[1]); reps = int(sys.argv[2]); dirty = len(sys.argv) > 3
A, B, opts = fz.make_case(seed)
E = orc.spgemm(fz.to_o(A), fz.to_o(B))
Ec = E.canonical()
print({k: v for k, v in opts.items() if fz.DEFAULTS.get(k) != v})
la = np.diff(A.row_ptr.astype(np.int64)); print('A', A.rows, A.cols, A.nnz, 'max row', la.max(), 'B', B.rows, B.cols, B.nnz)
for over in ({}, {"slice_sorted": 0}, {"big_sym_sorted": 0}, {"slice_sorted": 0, "big_sym_sorted": 0}):
    bad = []
    for rep in range(reps):
        if dirty:  # freed device memory keeps its contents: what the next hipMalloc hands out is then not zero
            import torch
            x = torch.randint(-2**31, 2**31 - 1, (1 << 28,), dtype=torch.int32, device="cuda:0")
            y = torch.full((1 << 28,), -1, dtype=torch.int32, device="cuda:0")
            torch.cuda.synchronize(); del x, y; torch.cuda.empty_cache()
        o = dict(opts); o.update(over)
        ctx = capi.Context(0)
        for k, v in o.items():
            ctx.set_option(k, v)
        dA = ctx.from_host(A.rows, A.cols, A.row_ptr, A.col_idx, A.val)
        dB = ctx.from_host(B.rows, B.cols, B.row_ptr, B.col_idx, B.val)
        for it in range(2):
            dC = ctx.multiply(dA, dB)
            rp, ci, v = dC.to_host()
            st = ctx.stats()
            if not (int(rp[-1]) == len(ci) == E.nnz):
                bad.append((rep, it, int(rp[-1]), len(ci)))
            else:
                C = orc.OCsr(A.rows, B.cols, rp, ci, v)
                same_c = np.array_equal(C.col_idx, Ec.col_idx)
                dv = np.flatnonzero(~((C.val == Ec.val) | (np.isnan(C.val) & np.isnan(Ec.val))))
                if not same_c or len(dv):
                    rows_of = np.searchsorted(rp, dv, side="right") - 1
                    bad.append((rep, it, "cols" if not same_c else "vals", len(dv), int(np.isnan(C.val).sum()), sorted(set(rows_of.tolist()))[:5],
                                [int(x) for x in dv[:4] - rp[rows_of[:4]]], [int(rp[r + 1] - rp[r]) for r in rows_of[:4]]))
            dC.free()
        ctx.close()
    print(over, "bad", len(bad), "of", reps, bad[:6], flush=True)
