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
seed = int(sys.argv[1]); reps = int(sys.argv[2])
A, B, opts = fz.make_case(seed)
E = orc.spgemm(fz.to_o(A), fz.to_o(B))
for over in ({}, {"slice_sorted": 0}, {"big_sym_sorted": 0}, {"slice_sorted": 0, "big_sym_sorted": 0}):
    bad = []
    for rep in range(reps):
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
            dC.free()
        ctx.close()
    print(over, "bad", len(bad), "of", reps, bad[:6], flush=True)
