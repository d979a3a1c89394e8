"""Single-process check of the per-rank multiply A_part x (B restricted to the marked sections)."""
import sys
import numpy as np
sys.path.insert(0, ".")
from oracle import oracle as orc
from spgemm_amd import capi, hostcsr, dist as sd

def to_o(h): return orc.OCsr(h.rows, h.cols, h.row_ptr, h.col_idx, h.val)
ctx = capi.Context(0)
A = hostcsr.banded_random(9000, 7, 40, seed=4); N = 512
world = int(sys.argv[1]) if len(sys.argv) > 1 else 4
for impl in ("baseline", "drop", "shuffle"):
    a_sh, b_sh = sd.make_shuffles(A.rows, A.cols, impl, seed=7)
    parts = capi.partition_naive(A.rows, A.cols, world)
    inv = np.empty(A.cols, dtype=np.int64); inv[b_sh] = np.arange(A.cols)
    Bfull = sd.relabel_cols(A, inv) if impl == "shuffle" else A
    sb = sd.section_bounds(A.rows, N)
    for rank in range(world):
        ops = sd.build_rank_operands(A, A, rank, world, parts, a_sh, b_sh, N)
        bits = np.ones(N, dtype=bool) if impl == "baseline" else orc.section_bitmap(to_o(ops.A_part), N)
        mask = np.zeros(A.rows, dtype=bool)
        for s in np.flatnonzero(bits): mask[sb[s]:sb[s + 1]] = True
        lens = np.where(mask, np.diff(Bfull.row_ptr.astype(np.int64)), 0)
        keep = np.repeat(mask, np.diff(Bfull.row_ptr.astype(np.int64)))
        rp = np.zeros(A.rows + 1, dtype=np.uint64); rp[1:] = np.cumsum(lens)
        Bl = hostcsr.HostCsr(A.rows, A.cols, rp, Bfull.col_idx[keep], Bfull.val[keep])
        ap = ops.A_part
        dA = ctx.from_host(ap.rows, ap.cols, ap.row_ptr, ap.col_idx, ap.val)
        dB = ctx.from_host(Bl.rows, Bl.cols, Bl.row_ptr, Bl.col_idx, Bl.val)
        dbits = ctx.section_bitmap(dA, N)
        C = ctx.multiply(dA, dB)
        st = ctx.stats()
        crp, cci, cv = C.to_host()
        E = orc.spgemm(to_o(ap), to_o(Bl))
        ok_bits = impl == "baseline" or np.array_equal(dbits, bits)
        bad = np.flatnonzero(np.diff(crp.astype(np.int64)) != np.diff(E.row_ptr.astype(np.int64)))
        same = len(bad) == 0 and np.array_equal(cci, E.col_idx) and np.array_equal(cv, E.val)
        print(impl, rank, "bits", ok_bits, "same", same, "badrows", bad[:10], len(bad),
              {k: st[k] for k in st if k.startswith("rows_")}, flush=True)
        if len(bad):
            r = bad[0]
            print("  row", r, "lenA", ap.row_ptr[r+1]-ap.row_ptr[r], "got", crp[r+1]-crp[r], "want", E.row_ptr[r+1]-E.row_ptr[r])
        C.free(); dA.free(); dB.free()
