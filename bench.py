#!/usr/bin/env python3
"""Benchmark of the SpGEMM hot path: C = A * A, fp64, on N MI355X GPUs of one node.

One step = one full multiply (row analysis, symbolic pass, row-pointer scan, numeric pass and,
for N > 1, the exchange of the B row sections) with A already resident in HBM.
Prints ONE JSON line (rank 0).

  python bench.py --gpus 1 --steps 5 --warmup 2
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workloads (synthetic; there is no network to fetch SuiteSparse files):
  shell  stand-in for Schenk_AFE/af_shell10: same order (1,508,065) and ~35 entries per row
  er     Erdos-Renyi, n rows, 16 draws per row
  rmat   R-MAT (0.57, 0.19, 0.19, 0.05), 2^scale rows, edge factor 4
  mtx    a Matrix Market file given with --matrix (e.g. the real af_shell10)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

# RCCL and device-tensor sharing between the ranks of a node need the dmabuf IPC mode on this driver
# (the boxes export it already; a launcher that scrubs the environment must not lose it)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak (spec); the achievable rate is measured in the run (stream copy)


def make_workload(args):
    from spgemm_amd import hostcsr
    if args.workload == "shell":
        A = hostcsr.shell_like()
        name = "af_shell10-like synthetic shell (n=1508065, ~35 nnz/row), C=A*A"
    elif args.workload == "diel":
        A = hostcsr.diel_like()
        name = "dielFilterV3real-like synthetic 3-D grid (n=1104192, ~81 nnz/row), C=A*A"
    elif args.workload == "er":
        A = hostcsr.erdos_renyi(args.n, 16, seed=42)
        name = f"Erdos-Renyi n={args.n}, 16 draws/row, C=A*A"
    elif args.workload == "rmat":
        A = hostcsr.rmat(args.scale, args.edge_factor, seed=1)
        name = f"R-MAT scale {args.scale}, edge factor {args.edge_factor}, C=A*A"
    elif args.workload == "mtx":
        A = hostcsr.read_mtx(args.matrix, eliminate_zeros=True)
        name = f"{os.path.basename(args.matrix)} C=A*A"
    else:
        raise SystemExit(f"unknown workload {args.workload}")
    return A, name


def csrc_digest():
    """SHA-1 over the kernel and C-ABI sources: names the build a committed profile was made from."""
    import glob
    import hashlib
    h = hashlib.sha1()
    for f in sorted(glob.glob(os.path.join(ROOT, "spgemm_amd", "csrc", "*.h")) +
                    glob.glob(os.path.join(ROOT, "spgemm_amd", "csrc", "*.hip"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def measured_traffic(workload, kernel):
    """HBM bytes per launch of `kernel` and the SQ counters of that launch, from the committed PMC
    profile of this workload (profiles/*_traffic_<workload>.json, made by tools/prof_all.sh +
    tools/make_traffic.py: separate rocprofv3 --pmc passes of this same command, the read side
    rebuilt from the per-size request counters because FETCH_SIZE tallies 128-B requests at 64 B on
    gfx950).  Counters cannot be read from inside the run, so the figures are those of the last
    profiled build: they are returned only when that profile was made from THESE sources (the
    profile records a digest of spgemm_amd/csrc) and holds this kernel; else everything is None and
    `stale` says why."""
    import glob
    out = {"traffic": None, "file": None, "sq": None, "stale": None}
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"*_traffic_{workload}.json")))
    if not files:
        out["stale"] = "no profile committed for this workload"
        return out
    prof = json.load(open(files[-1]))
    out["file"] = os.path.relpath(files[-1], ROOT)
    if prof.get("csrc_sha1") != csrc_digest():
        out["stale"] = "the committed profile was made from other kernel sources than the ones timed here"
        return out
    # template arguments differ between instantiations of one kernel: the largest launch counts
    match = [v for name, v in prof["kernels"].items() if name == kernel or name.startswith(kernel + "<")]
    if not match:
        out["stale"] = "the committed profile holds no launch of this kernel"
        return out
    best = max(match, key=lambda m: m["hbm_bytes"])
    out["traffic"], out["sq"] = best["hbm_bytes"], best.get("sq")
    return out


def issue_bounds(sq, kernel_ms, n_cu=256, clock_ghz=2.4):
    """Lower bounds on a kernel's time from what it issues, out of the SQ counters of the same
    launch: every vector instruction holds one of the 4 * n_cu SIMDs for 4 cycles, every LDS-array
    cycle one of the n_cu LDS units.  The clock is the peak one, so the bounds are optimistic."""
    if not sq:
        return None
    valu_ms = sq["SQ_INSTS_VALU"] * 4.0 / (4 * n_cu) / (clock_ghz * 1e6)
    # a SIMD's issue slot comes round every 4 cycles and takes one scalar instruction at most
    salu_ms = sq.get("SQ_INSTS_SALU", 0.0) * 4.0 / (4 * n_cu) / (clock_ghz * 1e6)
    lds_ms = sq["SQ_LDS_IDX_ACTIVE"] / n_cu / (clock_ghz * 1e6)
    return {"valu_issue_ms": valu_ms, "salu_issue_ms": salu_ms, "lds_array_ms": lds_ms,
            "valu_frac_of_kernel_time": valu_ms / kernel_ms, "salu_frac_of_kernel_time": salu_ms / kernel_ms,
            "lds_frac_of_kernel_time": lds_ms / kernel_ms, "valu_wave_instructions": sq["SQ_INSTS_VALU"],
            "salu_wave_instructions": sq.get("SQ_INSTS_SALU"),
            "lds_bank_conflict_share": sq["SQ_LDS_BANK_CONFLICT"] / max(sq["SQ_LDS_IDX_ACTIVE"], 1.0),
            "assumed_clock_GHz": clock_ghz,
            "note": "second bound: time the kernel's own vector / scalar instructions / LDS cycles need at peak clock"}


def kernel_name(launch):
    """Kernel behind a numeric launch record (spgemm_get_launches)."""
    if launch["path"] == "window_groups":
        return "k_num_window_grp"
    if launch["path"] == "window_group_runs":
        return "k_num_grp_run"
    if launch["path"] == "big_rows":
        return "k_num_slice"
    if launch["path"] == "hash" and launch["lanes"] == 1:
        return "k_num_short"
    return f"k_num_{launch['path']}<{'true' if launch['lanes'] == 0 else 'false'}>"


def algorithmic_bytes(nnz_a, nprod, nnz_c, rows):
    """Row-wise Gustavson traffic model: every A entry, every product's B entry and every C
    entry moves once as {u32 index, f64 value}; plus the A (u32) and C (u64) row pointers."""
    return 12 * (nnz_a + nprod + nnz_c) + 4 * (rows + 1) + 8 * (rows + 1)


def compulsory_bytes(nnz_a, nnz_c, rows):
    """The stricter figure: A and B (= A) read once, C written once, plus the row pointers."""
    return 12 * (nnz_a + nnz_c) + 4 * (rows + 1) + 8 * (rows + 1)


def cpu_baseline(A, target_seconds=10.0):
    """Times the CPU oracle on a bounded sample of the same workload (a block of rows of A times all
    of A), on every thread OpenMP gives this process:
      * the port of the reference's DropParallel loop (one map per output row, OpenMP over rows).
        The port's map is open addressing; the reference's std::unordered_map is several times
        slower (0.09 GFLOP/s on 8 cores for Erdos-Renyi 2^16), so this line flatters the CPU;
      * a tuned Gustavson with a dense accumulator per thread, as the fairer second line.
    Checker code used as a baseline; it is not part of the product path."""
    from oracle import oracle as orc
    cores = orc.lib().oracle_max_threads()   # OMP_NUM_THREADS if set (the GPU box sets it), else all cores
    full = orc.OCsr(A.rows, A.cols, A.row_ptr, A.col_idx, A.val)
    probe_rows = min(A.rows, 20000)
    start = (A.rows - probe_rows) // 2
    sub = full.select_rows(np.arange(start, start + probe_rows))
    _, st = orc.spgemm(sub, full, with_stats=True, nthreads=cores)
    rate = st["nprod"] / max(st["seconds"], 1e-9)  # products / s
    prod_per_row = max(st["nprod"] / probe_rows, 1.0)
    rows = int(min(A.rows, max(probe_rows, target_seconds * rate / prod_per_row)))
    start = (A.rows - rows) // 2
    sub = full.select_rows(np.arange(start, start + rows))
    # repeated until about target_seconds of CPU work have been timed (the whole workload is short of it
    # on a many-core host)
    seconds, nprod, reps = 0.0, 0, 0
    while reps == 0 or (seconds < 0.8 * target_seconds and reps < 32):
        _, st = orc.spgemm(sub, full, with_stats=True, nthreads=cores)
        seconds += st["seconds"]
        nprod += st["nprod"]
        reps += 1
    gflops = 2.0 * nprod / seconds / 1e9
    # second line: dense accumulator per thread, same sample, about a third of the time budget
    s2, n2, r2 = 0.0, 0, 0
    while r2 == 0 or (s2 < 0.3 * target_seconds and r2 < 32):
        _, st = orc.spgemm_spa(sub, full, with_stats=True, nthreads=cores)
        s2 += st["seconds"]
        n2 += st["nprod"]
        r2 += 1
    return {"value": gflops, "unit": "GFLOP/s", "cores": cores, "kind": "port",
            "host_cpus": os.cpu_count(),
            "threads_note": "cores = OpenMP's thread limit for this process (OMP_NUM_THREADS when the harness "
                            "sets it), host_cpus = os.cpu_count()",
            "sample": f"rows [{start}, {start + rows}) of A times all of A, {reps} time(s): {nprod} products "
                      f"in {seconds:.2f} s (accumulate loop only, as the reference times it); one "
                      f"open-addressing map per output row (faster than the reference's std::unordered_map)",
            "tuned_dense_spa": {"value": 2.0 * n2 / s2 / 1e9, "unit": "GFLOP/s", "cores": cores,
                                "sample": f"same rows, {r2} time(s): {n2} products in {s2:.2f} s, count pass + "
                                          f"fill pass with a dense accumulator per thread"}}


def measure_workload(ctx, A, workload_key, steps, warmup, opts=(), alias_b=False):
    """Times `steps` multiplies C = A * A on the context (after `warmup` untimed ones, A resident in
    HBM; B is a second copy of the matrix in HBM, as the reference reads its two operands apart, unless
    alias_b) and returns (ms per step, stats, launches with their mean durations, roofline of the
    dominant numeric launch, whole-multiply figures)."""
    import torch
    dA = ctx.from_host(A.rows, A.cols, A.row_ptr, A.col_idx, A.val)
    dB = dA if alias_b else ctx.from_host(A.rows, A.cols, A.row_ptr, A.col_idx, A.val)
    ctx.set_option("profile_launches", 1)
    for o in opts:
        opt_name, opt_value = o.split("=")
        ctx.set_option(opt_name, int(opt_value))
    for _ in range(warmup):
        ctx.multiply(dA, dB).free()
    ctx.synchronize()
    torch.cuda.synchronize()
    per_launch_ms = None
    phase_ms = {}
    t0 = time.perf_counter()
    for _ in range(steps):
        C = ctx.multiply(dA, dB)
        C.free()
        st = ctx.stats()
        ms = ctx.launch_ms()
        per_launch_ms = ms if per_launch_ms is None else per_launch_ms + ms
        for k in ("ms_analyze", "ms_symbolic", "ms_scan", "ms_numeric", "ms_compact", "ms_total"):
            phase_ms[k] = phase_ms.get(k, 0.0) + st[k]
    ctx.synchronize()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    ms_per_step = (t1 - t0) * 1e3 / steps
    per_launch_ms = per_launch_ms / steps
    phase_ms = {k: v / steps for k, v in phase_ms.items()}

    # one untimed multiply that also sums the rows of every launch (bytes per launch)
    ctx.set_option("profile_launches", 2)
    ctx.multiply(dA, dB).free()
    launches = ctx.launches()
    st = ctx.stats()
    ctx.set_option("profile_launches", 0)
    if dB is not dA:
        dB.free()
    dA.free()
    for l, ms in zip(launches, per_launch_ms):
        l["ms"] = float(ms)
    num = [l for l in launches if l["phase"] == "numeric"]
    dom = max(num, key=lambda l: l["ms"]) if num else None
    copy_gbs = ctx.stream_copy_rate(1 << 30, 5)  # what a plain copy gets out of this GPU's HBM, GB/s
    roofline = None
    if dom:
        b = algorithmic_bytes(dom["nnz_a"], dom["nprod"], dom["nnz_c"], dom["rows"])
        bc = compulsory_bytes(dom["nnz_a"], dom["nnz_c"], dom["rows"])
        sec = dom["ms"] * 1e-3
        achieved = bc / sec / 1e9
        kname = kernel_name(dom)
        prof = measured_traffic(workload_key, kname)
        traffic = prof["traffic"]
        roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                    "basis": "compulsory bytes of the launch's rows (their A entries and as many B entries read once, "
                             "their C entries written once, the row pointers) / kernel time: no kernel moves less, so "
                             "frac <= 1 by construction",
                    "kernel": kname, "kernel_ms": dom["ms"], "kernel_rows": dom["rows"],
                    "kernel_compulsory_bytes": bc, "kernel_algorithmic_bytes": b,
                    "frac_model_no_reuse": b / sec / 1e9 / HBM_PEAK_GBS,
                    "model_no_reuse_note": "algorithmic bytes (12 B per product: every product's B entry from memory, no "
                                           "reuse of B rows) / kernel time / peak: the traffic MODEL of row-wise Gustavson, "
                                           "not a bound - above 1 where neighbouring rows share B rows in L2 / Infinity Cache",
                    "measured_copy_GBs": copy_gbs, "frac_of_measured_copy": achieved / copy_gbs if copy_gbs else None,
                    "traffic_source": prof["file"], "traffic_stale": prof["stale"]}
        bounds = {"hbm_compulsory": bc / (HBM_PEAK_GBS * 1e9) / sec}
        if traffic:
            hbm_rate = traffic / sec / 1e9
            roofline.update({"traffic_over_algorithmic": traffic / b, "traffic_over_compulsory": traffic / bc,
                             "hbm_rate_GBs": hbm_rate, "hbm_rate_frac_of_peak": hbm_rate / HBM_PEAK_GBS,
                             "limited_by": ("hbm" if copy_gbs and hbm_rate > 0.6 * copy_gbs else
                                            "instruction issue and LDS, not HBM (measured HBM rate well below the copy rate)")})
            bounds["hbm_measured_traffic"] = hbm_rate / HBM_PEAK_GBS
            ib = issue_bounds(prof["sq"], dom["ms"])
            if ib:
                roofline["issue_bound"] = ib
                bounds["lds_array"] = min(ib["lds_frac_of_kernel_time"], 1.0)
                bounds["valu_issue"] = min(ib["valu_frac_of_kernel_time"], 1.0)
                bounds["salu_issue"] = min(ib["salu_frac_of_kernel_time"], 1.0)
        tight = max(bounds, key=bounds.get)
        roofline["bounds"] = {"frac_of_each": bounds, "tightest": tight, "frac_of_tightest": bounds[tight],
                              "note": "each entry = (time that bound alone would need) / kernel time, all <= 1; the LDS "
                                      "and issue bounds exist only beside a profile of these very sources"}
    b_all = st["bytes_algorithmic"]
    bc_all = compulsory_bytes(st["nnz_a"], st["nnz_c"], st["rows_a"])
    sec_all = ms_per_step * 1e-3
    whole = {"algorithmic_bytes": b_all, "compulsory_bytes": bc_all,
             "compulsory_frac_of_hbm_peak": bc_all / sec_all / 1e9 / HBM_PEAK_GBS,
             "frac_model_no_reuse": b_all / sec_all / 1e9 / HBM_PEAK_GBS,
             "model_no_reuse_GBs": b_all / sec_all / 1e9,
             "model_no_reuse_over_measured_copy": b_all / sec_all / 1e9 / copy_gbs if copy_gbs else None,
             "measured_copy_GBs": copy_gbs, "phase_ms": phase_ms}
    return ms_per_step, st, launches, roofline, whole


def run_single(args):
    import torch  # device selection / presence check only
    from spgemm_amd import capi, hostcsr
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device")
    A, name = make_workload(args)
    ctx = capi.Context(0)
    ms_per_step, st, launches, roofline, whole = measure_workload(ctx, A, args.workload, args.steps, args.warmup, args.opt, args.alias_b)
    gflops = 2.0 * st["nprod"] / (ms_per_step * 1e-3) / 1e9
    out = {
        "metric": "SpGEMM GFLOP/s (useful flops, 2*nprod/t) C=A*A fp64",
        "value": gflops, "unit": "GFLOP/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic" if args.workload != "mtx" else "file",
        "config": {"workload": name, "rows": A.rows, "nnz_a": A.nnz, "nprod": st["nprod"], "nnz_c": st["nnz_c"],
                   "impl": "baseline", "parallelism": "1 GPU",
                   "operands": "one device matrix as A and B" if args.alias_b else "A and B: two copies of the matrix in HBM",
                   "stand_in": "synthetic matrix of the named SuiteSparse matrix's order and row length; the real file "
                               "(not fetchable here) runs through --workload mtx" if args.workload in ("shell", "diel") else None},
        "roofline": roofline,
        "whole_multiply": whole,
        "launches": [{k: l[k] for k in ("phase", "path", "lanes", "rows", "ms", "lds_bytes", "nprod", "nnz_c")}
                     for l in launches],
    }
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(A)
    # the other workloads of the benchmark set, a bounded number of steps each: R-MAT scale 22 (the
    # second workload the target names), the 3-D grid stand-in and Erdos-Renyi 2^22
    if args.workload == "shell" and not args.no_also:
        del A
        out["also"] = []
        k = max(1, min(args.steps, 5))
        for key, make, label in (("rmat", lambda: hostcsr.rmat(22, 4, seed=1), "R-MAT scale 22, edge factor 4, C=A*A"),
                                 ("diel", hostcsr.diel_like, "dielFilterV3real-like synthetic 3-D grid (n=1104192, ~81 nnz/row), C=A*A"),
                                 ("er", lambda: hostcsr.erdos_renyi(1 << 22, 16, seed=42), "Erdos-Renyi n=4194304, 16 draws/row, C=A*A")):
            M = make()
            ctx.close()  # (a context of its own per workload: the pooled buffers of the last one, 100 GB after R-MAT, go back)
            ctx = capi.Context(0)
            ms2, st2, l2, roof2, whole2 = measure_workload(ctx, M, key, k, 1, args.opt, args.alias_b)
            out["also"].append({
                "config": {"workload": label, "rows": M.rows, "nnz_a": M.nnz, "nprod": st2["nprod"], "nnz_c": st2["nnz_c"],
                           "impl": "baseline", "parallelism": "1 GPU",
                           "operands": "one device matrix as A and B" if args.alias_b else "A and B: two copies of the matrix in HBM",
                           "stand_in": "synthetic matrix of the named SuiteSparse matrix's order and row length" if key == "diel" else None},
                "steps": k, "warmup": 1, "ms_per_step": ms2, "value": 2.0 * st2["nprod"] / (ms2 * 1e-3) / 1e9, "unit": "GFLOP/s",
                "roofline": roof2, "whole_multiply": whole2,
                "launches": [{kk: l[kk] for kk in ("phase", "path", "lanes", "rows", "ms", "lds_bytes", "nprod", "nnz_c")} for l in l2],
            })
            del M
    ctx.close()
    print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="shell", choices=["shell", "diel", "er", "rmat", "mtx"])
    ap.add_argument("--n", type=int, default=1 << 22, help="rows of the Erdos-Renyi workload")
    ap.add_argument("--scale", type=int, default=22, help="log2 rows of the R-MAT workload")
    ap.add_argument("--edge-factor", type=int, default=4)
    ap.add_argument("--matrix", default=None, help="Matrix Market file for --workload mtx")
    ap.add_argument("--impl", default="drop", choices=["baseline", "drop", "shuffle"],
                    help="distributed variant (N > 1): all sections, bitmap-filtered sections, "
                         "or random row/column shuffle + bitmap")
    ap.add_argument("--partition", default="naive", choices=["naive", "balanced"],
                    help="row blocks of the distributed variant: equal row counts, or equal product counts")
    ap.add_argument("--stages", type=int, default=1,
                    help="distributed variant: > 1 cuts a rank's rows into that many blocks and overlaps the "
                         "exchange of the next block with the multiply of the current one")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true",
                    help="default run (shell) only: leave out the blocks of the other workloads (R-MAT scale 22, 3-D grid, Erdos-Renyi)")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE",
                    help="engine option (spgemm_set_option), e.g. --opt row_groups=0")
    ap.add_argument("--alias-b", action="store_true",
                    help="pass one device matrix as both operands instead of two copies of it (its B reads then warm the caches for A)")
    ap.add_argument("--dist", action="store_true",
                    help="use the distributed operator even at N = 1 (rehearsal of the N > 1 path)")
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # Standard output carries the one JSON line and nothing else: libraries that write to file
    # descriptor 1 (RCCL prints a version banner when its first communicator is made) are sent to
    # standard error for the length of the run.
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    sys.stdout = json_out
    try:
        if args.gpus == 1 and world == 1 and not args.dist:
            run_single(args)
        else:
            from spgemm_amd import dist
            dist.bench_main(args)
    finally:
        json_out.flush()


if __name__ == "__main__":
    main()
