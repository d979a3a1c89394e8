#!/bin/bash
# usage: bw.sh "<workload args>" ... ; prints phase times per workload
for w in "$@"; do timeout -k 10 280 python -u bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-also --workload $w > gpurun_out/bw.json 2> gpurun_out/bw.err || { echo "$w FAILED"; tail -3 gpurun_out/bw.err; continue; }; python - <<PY
import json; d=json.load(open("gpurun_out/bw.json")); p=d["whole_multiply"]["phase_ms"]
print("$w", "ms/step %.2f"%d["ms_per_step"], "GFLOP/s %.1f"%d["value"], "frac %.3f"%d["whole_multiply"]["frac_of_hbm_peak"], {k:round(v,2) for k,v in p.items()}, "nprod", d["config"]["nprod"], "nnzC", d["config"]["nnz_c"])
for l in d["launches"]:
    if l["ms"]>0.3: print("   ", l)
PY
done
