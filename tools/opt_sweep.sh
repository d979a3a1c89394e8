#!/bin/bash
# GPU box, repo root: bash tools/opt_sweep.sh <workload> <option> <value>...  -> phase times of bench.py with that option set to each value
W=$1; O=$2; shift 2
for v in "$@"; do
  timeout -k 10 200 python -u bench.py --workload $W --steps 5 --warmup 2 --no-cpu-baseline --no-also --opt $O=$v > gpurun_out/os.json 2> gpurun_out/os.err || { echo "$O=$v FAILED"; tail -3 gpurun_out/os.err; exit 1; }
  O=$O V=$v W=$W python - <<'PY'
import json, os
d = json.load(open("gpurun_out/os.json")); p = d["whole_multiply"]["phase_ms"]
print(os.environ["W"], os.environ["O"], os.environ["V"], "ms/step %.3f" % d["ms_per_step"], "ana %.3f sym %.3f num %.3f" % (p["ms_analyze"], p["ms_symbolic"], p["ms_numeric"]), flush=True)
PY
done
