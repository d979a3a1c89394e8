#!/bin/bash
# GPU box, repo root: bash tools/bench_one.sh <workload> [bench options...]  -> phase times and the launches over 5 % of the step
set -o pipefail
W=$1; shift
timeout -k 10 300 python bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-also "$@" > gpurun_out/b_$W.json 2> gpurun_out/b_$W.err || { tail -5 gpurun_out/b_$W.err; exit 1; }
W=$W python - <<'PY'
import json, os
d = json.load(open('gpurun_out/b_%s.json' % os.environ['W']))
print(os.environ['W'], round(d['ms_per_step'], 3), {k: round(v, 3) for k, v in d['whole_multiply']['phase_ms'].items()})
for l in d['launches']:
    if l['ms'] > 0.05 * d['ms_per_step']:
        print(l)
PY
