set -o pipefail
# usage: bash tools/run2.sh [bench opts...]   -> R-MAT bench, prints phase times and the big launches
timeout -k 10 300 python bench.py --workload rmat --steps 3 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/b_rmat.json 2> gpurun_out/b_rmat.err || { tail -5 gpurun_out/b_rmat.err; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/b_rmat.json'))
print('RMAT', d['ms_per_step'], d['whole_multiply']['phase_ms'])
for l in d['launches']:
    if l['ms']>20: print(l)
PY
