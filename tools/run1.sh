set -o pipefail
timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q --timeout 200 --timeout-method thread > gpurun_out/t4.log 2>&1; rc=$?; tail -3 gpurun_out/t4.log; [ $rc -eq 0 ] || exit $rc
for w in shell diel; do
  timeout -k 10 300 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/b_$w.json 2> gpurun_out/b_$w.err || { tail -5 gpurun_out/b_$w.err; exit 1; }
  W=$w python - <<'PY'
import json, os
d=json.load(open('gpurun_out/b_%s.json' % os.environ['W']))
print(os.environ['W'], d['ms_per_step'], d['whole_multiply']['phase_ms'])
for l in d['launches']:
    if l['ms']>0.3: print(l)
PY
done
