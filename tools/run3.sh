set -o pipefail
timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q --timeout 200 --timeout-method thread > gpurun_out/t4.log 2>&1; rc=$?; tail -4 gpurun_out/t4.log; [ $rc -eq 0 ] || exit $rc
for w in rmat er diel shell; do
for cfg in "lane_order_adds=1" "lane_order_adds=0"; do
  opts=""; for o in $cfg; do opts="$opts --opt $o"; done
  timeout -k 10 300 python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline $opts > gpurun_out/b_$w.json 2> gpurun_out/b_$w.err || { tail -5 gpurun_out/b_$w.err; exit 1; }
  echo "== $w $cfg"
  W=$w python - <<'PY'
import json, os
d=json.load(open('gpurun_out/b_%s.json' % os.environ['W']))
print(d['ms_per_step'], d['whole_multiply']['phase_ms'])
for l in d['launches']:
    if l['ms']>0.15*d['ms_per_step']: print(l)
PY
done
done
