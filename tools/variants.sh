#!/bin/bash
# usage: variants.sh <bench args...> ; runs bench.py once per var_build/*.so and prints phase times
for so in var_build/*.so; do
  SPGEMM_AMD_LIB=$PWD/$so timeout -k 10 200 python -u bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-also "$@" > gpurun_out/v.json 2> gpurun_out/v.err || { echo "$so FAILED"; tail -3 gpurun_out/v.err; continue; }
  python - <<PY
import json; d=json.load(open("gpurun_out/v.json")); p=d["whole_multiply"]["phase_ms"]
print("$so", "ms/step %.2f"%d["ms_per_step"], "sym %.2f num %.2f ana %.2f"%(p["ms_symbolic"],p["ms_numeric"],p["ms_analyze"]), "GFLOP/s %.0f"%d["value"])
PY
done
