#!/bin/bash
# usage: tb.sh <tag> [bench args]  -- GPU parity tests then a short bench, summaries printed
tag=$1; shift
timeout -k 10 400 python -u -m pytest tests/ -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/t_$tag.log 2>&1; rc=$?
tail -3 gpurun_out/t_$tag.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python -u bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-also "$@" > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err || { tail -5 gpurun_out/bench_$tag.err; exit 1; }
python - <<PY
import json; d=json.load(open("gpurun_out/bench_$tag.json"))
print({k:d[k] for k in ("value","ms_per_step")}); print(d["roofline"]); print(d["whole_multiply"])
for l in d["launches"]:
    if l["ms"]>0.1: print(l)
PY
