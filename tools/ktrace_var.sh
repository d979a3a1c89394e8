#!/bin/bash
# GPU box, repo root: bash tools/ktrace_var.sh <lib.so> <workload> <kernel substring>  -> durations of that kernel in a traced run of the variant library
set -o pipefail
L=$1; W=$2; K=$3
R=$PWD; OUT=$R/gpurun_out/ktv; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SPGEMM_AMD_LIB=$R/$L timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python3 $R/bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-also > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
cd $R
F=$(find $OUT -name '*kernel_trace.csv' | head -1)
K="$K" python - $F <<'PY'
import csv, sys, os
k = os.environ["K"]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(sys.argv[1])) if k in r["Kernel_Name"]]
print(k, "dispatches", len(d), "avg us %.1f" % (sum(d) / max(len(d), 1)), "max %.1f" % max(d))
PY
