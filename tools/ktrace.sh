#!/bin/bash
# GPU box, repo root: bash tools/ktrace.sh <workload> [bench options]  -> kernels of the last step, in launch order
set -o pipefail
W=$1; shift
R=$PWD; OUT=$R/gpurun_out/ktrace_$W; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python3 $R/bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-also "$@" > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
cd $R
F=$(find $OUT -name '*kernel_trace.csv' | head -1)
python tools/dist_step_timeline.py $F k_sym_clear
