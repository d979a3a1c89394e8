#!/bin/bash
# GPU box, repo root: kernel timeline of one distributed step on 1/P of the shell rows (P = $1, default 8)
set -o pipefail
P=${1:-8}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/dtrace
rm -rf $OUT && mkdir -p $OUT
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python3 tools/dist_step_run.py $P > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
F=$(find $OUT -name '*kernel_trace.csv' | head -1)
python tools/dist_step_timeline.py $F > gpurun_out/dtimeline_$P.txt && tail -80 gpurun_out/dtimeline_$P.txt
