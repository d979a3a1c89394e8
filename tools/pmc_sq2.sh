#!/bin/bash
# Usage (GPU box, repo root): bash tools/pmc_sq2.sh <workload> <tag> [extra bench args]
# Two rocprofv3 --pmc passes of SQ counters (each its own run) over one multiply of the workload; per-kernel figures
# of the largest dispatch go to gpurun_out/pmc_<tag>/p{1,2}.txt
set -u
W=$1; TAG=$2; shift 2
R=$PWD; OUT=$R/gpurun_out/pmc_$TAG; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM --output-format csv -d $OUT/p1 -- python3 $R/bench.py --workload $W --steps 1 --warmup 0 --no-cpu-baseline --no-also "$@" > $OUT/p1.log 2>&1 || { tail -5 $OUT/p1.log; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/p2 -- python3 $R/bench.py --workload $W --steps 1 --warmup 0 --no-cpu-baseline --no-also "$@" > $OUT/p2.log 2>&1 || { tail -5 $OUT/p2.log; exit 1; }
cd $R
for p in p1 p2; do python3 tools/pmc_summary.py $(find $OUT/$p -name "*counter_collection.csv" | head -1) "${KERNELS:-k_}" > $OUT/$p.txt; cat $OUT/$p.txt; done
