#!/bin/bash
# Usage (GPU box, repo root): bash tools/pmc_sq.sh <workload> <tag> [extra bench args]
# Four rocprofv3 --pmc passes (two of 8 SQ counters, two of TCC counters; each its own run) over one multiply of the
# workload; per-kernel sums go to gpurun_out/pmc_<tag>/p{1,2,3,4}.txt
set -u
W=$1; TAG=$2; shift 2
R=$PWD; OUT=$R/gpurun_out/pmc_$TAG; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM --output-format csv -d $OUT/p1 -- python3 $R/bench.py --workload $W --steps 1 --warmup 0 --no-cpu-baseline --no-also "$@" > $OUT/p1.log 2>&1 || { tail -5 $OUT/p1.log; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/p2 -- python3 $R/bench.py --workload $W --steps 1 --warmup 0 --no-cpu-baseline --no-also "$@" > $OUT/p2.log 2>&1 || { tail -5 $OUT/p2.log; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $OUT/p3 -- python3 $R/bench.py --workload $W --steps 1 --warmup 0 --no-cpu-baseline --no-also "$@" > $OUT/p3.log 2>&1 || { tail -5 $OUT/p3.log; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/p4 -- python3 $R/bench.py --workload $W --steps 1 --warmup 0 --no-cpu-baseline --no-also "$@" > $OUT/p4.log 2>&1 || { tail -5 $OUT/p4.log; exit 1; }
cd $R
for p in p1 p2 p3 p4; do python3 tools/pmc_summary.py $(find $OUT/$p -name "*counter_collection.csv" | head -1) "${KERNELS:-k_}" > $OUT/$p.txt; cat $OUT/$p.txt; done
