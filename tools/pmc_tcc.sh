#!/bin/bash
# Usage (GPU box, repo root): bash tools/pmc_tcc.sh <workload> <tag> <kernel fragments> [extra bench args]
# Two rocprofv3 --pmc passes of L2 / fabric counters over one multiply; sums of the named kernels go to
# gpurun_out/pmc_<tag>/t{1,2}.txt
set -u
W=$1; TAG=$2; K=$3; shift 3
R=$PWD; OUT=$R/gpurun_out/pmc_$TAG; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum TCC_REQ_sum --output-format csv -d $OUT/t1 -- python3 $R/bench.py --workload $W --steps 1 --warmup 0 --no-cpu-baseline --no-also "$@" > $OUT/t1.log 2>&1 || { tail -5 $OUT/t1.log; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/t2 -- python3 $R/bench.py --workload $W --steps 1 --warmup 0 --no-cpu-baseline --no-also "$@" > $OUT/t2.log 2>&1 || { tail -5 $OUT/t2.log; exit 1; }
cd $R
for p in t1 t2; do python3 tools/pmc_summary.py $(find $OUT/$p -name "*counter_collection.csv" | head -1) "$K" > $OUT/$p.txt; cat $OUT/$p.txt; done
