#!/bin/bash
# Runs on the GPU box from the repo root: kernel-trace stats + HBM traffic counters for the default bench
# (shell workload), and kernel-trace stats of the other workloads.  Each rocprofv3 pass is its own run.
set -u
R=$PWD; OUT=$R/gpurun_out/prof_final; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/trace.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/fetch.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/write.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $OUT/rdreq -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/rdreq.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/wrreq -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/wrreq.log 2>&1 || exit 1
for w in diel er rmat; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$w -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --workload $w > $OUT/trace_$w.log 2>&1 || exit 1
done
cd $R
for d in fetch write rdreq wrreq; do echo "== $d"; python3 tools/pmc_summary.py $(find $OUT/$d -name "*counter_collection.csv" | head -1) 2>/dev/null | grep -E "k_num_window_grp|k_sym_window" ; done
head -12 $(find $OUT/trace -name "*kernel_stats.csv" | head -1) | cut -c1-160
