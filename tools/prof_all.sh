#!/bin/bash
# Usage (GPU box, repo root): bash tools/prof_all.sh [workloads...]   (default: shell diel er rmat)
# Per workload: rocprofv3 kernel-trace stats of a short bench run, four --pmc passes (each its own
# run) for the HBM traffic per kernel launch, and one of instruction and LDS counters.  Results: gpurun_out/prof_final/<workload>/...
set -u
R=$PWD; OUT=$R/gpurun_out/prof_final; rm -rf $OUT; mkdir -p $OUT
WL=${@:-shell diel er rmat}
cd /tmp && export TMPDIR=/tmp
for w in $WL; do
  D=$OUT/$w; mkdir -p $D
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace -- python3 $R/bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-also > $D/trace.log 2>&1 || { tail -3 $D/trace.log; exit 1; }
  timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $D/fetch -- python3 $R/bench.py --workload $w --steps 1 --warmup 0 --no-cpu-baseline --no-also > $D/fetch.log 2>&1 || { tail -3 $D/fetch.log; exit 1; }
  timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $D/write -- python3 $R/bench.py --workload $w --steps 1 --warmup 0 --no-cpu-baseline --no-also > $D/write.log 2>&1 || { tail -3 $D/write.log; exit 1; }
  timeout -s KILL 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $D/rdreq -- python3 $R/bench.py --workload $w --steps 1 --warmup 0 --no-cpu-baseline --no-also > $D/rdreq.log 2>&1 || { tail -3 $D/rdreq.log; exit 1; }
  timeout -s KILL 200 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $D/wrreq -- python3 $R/bench.py --workload $w --steps 1 --warmup 0 --no-cpu-baseline --no-also > $D/wrreq.log 2>&1 || { tail -3 $D/wrreq.log; exit 1; }
  timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES --output-format csv -d $D/sq -- python3 $R/bench.py --workload $w --steps 1 --warmup 0 --no-cpu-baseline --no-also > $D/sq.log 2>&1 || { tail -3 $D/sq.log; exit 1; }
  echo "profiled $w"
done
