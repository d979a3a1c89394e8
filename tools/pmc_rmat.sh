#!/bin/bash
# Runs on the GPU box from the repo root: SQ counters of the R-MAT workload's kernels, two passes.
set -u
R=$PWD; OUT=$R/gpurun_out/pmc_rmat; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES --output-format csv -d $OUT/p1 -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --workload rmat "$@" > $OUT/p1.log 2>&1 || { tail -5 $OUT/p1.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/p2 -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --workload rmat "$@" > $OUT/p2.log 2>&1 || { tail -5 $OUT/p2.log; exit 1; }
cd $R
for d in p1 p2; do echo "== $d"; python3 tools/pmc_summary.py $(find $OUT/$d -name "*counter_collection.csv" | head -1) > $OUT/$d.txt 2>&1; grep -E "k_sym_big|k_num_slice|k_num_hash|k_sym_hash" $OUT/$d.txt; done
