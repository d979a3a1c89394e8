#!/bin/bash
# GPU box, repo root: the whole GPU test suite, then the four single-GPU bench lines
# (gpurun_out/t_full.log, gpurun_out/bench_<workload>.json).
set -o pipefail
export LIBC_FATAL_STDERR_=1  # glibc's abort messages (heap checks) to standard error, not to the terminal
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/t_full.log 2>&1; rc=$?; tail -4 gpurun_out/t_full.log; [ $rc -eq 0 ] || exit $rc
bash tools/bench_all.sh
