#!/bin/bash
# GPU box, repo root: parity + fuzz tests, then bench_one.sh for the workloads named
set -o pipefail
export LIBC_FATAL_STDERR_=1  # glibc's abort messages (heap checks) to standard error, not to the terminal
timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q --timeout 200 --timeout-method thread > gpurun_out/t4.log 2>&1; rc=$?; tail -3 gpurun_out/t4.log; [ $rc -eq 0 ] || exit $rc
for w in "$@"; do bash tools/bench_one.sh $w || exit 1; done
