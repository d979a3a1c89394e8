#!/bin/bash
# GPU box, repo root: the GPU suite a few times in fresh processes with glibc's abort
# message on standard error; stops at the first run that fails (a host-side abort seen once in ~5 runs).
export LIBC_FATAL_STDERR_=1
for i in 1 2 3 4; do
  timeout -k 10 300 python -X faulthandler -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/hunt_$i.log 2>&1
  rc=$?
  echo "run $i rc $rc"; tail -1 gpurun_out/hunt_$i.log
  if [ $rc -ne 0 ]; then grep -v "^  File" gpurun_out/hunt_$i.log | head -40; exit $rc; fi
done
