set -o pipefail
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/t_full.log 2>&1; rc=$?; tail -4 gpurun_out/t_full.log; [ $rc -eq 0 ] || exit $rc
bash tools/bench_all.sh
