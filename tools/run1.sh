set -o pipefail
bash tools/run2.sh "$@" | grep symbolic || exit 1
timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q --timeout 200 --timeout-method thread > gpurun_out/t4.log 2>&1; rc=$?; tail -3 gpurun_out/t4.log; [ $rc -eq 0 ] || exit $rc
