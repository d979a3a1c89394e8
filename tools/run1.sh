set -o pipefail
for t in 256 512; do
SPGEMM_AMD_LIB=$PWD/build_var/libt$t.so bash tools/run2.sh | grep symbolic
SPGEMM_AMD_LIB=$PWD/build_var/libt$t.so timeout -k 10 300 python -u -m pytest tests/test_gpu_fuzz.py tests/test_gpu_parity.py -m gpu -x -q --timeout 200 --timeout-method thread 2>&1 | tail -2
done
