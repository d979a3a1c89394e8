set -o pipefail
SPGEMM_AMD_LIB=$PWD/build_var/libsymst.so timeout -k 10 300 python bench.py --workload rmat --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/b_st.json 2> gpurun_out/b_st.err; grep -A1 "long:" gpurun_out/b_st.err | tail -2
