#!/bin/bash
# GPU box, repo root: the fuzz cases through a build of the library whose HOST code runs under
# AddressSanitizer (the device code is the shipped one; GPU sanitizers are not used).
#   build (CPU container):  bash tools/asan_replay/run.sh build
#   run   (GPU box):        bash tools/asan_replay/run.sh run
set -o pipefail
D=tools/asan_replay; B=$D/_build
if [ "$1" = build ]; then
  mkdir -p $B
  F="--offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -Xarch_host -fsanitize=address -Xarch_host -fno-omit-frame-pointer -Wno-unused-function"
  /opt/rocm/bin/hipcc $F -c -o $B/capi.o spgemm_amd/csrc/spgemm_capi.hip || exit 1
  /opt/rocm/bin/hipcc $F -Wno-unused-parameter -c -o $B/shuffle.o spgemm_amd/csrc/spgemm_shuffle.hip || exit 1
  /opt/rocm/lib/llvm/bin/clang++ -O1 -g -std=c++17 -fsanitize=address -fno-omit-frame-pointer -c -o $B/replay.o $D/replay.cpp || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -Xarch_host -fsanitize=address -o $B/replay $B/replay.o $B/capi.o $B/shuffle.o || exit 1
  echo built $B/replay
else
  python3 $D/dump_inputs.py gpurun_out/fuzz_inputs.bin || exit 1
  # protect_shadow_gap=0: the HIP runtime maps address ranges that ASan's shadow gap would claim;
  # detect_leaks=0: the runtime's own process-lifetime allocations are not what is looked for
  ASAN_OPTIONS=protect_shadow_gap=0:detect_leaks=0:verify_asan_link_order=0:abort_on_error=0 \
    timeout -k 10 280 $B/replay gpurun_out/fuzz_inputs.bin > gpurun_out/asan_replay.log 2>&1
  rc=$?; tail -25 gpurun_out/asan_replay.log; exit $rc
fi
