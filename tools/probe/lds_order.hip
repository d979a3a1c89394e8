// Probe: does ONE ds_add_f64 instruction whose lanes hit the same LDS address apply the lanes'
// operands in ascending lane order?  Each trial: 64 lanes pick one of `nslots` slots (random) and add
// a double of wildly varying magnitude; the result is compared bit for bit with a sequential sum
// in lane order, in descending lane order, and classified.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ uint64_t mix(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
  return x;
}

__global__ void __launch_bounds__(256) probe(uint32_t trials, uint32_t nslots, uint32_t active_mod, uint32_t stride, unsigned long long *out) {
  __shared__ double acc[4][1024];
  __shared__ double xs[4][64];
  __shared__ uint32_t ps[4][64];
  const uint32_t w = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  unsigned long long bad_asc = 0, bad_desc = 0;
  for (uint32_t t = 0; t < trials; ++t) {
    const uint64_t h = mix(((uint64_t)blockIdx.x << 40) ^ ((uint64_t)w << 34) ^ ((uint64_t)t << 8) ^ lane);
    const uint32_t slot = (uint32_t)(h % nslots);
    const int e = (int)((h >> 20) % 60) - 30;
    double x = ldexp(1.0 + (double)((h >> 32) & 0xFFFFF) / 1048576.0, e);
    if (h & (1ull << 60)) x = -x;
    const bool active = active_mod == 0 || (mix(h) % active_mod) != 0;
    for (uint32_t i = lane; i < 1024; i += 64) acc[w][i] = 0.0;
    xs[w][lane] = x;
    ps[w][lane] = active ? slot : 0xFFFFFFFFu;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_barrier();
    if (active) __hip_atomic_fetch_add(&acc[w][(slot * stride) & 1023u], x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_barrier();
    if (lane < nslots) {
      double up = 0.0, down = 0.0;
      for (int l = 0; l < 64; ++l)
        if (ps[w][l] == lane) up += xs[w][l];
      for (int l = 63; l >= 0; --l)
        if (ps[w][l] == lane) down += xs[w][l];
      const double got = acc[w][(lane * stride) & 1023u];
      if (__double_as_longlong(got) != __double_as_longlong(up)) ++bad_asc;
      if (__double_as_longlong(got) != __double_as_longlong(down)) ++bad_desc;
    }
    __builtin_amdgcn_s_barrier();
  }
  if (bad_asc) atomicAdd(out, bad_asc);
  if (bad_desc) atomicAdd(out + 1, bad_desc);
}

int main() {
  unsigned long long *d, h[2];
  (void)hipMalloc(&d, 16);
  const uint32_t slots[] = {1, 2, 3, 5, 8, 16, 31, 32, 33, 64};
  for (uint32_t stride : {1u, 2u, 16u, 17u, 32u, 33u, 64u})
  for (uint32_t am : {0u, 3u})
    for (uint32_t ns : slots) {
      if ((uint64_t)ns * stride > 1024 && (stride % 2 == 0)) continue; // distinct slots have to stay distinct
      (void)hipMemset(d, 0, 16);
      hipLaunchKernelGGL(probe, dim3(1024), dim3(256), 0, 0, 100u, ns, am, stride, d);
      if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
      (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
      printf("stride %2u slots %2u inactive 1/%u: sums checked %llu  differ from ascending-lane order %llu  from descending %llu\n", stride, ns, am,
             (unsigned long long)1024 * 4 * 100 * ns, h[0], h[1]);
    }
  return 0;
}
