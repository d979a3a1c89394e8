// Probe: what a divergent vector load costs.  Every lane of every wavefront loads W bytes from a
// random 128-byte line of a 1-GiB buffer (or of a 2-MiB one: L2 hits), 8 independent loads in flight
// per lane; reports wave-instructions per microsecond per CU and bytes fetched.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

template <int W> struct Vec;
template <> struct Vec<4> { using t = uint32_t; };
template <> struct Vec<8> { using t = uint2; };
template <> struct Vec<16> { using t = uint4; };
__device__ uint32_t fold(uint32_t v) { return v; }
__device__ uint32_t fold(uint2 v) { return v.x ^ v.y; }
__device__ uint32_t fold(uint4 v) { return v.x ^ v.y ^ v.z ^ v.w; }

template <int W, bool DIVERGENT>
__global__ void __launch_bounds__(256) gather(const unsigned char *__restrict__ buf, uint64_t lines, uint32_t iters, uint32_t *__restrict__ out) {
  using T = typename Vec<W>::t;
  uint64_t x = (uint64_t)blockIdx.x * 256 + threadIdx.x + 12345;
  uint32_t acc = 0;
  for (uint32_t it = 0; it < iters; ++it) {
    T v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      x = x * 6364136223846793005ull + 1442695040888963407ull;
      uint64_t line = (x >> 20) % lines;
      if (!DIVERGENT) line = (line & ~63ull) | (threadIdx.x & 63) / (128 / W ? 128 / W : 1); // lanes of a wave share lines
      const uint64_t off = DIVERGENT ? 0 : ((threadIdx.x & 63) % (128 / W)) * W;
      v[u] = *reinterpret_cast<const T *>(buf + line * 128 + off);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc ^= fold(v[u]);
  }
  if (acc == 0x12345678u) out[0] = acc;
}

template <int W, bool D> void run(const unsigned char *buf, uint64_t lines, const char *what, uint32_t *out) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  const uint32_t blocks = 256 * 8, iters = 200;
  hipLaunchKernelGGL((gather<W, D>), dim3(blocks), dim3(256), 0, 0, buf, lines, 20u, out);
  (void)hipEventRecord(a);
  hipLaunchKernelGGL((gather<W, D>), dim3(blocks), dim3(256), 0, 0, buf, lines, iters, out);
  (void)hipEventRecord(b);
  (void)hipEventSynchronize(b);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, a, b);
  const double instr = (double)blocks * 4 * iters * 8;
  printf("%-28s %2d B/lane %s: %.2f ms, %.1f wave-loads/us/CU, %.0f cycles per wave-load per CU at 2.4 GHz\n", what, W, D ? "divergent" : "coalesced",
         ms, instr / (ms * 1e3) / 256, 2400.0 / (instr / (ms * 1e3) / 256));
}

int main() {
  unsigned char *buf; uint32_t *out;
  const uint64_t big = 1ull << 30, small = 2ull << 20;
  (void)hipMalloc(&buf, big); (void)hipMemset(buf, 1, big); (void)hipMalloc(&out, 4);
  run<4, true>(buf, big / 128, "1 GiB (HBM)", out);
  run<8, true>(buf, big / 128, "1 GiB (HBM)", out);
  run<16, true>(buf, big / 128, "1 GiB (HBM)", out);
  run<4, true>(buf, small / 128, "2 MiB (L2)", out);
  run<8, true>(buf, small / 128, "2 MiB (L2)", out);
  run<16, true>(buf, small / 128, "2 MiB (L2)", out);
  run<4, false>(buf, big / 128, "1 GiB (HBM)", out);
  run<16, false>(buf, big / 128, "1 GiB (HBM)", out);
  run<4, false>(buf, small / 128, "2 MiB (L2)", out);
  run<16, false>(buf, small / 128, "2 MiB (L2)", out);
  return 0;
}
