// Compares the kernels' integer-exponent power of the Phong highlight (rtk::highlight_power) with the device
// library's pow() over random cosines and every exponent from 1 to 4095: never more than an ulp apart (the
// library's own accuracy; tests/test_powi_model.py shows the kernels' value to be the correctly rounded one).
// Build and run on the GPU box:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iraytracer-rust_amd/csrc tools/power_probe.hip -o tools/power_probe && tools/power_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "rtk_shading.h"

__device__ __forceinline__ uint64_t mix(uint64_t z) {
  z += 0x9e3779b97f4a7c15ull;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}

// result[0]: unused; result[1]: powers more than 1 ulp from pow(); result[2]: powers that differ at all
__global__ void probe(uint64_t per_thread, unsigned long long* result) {
  const uint64_t id = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long bad_root = 0, far_pow = 0, diff_pow = 0;
  for (uint64_t i = 0; i < per_thread; ++i) {
    const uint64_t bits = mix(id * per_thread + i);
    // highlight cosines in (0, 1], exponents 1 .. 4095
    const double cosv = 1.0 - __builtin_bit_cast(double, (bits & 0x000fffffffffffffull) | ((uint64_t)(1023 - 1 - ((bits >> 52) & 31u)) << 52));
    const double n = (double)(1u + (uint32_t)((bits >> 40) & 0xfffu) % 4095u);
    const double p = rtk::highlight_power(cosv, n), q = pow(cosv, n);
    if (q > 0x1p-900) {
      const double ulp = fabs(q) * 0x1p-52;
      far_pow += fabs(p - q) > ulp ? 1 : 0;
      diff_pow += p != q ? 1 : 0;
    }
  }
  atomicAdd(&result[0], bad_root);
  atomicAdd(&result[1], far_pow);
  atomicAdd(&result[2], diff_pow);
}

int main() {
  unsigned long long* d = nullptr;
  if (hipMalloc((void**)&d, 3 * sizeof(unsigned long long)) != hipSuccess) { std::printf("no device\n"); return 2; }
  (void)hipMemset(d, 0, 3 * sizeof(unsigned long long));
  const uint64_t per_thread = 4096, blocks = 4096, threads = 256;
  hipLaunchKernelGGL(probe, dim3(blocks), dim3(threads), 0, 0, per_thread, d);
  unsigned long long h[3] = {~0ull, ~0ull, ~0ull};
  if (hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) { std::printf("kernel failed\n"); return 2; }
  const double total = (double)per_thread * blocks * threads;
  std::printf("%.3g arguments: powers beyond 1 ulp of pow(): %llu; powers differing from pow() at all: %llu (%.2f %%)\n",
              total, h[1], h[2], 100.0 * (double)h[2] / total);
  return h[0] == 0 && h[1] == 0 ? 0 : 1;
}
