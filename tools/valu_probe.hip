// valu_probe.hip -- micro-benchmarks that decide the shape of the sweep loops:
//   * v_fma_f32 vs v_pk_fma_f32 issue rate at different occupancies
//   * sphere data from LDS (broadcast ds_read_b128) vs wave-uniform global loads (SMEM)
//   * 1 vs 2 rays per lane, scalar vs packed arithmetic
// Build: hipcc -O3 --offload-arch=gfx950 tools/valu_probe.hip -o tools/valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

typedef float float2v __attribute__((ext_vector_type(2)));

// ---- raw FMA rate ----------------------------------------------------------------
template <int CHAINS>
__global__ void __launch_bounds__(256) k_fma(float* out, int iters, float a, float b) {
  float x[CHAINS];
#pragma unroll
  for (int i = 0; i < CHAINS; ++i) x[i] = threadIdx.x * 0.001f + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) x[i] = __builtin_fmaf(x[i], a, b);
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < CHAINS; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int CHAINS>
__global__ void __launch_bounds__(256) k_pkfma(float* out, int iters, float a, float b) {
  float2v x[CHAINS];
  float2v va = {a, a * 1.0001f}, vb = {b, b * 0.999f};
#pragma unroll
  for (int i = 0; i < CHAINS; ++i) x[i] = float2v{threadIdx.x * 0.001f + i, threadIdx.x * 0.002f + i};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) x[i] = __builtin_elementwise_fma(x[i], va, vb);
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < CHAINS; ++i) s += x[i].x + x[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// ---- primary-ray filter loops ------------------------------------------------------
// sphere record: {Lx, Ly, Lz, CC}; test: (L.d)^2 > CC
#define CHUNK 1024
template <int RAYS>
__global__ void __launch_bounds__(256) k_primary_lds(const float4* __restrict__ sph, int n, float* out) {
  __shared__ float4 s[CHUNK];
  float dx[RAYS], dy[RAYS], dz[RAYS];
  unsigned cnt = 0;
#pragma unroll
  for (int r = 0; r < RAYS; ++r) {
    float u = (threadIdx.x + blockIdx.x * 256 + r * 7919) * 1e-6f;
    dx[r] = u; dy[r] = -0.5f + u; dz[r] = 0.8f;
  }
  for (int base = 0; base < n; base += CHUNK) {
    __syncthreads();
    for (int i = threadIdx.x; i < CHUNK; i += 256) s[i] = sph[base + i];
    __syncthreads();
#pragma unroll 4
    for (int j = 0; j < CHUNK; ++j) {
      float4 q = s[j];
      bool any = false;
#pragma unroll
      for (int r = 0; r < RAYS; ++r) {
        float t = q.x * dx[r];
        t = __builtin_fmaf(q.y, dy[r], t);
        t = __builtin_fmaf(q.z, dz[r], t);
        any |= (t * t > q.w);
      }
      if (__builtin_expect(__any(any), 0)) cnt += j;  // stand-in for the exact path
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = (float)cnt;
}
template <int RAYS>
__global__ void __launch_bounds__(256) k_primary_smem(const float4* __restrict__ sph, int n, float* out) {
  float dx[RAYS], dy[RAYS], dz[RAYS];
  unsigned cnt = 0;
#pragma unroll
  for (int r = 0; r < RAYS; ++r) {
    float u = (threadIdx.x + blockIdx.x * 256 + r * 7919) * 1e-6f;
    dx[r] = u; dy[r] = -0.5f + u; dz[r] = 0.8f;
  }
#pragma unroll 8
  for (int j = 0; j < n; ++j) {
    float4 q = sph[j];  // wave-uniform address -> scalar loads
    bool any = false;
#pragma unroll
    for (int r = 0; r < RAYS; ++r) {
      float t = q.x * dx[r];
      t = __builtin_fmaf(q.y, dy[r], t);
      t = __builtin_fmaf(q.z, dz[r], t);
      any |= (t * t > q.w);
    }
    if (__builtin_expect(__any(any), 0)) cnt += j;
  }
  out[blockIdx.x * 256 + threadIdx.x] = (float)cnt;
}
// two rays per lane, packed arithmetic
__global__ void __launch_bounds__(256) k_primary_lds_pk(const float4* __restrict__ sph, int n, float* out) {
  __shared__ float4 s[CHUNK];
  float u0 = (threadIdx.x + blockIdx.x * 256) * 1e-6f, u1 = u0 + 0.007919f;
  float2v dx = {u0, u1}, dy = {-0.5f + u0, -0.5f + u1}, dz = {0.8f, 0.8f};
  unsigned cnt = 0;
  for (int base = 0; base < n; base += CHUNK) {
    __syncthreads();
    for (int i = threadIdx.x; i < CHUNK; i += 256) s[i] = sph[base + i];
    __syncthreads();
#pragma unroll 4
    for (int j = 0; j < CHUNK; ++j) {
      float4 q = s[j];
      float2v qx = {q.x, q.x}, qy = {q.y, q.y}, qz = {q.z, q.z};
      float2v t = qx * dx;
      t = __builtin_elementwise_fma(qy, dy, t);
      t = __builtin_elementwise_fma(qz, dz, t);
      float2v t2 = t * t;
      bool any = (t2.x > q.w) | (t2.y > q.w);
      if (__builtin_expect(__any(any), 0)) cnt += j;
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = (float)cnt;
}

// ---- shadow filter loops: record {cx,cy,cz,R2}; L lights fused -----------------------
template <int L, bool PK>
__global__ void __launch_bounds__(256) k_shadow_lds(const float4* __restrict__ sph, int n, float* out) {
  __shared__ float4 s[CHUNK];
  float px = (threadIdx.x & 15) * 0.03f, py = 0.0001f, pz = (threadIdx.x >> 4) * 0.03f + blockIdx.x * 0.01f;
  float gx[L], gy[L], gz[L], rho[L];
#pragma unroll
  for (int k = 0; k < L; ++k) { gx[k] = 0.3f + 0.1f * k; gy[k] = 0.8f; gz[k] = -0.2f - 0.05f * k; rho[k] = -0.01f; }
  unsigned cnt = 0;
  const float kappa = 1.0f - 9.5e-7f;
  for (int base = 0; base < n; base += CHUNK) {
    __syncthreads();
    for (int i = threadIdx.x; i < CHUNK; i += 256) s[i] = sph[base + i];
    __syncthreads();
#pragma unroll 2
    for (int j = 0; j < CHUNK; ++j) {
      float4 q = s[j];
      float ux = px - q.x, uy = py - q.y, uz = pz - q.z;
      float uu = __builtin_fmaf(uz, uz, __builtin_fmaf(uy, uy, ux * ux));
      float cf = __builtin_fmaf(uu, kappa, -q.w);
      bool any = false;
      if (!PK) {
#pragma unroll
        for (int k = 0; k < L; ++k) {
          float h = __builtin_fmaf(uz, gz[k], __builtin_fmaf(uy, gy[k], __builtin_fmaf(ux, gx[k], q.w * rho[k])));
          any |= (h * h >= cf);
        }
      } else {
#pragma unroll
        for (int k = 0; k < L; k += 2) {
          float2v Gx = {gx[k], gx[k + 1]}, Gy = {gy[k], gy[k + 1]}, Gz = {gz[k], gz[k + 1]}, Rh = {rho[k], rho[k + 1]};
          float2v Ux = {ux, ux}, Uy = {uy, uy}, Uz = {uz, uz}, W = {q.w, q.w};
          float2v h = W * Rh;
          h = __builtin_elementwise_fma(Ux, Gx, h);
          h = __builtin_elementwise_fma(Uy, Gy, h);
          h = __builtin_elementwise_fma(Uz, Gz, h);
          float2v h2 = h * h;
          any |= (h2.x >= cf) | (h2.y >= cf);
        }
      }
      if (__builtin_expect(__any(any), 0)) cnt += j;
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = (float)cnt;
}


// ---- grouped variants: G spheres per early-out branch, loads issued up front ----------
template <int RAYS, int G>
__global__ void __launch_bounds__(256) k_primary_lds_grp(const float4* __restrict__ sph, int n, float* out) {
  __shared__ float4 s[CHUNK];
  float dx[RAYS], dy[RAYS], dz[RAYS];
  unsigned cnt = 0;
#pragma unroll
  for (int r = 0; r < RAYS; ++r) {
    float u = (threadIdx.x + blockIdx.x * 256 + r * 7919) * 1e-6f;
    dx[r] = u; dy[r] = -0.5f + u; dz[r] = 0.8f;
  }
  for (int base = 0; base < n; base += CHUNK) {
    __syncthreads();
    for (int i = threadIdx.x; i < CHUNK; i += 256) s[i] = sph[base + i];
    __syncthreads();
    for (int j = 0; j < CHUNK; j += G) {
      float4 q[G];
#pragma unroll
      for (int g = 0; g < G; ++g) q[g] = s[j + g];
      float m = -1.0f;
#pragma unroll
      for (int g = 0; g < G; ++g) {
#pragma unroll
        for (int r = 0; r < RAYS; ++r) {
          float t = q[g].x * dx[r];
          t = __builtin_fmaf(q[g].y, dy[r], t);
          t = __builtin_fmaf(q[g].z, dz[r], t);
          m = fmaxf(m, __builtin_fmaf(t, t, -q[g].w));   // disc; >= 0 means candidate
        }
      }
      if (__builtin_expect(__any(m >= 0.0f), 0)) cnt += j;
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = (float)cnt;
}
template <int L, int G>
__global__ void __launch_bounds__(256) k_shadow_lds_grp(const float4* __restrict__ sph, int n, float* out) {
  __shared__ float4 s[CHUNK];
  float px = (threadIdx.x & 15) * 0.03f, py = 0.0001f, pz = (threadIdx.x >> 4) * 0.03f + blockIdx.x * 0.01f;
  float gx[L], gy[L], gz[L], rho[L];
#pragma unroll
  for (int k = 0; k < L; ++k) { gx[k] = 0.3f + 0.1f * k; gy[k] = 0.8f; gz[k] = -0.2f - 0.05f * k; rho[k] = -0.01f; }
  unsigned cnt = 0;
  const float kappa = 1.0f - 9.5e-7f;
  for (int base = 0; base < n; base += CHUNK) {
    __syncthreads();
    for (int i = threadIdx.x; i < CHUNK; i += 256) s[i] = sph[base + i];
    __syncthreads();
    for (int j = 0; j < CHUNK; j += G) {
      float4 q[G];
#pragma unroll
      for (int g = 0; g < G; ++g) q[g] = s[j + g];
      float m = -1.0f;
#pragma unroll
      for (int g = 0; g < G; ++g) {
        float ux = px - q[g].x, uy = py - q[g].y, uz = pz - q[g].z;
        float uu = __builtin_fmaf(uz, uz, __builtin_fmaf(uy, uy, ux * ux));
        float cf = __builtin_fmaf(uu, kappa, -q[g].w);
#pragma unroll
        for (int k = 0; k < L; ++k) {
          float h = __builtin_fmaf(uz, gz[k], __builtin_fmaf(uy, gy[k], __builtin_fmaf(ux, gx[k], q[g].w * rho[k])));
          m = fmaxf(m, __builtin_fmaf(h, h, -cf));
        }
      }
      if (__builtin_expect(__any(m >= 0.0f), 0)) cnt += j;
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = (float)cnt;
}

template <typename F>
static float time_ms(F launch, int reps = 5) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  launch();
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int i = 0; i < reps; ++i) {
    CK(hipEventRecord(a));
    launch();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (ms < best) best = ms;
  }
  return best;
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device %s arch %s CUs %d clock %d kHz\n", p.name, p.gcnArchName, p.multiProcessorCount, p.clockRate);
  const int cus = p.multiProcessorCount;
  float* out; CK(hipMalloc(&out, sizeof(float) * 256 * cus * 16));
  const int N = 10240;
  std::vector<float4> h(N);
  for (int i = 0; i < N; ++i) h[i] = make_float4(100.f + (i % 100) * 1.5f, -50.f, 50.f + (i / 100) * 1.5f, 3e4f);
  float4* sph; CK(hipMalloc(&sph, sizeof(float4) * N));
  CK(hipMemcpy(sph, h.data(), sizeof(float4) * N, hipMemcpyHostToDevice));

  // raw fma rate: blocks per CU = occupancy knob (each block 256 thr = 1 wave/SIMD)
  for (int bpc : {1, 2, 4, 8}) {
    int iters = 20000;
    float ms = time_ms([&] { k_fma<8><<<cus * bpc, 256>>>(out, iters, 1.0001f, 0.5f); });
    double fl = 2.0 * 8 * iters * 256.0 * cus * bpc;
    printf("fma   chains=8 blocks/CU=%d : %.3f ms  %.1f TFLOP/s\n", bpc, ms, fl / ms * 1e-9);
    ms = time_ms([&] { k_pkfma<8><<<cus * bpc, 256>>>(out, iters, 1.0001f, 0.5f); });
    printf("pkfma chains=8 blocks/CU=%d : %.3f ms  %.1f TFLOP/s\n", bpc, ms, 2 * fl / ms * 1e-9);
  }
  for (int bpc : {2, 4, 8}) {
    int blocks = cus * bpc;
    auto report = [&](const char* name, float ms, int rays) {
      double tests = (double)N * 256.0 * blocks * rays;
      printf("%-28s blocks/CU=%d : %.3f ms  %.2f Gtests/s (%.1f Mrays/s @10k)\n", name, bpc, ms, tests / ms * 1e-6, tests / ms * 1e-3 / 10000.0);
    };
    report("primary lds  1ray", time_ms([&] { k_primary_lds<1><<<blocks, 256>>>(sph, N, out); }), 1);
    report("primary lds  2ray", time_ms([&] { k_primary_lds<2><<<blocks, 256>>>(sph, N, out); }), 2);
    report("primary lds  4ray", time_ms([&] { k_primary_lds<4><<<blocks, 256>>>(sph, N, out); }), 4);
    report("primary lds  2ray packed", time_ms([&] { k_primary_lds_pk<<<blocks, 256>>>(sph, N, out); }), 2);
    report("primary smem 1ray", time_ms([&] { k_primary_smem<1><<<blocks, 256>>>(sph, N, out); }), 1);
    report("primary smem 2ray", time_ms([&] { k_primary_smem<2><<<blocks, 256>>>(sph, N, out); }), 2);
    report("shadow lds L=4 scalar", time_ms([&] { k_shadow_lds<4, false><<<blocks, 256>>>(sph, N, out); }), 4);
    report("shadow lds L=4 packed", time_ms([&] { k_shadow_lds<4, true><<<blocks, 256>>>(sph, N, out); }), 4);
    report("shadow lds L=8 scalar", time_ms([&] { k_shadow_lds<8, false><<<blocks, 256>>>(sph, N, out); }), 8);
    report("primary lds 1ray grp4", time_ms([&] { k_primary_lds_grp<1, 4><<<blocks, 256>>>(sph, N, out); }), 1);
    report("primary lds 1ray grp8", time_ms([&] { k_primary_lds_grp<1, 8><<<blocks, 256>>>(sph, N, out); }), 1);
    report("primary lds 2ray grp4", time_ms([&] { k_primary_lds_grp<2, 4><<<blocks, 256>>>(sph, N, out); }), 2);
    report("primary lds 2ray grp8", time_ms([&] { k_primary_lds_grp<2, 8><<<blocks, 256>>>(sph, N, out); }), 2);
    report("shadow lds L=4 grp4", time_ms([&] { k_shadow_lds_grp<4, 4><<<blocks, 256>>>(sph, N, out); }), 4);
    report("shadow lds L=4 grp8", time_ms([&] { k_shadow_lds_grp<4, 8><<<blocks, 256>>>(sph, N, out); }), 4);
    report("shadow lds L=8 grp4", time_ms([&] { k_shadow_lds_grp<8, 4><<<blocks, 256>>>(sph, N, out); }), 8);
    report("shadow lds L=8 packed", time_ms([&] { k_shadow_lds<8, true><<<blocks, 256>>>(sph, N, out); }), 8);
  }
  return 0;
}
