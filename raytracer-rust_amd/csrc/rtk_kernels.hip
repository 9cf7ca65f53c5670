// rtk_kernels.hip -- the single-kernel path (scenes with reflective or transparent
// materials, caller-supplied rays), the per-camera record kernel and the de-interleave
// kernel of the multi-rank path.  Scenes with opaque materials only run rtk_opaque.hip.
//
// One thread traces one pixel (reference loop: src/scene/cameras.rs:72-85); one
// 256-thread workgroup owns a 32x8 pixel tile, its four wavefronts own 8x8 sub-tiles so
// that the 64 camera rays of a wavefront stay coherent.
//
// Every "ray" is one `Scene::intersect` call of the reference (src/scene.rs:156-169): which
// objects does it hit, and where.
//   1. planes are few: each is intersected exactly in f64;
//   2. spheres are walked cluster by cluster (rtk_sweeps.h): a conservative FP32 test per
//      (ray, cluster ball), then per (ray, sphere) -- "can the ray's line touch the bounding
//      ball at all?" with enough slack to cover every rounding error, so no real hit is
//      ever rejected;
//   3. kept candidates are re-evaluated in f64 with the reference's own formulas
//      (object-space ray, unit sphere quadratic), so every decision -- closest hit,
//      tie-break, shadow -- is taken on f64 values like the reference takes it.
// Shading (LightingData, Phong, reflection, refraction, Schlick) is O(1) per ray and
// runs in f64 in registers; the result is narrowed to FP32 / RGBA8 at the store.
//
// No MFMA (nothing here is a dense contraction); HBM traffic is the framebuffer store; the
// scene records stay in L2.

#include "rtk_trace.h"
#include "rtk_kernels.h"

#include <algorithm>

#pragma clang fp contract(off)

namespace rtk {

// ------------------------------------------------------------------------------------
// `Camera::render` (src/scene/cameras.rs:72-85) for scenes with reflective or transparent
// materials: ray generation (cameras.rs:52-69 with the matrix inverse hoisted to the host), the
// ray tree of every pixel, and the framebuffer store (src/graphics.rs:44-46, 59-74).
//
// Ray trees differ wildly from pixel to pixel (1 to 62 rays with the reference's depth of 5), so a
// wavefront that walked the trees of 64 fixed pixels in lockstep would idle most lanes most of the
// time.  Here the lanes are persistent: a lane whose pixel is finished stores it and takes the next
// pixel of the wavefront's current 8x8 patch; when a patch is used up the wavefront draws the next
// one from a counter shared by the whole launch.  A pixel is traced from its camera ray to its last
// leaf by ONE lane, in the same order of operations whichever lane that is, so the frame does not
// depend on how the pixels were dealt.  Patches are numbered from the bottom of the frame (the
// expensive tiles) upwards.
struct PixelSlot {
  uint32_t x, y;      // frame coordinates
  uint32_t k, lrow;   // local tile, local tile row (band layouts)
  uint32_t lx, ly;    // position inside the tile
};
__device__ __forceinline__ PixelSlot pixel_slot(const RenderTarget& T, uint32_t patch, uint32_t slot) {
  PixelSlot p;
  p.k = T.n_local_tiles - 1u - (patch >> 2);
  const TileRef tr = tile_ref(T, p.k);
  p.lrow = tr.lrow;
  p.lx = ((patch & 3u) << 3) | (slot & 7u);
  p.ly = slot >> 3;
  p.x = tr.tx * 32u + p.lx;
  p.y = tr.ty * 8u + p.ly;
  return p;
}
// One pixel to the outputs (row-major frame, the rank's tile-major buffer, or its own tile rows):
// the narrowing to FP32 and the 8-bit cast of src/graphics.rs:59-74.
__device__ __forceinline__ void store_pixel(const DevCamera& C, const RenderTarget& T, const PixelSlot& p, V3 color) {
  size_t at;
  if (T.layout == 1u)
    at = (size_t)p.k * 256u + p.ly * 32u + p.lx;
  else
    at = (size_t)p.x + (size_t)(T.layout == 2u ? p.lrow * 8u + p.ly : p.y) * C.width;
  if (T.rgb) {
    T.rgb[at * 3u + 0u] = (float)color.x;
    T.rgb[at * 3u + 1u] = (float)color.y;
    T.rgb[at * 3u + 2u] = (float)color.z;
  }
  if (T.rgba8)
    reinterpret_cast<uint32_t*>(T.rgba8)[at] = cast_u8(color.x) | (cast_u8(color.y) << 8) | (cast_u8(color.z) << 16) | 0xff000000u;
}

// Scenes of a handful of objects (the reference's bundled scene): a node costs little more than its
// shading, and lanes that walk the trees of neighbouring pixels in lockstep take the same branches
// of it, which is worth more there than keeping every lane busy (bundled scene at 1080p: 0.52 ms
// against 0.74 ms with persistent lanes).  One lane per pixel of a 32x8 tile, as in the opaque path.
template <bool TRACK>
__global__ void __launch_bounds__(64, 4) render_small_kernel(DevScene S, DevCamera C, RenderTarget T, DevCounters* counters) {
  // one wavefront per workgroup (patch_of_group): ray trees differ from patch to patch
  const PatchOfGroup pg = patch_of_group();
  if (pg.tile >= T.n_local_tiles) return;
  // expensive tiles (ground, bottom of the frame) first, cheap sky tiles last
  const uint32_t k = T.n_local_tiles - 1u - pg.tile;
  const TileRef tr = tile_ref(T, k);
  const uint32_t wave = pg.quarter, lane = threadIdx.x & 63u;
  const uint32_t lx = (wave << 3) | (lane & 7u), ly = lane >> 3;
  const uint32_t x = tr.tx * 32u + lx, y = tr.ty * 8u + ly;
  const bool valid = x < C.width && y < C.height;
  const V3 origin = ld3(C.origin);
  const V3 dir = camera_ray_direction(C, x, y);  // cameras.rs:52-69
  WaveCount wc;
  V3 color = trace_ray<true, false, 1, false, true, TRACK>(S, nullptr, nullptr, valid, origin, dir, wc);
  store_patch(C, T, k, pg.quarter, color);
  flush_counts(counters + (pg.tile % kCounterShards), wc);
}

template <int LB, bool WIDE, bool TRACK>
__global__ void __launch_bounds__(kBlock, 4) render_kernel(DevScene S, DevCamera C, RenderTarget T, DevCounters* counters,
                                                        unsigned int* work, const unsigned int* only_if) {
  // the second try of a launch whose first (level by level) ran out of queue room: nothing to do otherwise
  if (only_if && *only_if == 0u) return;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t n_patches = T.n_local_tiles * 4u;
  const V3 origin = ld3(C.origin);

  StackEntry stack[kMaxDepth + 1];
  int sp = 0;
  V3 color = {0, 0, 0};
  bool busy = false;                      // this lane holds a pixel
  uint32_t my_patch = 0, my_slot = 0;     // ... this one
  uint32_t cur_patch = 0, cur_next = 64;  // the wavefront's current patch and its next unassigned pixel
  bool exhausted = false;                 // the launch has no patches left
  WaveCount wc;

  for (;;) {
    // pixels for the idle lanes
    while (!exhausted) {
      const unsigned long long idle = __builtin_amdgcn_ballot_w64(!busy);
      if (idle == 0ull) break;
      if (cur_next >= 64u) {
        unsigned int p = 0;
        if (lane == 0) p = atomicAdd(work, 1u);
        p = (unsigned int)__builtin_amdgcn_readfirstlane((int)p);
        if (p >= n_patches) {
          exhausted = true;
          break;
        }
        cur_patch = p;
        cur_next = 0;
      }
      const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
      const uint32_t take = min((uint32_t)__builtin_popcountll(idle), 64u - cur_next);
      if (!busy && rank < take) {
        const PixelSlot p = pixel_slot(T, cur_patch, cur_next + rank);
        if (p.x < C.width && p.y < C.height) {  // a slot outside the frame is used up without a pixel
          busy = true;
          my_patch = cur_patch;
          my_slot = cur_next + rank;
          color = {0, 0, 0};
          sp = 0;
          if (S.max_depth == 0) {
            color = ld3(S.ambient);  // trace_inner(depth 0 >= MAX_DEPTH)
          } else {
            const V3 dir = camera_ray_direction(C, p.x, p.y);  // cameras.rs:52-69
            StackEntry& e = stack[sp++];
            e.ox = origin.x, e.oy = origin.y, e.oz = origin.z;
            e.dx = dir.x, e.dy = dir.y, e.dz = dir.z;
            e.weight = 1.0;
            e.depth = 0;
          }
        }
      }
      cur_next += take;
    }
    if (!__any(busy)) break;  // nothing left to hand out and every pixel stored

    // one node per lane
    const bool have = busy && sp > 0;
    V3 ro = {0, 0, 0}, rd = {0, 0, 1};
    double weight = 0.0;
    int depth = 0;
    if (have) {
      --sp;
      const StackEntry& e = stack[sp];
      ro = {e.ox, e.oy, e.oz};
      rd = {e.dx, e.dy, e.dz};
      weight = e.weight;
      depth = e.depth;
    }
    if (__any(have)) {
      StackSink sink{stack, sp, color};
      trace_step<true, false, LB, WIDE, false, TRACK>(S, nullptr, nullptr, have, false, ro, rd, weight, depth, sink, wc);
    }

    // pixels whose tree is done
    if (busy && sp == 0) {
      store_pixel(C, T, pixel_slot(T, my_patch, my_slot), color);
      busy = false;
    }
  }

  // ray counters: one set of atomics per wavefront
  flush_counts(counter_shard(counters), wc);
}

// `Scene::trace` for caller-supplied rays (src/scene.rs:136-138): n x 4 doubles each.
template <int LB, bool WIDE, bool TRACK>
__global__ void __launch_bounds__(kBlock) trace_rays_kernel(DevScene S, const double* __restrict__ origins,
                                                            const double* __restrict__ directions, uint64_t n,
                                                            double* __restrict__ colors, DevCounters* counters) {
  const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  const bool valid = i < n;
  V3 o = {0, 0, 0}, d = {0, 0, 1};
  if (valid) {
    o = ld3(origins + 4 * i);
    d = ld3(directions + 4 * i);
  }
  WaveCount wc;
  V3 color = trace_ray<true, false, LB, WIDE, false, TRACK>(S, nullptr, nullptr, valid, o, d, wc);
  if (valid) {
    colors[3 * i + 0] = color.x;
    colors[3 * i + 1] = color.y;
    colors[3 * i + 2] = color.z;
  }
  flush_counts(counter_shard(counters), wc);
}

// Per-camera records of the camera-ray tests: one thread per (padded) sphere.
__global__ void __launch_bounds__(kBlock) prep_primary_kernel(DevScene S, DevCamera C, float4* __restrict__ prim,
                                                              float4* __restrict__ prim_ball,
                                                              float4* __restrict__ prim_cluster,
                                                              unsigned long long* __restrict__ zero, uint32_t n_zero) {
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  // the frame's ray counters start at zero: cleared here rather than by a fill kernel of their own
  for (uint32_t z = i; z < n_zero; z += gridDim.x * kBlock) zero[z] = 0ull;
  const double eps = (double)kEps32;
  if (prim_cluster && i < S.n_clusters_padded) {
    // cluster ball as seen from the camera; the ball already contains its members' balls
    const float4 cb = S.cluster_ball[i];
    const double Lx = (double)cb.x + S.center[0] - C.origin[0], Ly = (double)cb.y + S.center[1] - C.origin[1],
                 Lz = (double)cb.z + S.center[2] - C.origin[2];
    const double Rs = (double)cb.w * (1.0 + 1e-6) + 16.0 * eps * sqrt(Lx * Lx + Ly * Ly + Lz * Lz);
    float rsf = (float)Rs;
    if ((double)rsf < Rs) rsf = nextafterf(rsf, INFINITY);
    prim_cluster[i] = cb.w >= 0.0f ? make_float4((float)Lx, (float)Ly, (float)Lz, rsf) : make_float4(0.f, 0.f, 0.f, -1.0f);
  }
  if (i >= S.n_spheres_padded) return;
  // padding of the per-ray records repeats the last real sphere (see rtk_scene_upload);
  // padding of the wavefront-level records can never pass (negative radius)
  const double4 s = S.sph_exact[min(i, S.n_spheres - 1u)];
  const double Lx = s.x - C.origin[0], Ly = s.y - C.origin[1], Lz = s.z - C.origin[2];
  const double LL = Lx * Lx + Ly * Ly + Lz * Lz;
  // rounding L to FP32 moves the centre by at most eps|L|: grow the ball by that much
  const double R = s.w * (1.0 + 1e-12) + 1.01 * eps * sqrt(LL);
  // 12 eps |L|^2 covers the dot product (3 roundings), the rounded direction and this record's own rounding
  double cc = LL * (1.0 - 12.0 * eps) - R * R;
  float ccf = (float)cc;
  if ((double)ccf > cc) ccf = nextafterf(ccf, -INFINITY);
  ccf = nextafterf(ccf, -INFINITY);
  prim[i] = make_float4((float)Lx, (float)Ly, (float)Lz, ccf);
  if (prim_ball) {
    // cone test: 16 eps |L| covers the rounding of L, of the axis and of the distance arithmetic
    const double Rs = s.w * (1.0 + 1e-6) + 16.0 * eps * sqrt(LL);
    float rsf = (float)Rs;
    if ((double)rsf < Rs) rsf = nextafterf(rsf, INFINITY);
    prim_ball[i] = i < S.n_spheres ? make_float4((float)Lx, (float)Ly, (float)Lz, rsf) : make_float4(0.f, 0.f, 0.f, -1.0f);
  }
}

// Rebuilds row-major frames from `world` compact tile buffers: rank r's FP32 tiles start
// rgb_stride floats after rank r-1's, its RGBA8 tiles px_stride pixels after rank r-1's.
__global__ void __launch_bounds__(kBlock) untile_kernel(uint32_t width, uint32_t height, uint32_t tiles_x, uint32_t n_tiles,
                                                        uint32_t world, uint64_t rgb_stride, uint64_t px_stride,
                                                        const float* __restrict__ rgb_c, float* __restrict__ rgb_f,
                                                        const uint32_t* __restrict__ px_c, uint32_t* __restrict__ px_f) {
  const uint32_t tile = blockIdx.x;
  if (tile >= n_tiles) return;
  const uint32_t rank = tile % world, k = tile / world;
  const uint32_t tx = tile % tiles_x, ty = tile / tiles_x;
  const uint32_t rx = threadIdx.x & 31u, ry = threadIdx.x >> 5;
  const uint32_t gx = tx * 32u + rx, gy = ty * 8u + ry;
  if (px_c && gx < width && gy < height)
    px_f[(size_t)gx + (size_t)gy * width] = px_c[(size_t)rank * px_stride + (size_t)k * 256u + threadIdx.x];
  if (rgb_c) {
    const float* src = rgb_c + (size_t)rank * rgb_stride + (size_t)k * 768u;
#pragma unroll
    for (uint32_t pass = 0; pass < 3; ++pass) {
      const uint32_t e = threadIdx.x + pass * 256u;
      const uint32_t row = e / 96u, col = e % 96u;
      const uint32_t px = tx * 32u + col / 3u, py = ty * 8u + row;
      if (px < width && py < height) rgb_f[((size_t)px + (size_t)py * width) * 3u + col % 3u] = src[e];
    }
  }
}

// Box filter of a supersampled frame: pixel (x, y) is the mean of the s x s samples
// (s x + i, s y + j) of a frame of s*width x s*height FP32 RGB samples, summed in f64 in
// row-major order (j outer, i inner), divided by s*s, then narrowed to FP32 and cast to 8 bits
// as a rendered colour is (src/graphics.rs:59-74).  One thread per pixel.
__global__ void __launch_bounds__(kBlock) resolve_kernel(uint32_t width, uint32_t height, uint32_t s,
                                                         const float* __restrict__ samples, float* __restrict__ rgb,
                                                         uint32_t* __restrict__ rgba8) {
  const uint64_t p = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (p >= (uint64_t)width * height) return;
  const uint32_t x = (uint32_t)(p % width), y = (uint32_t)(p / width);
  const size_t row = (size_t)width * s * 3u;
  const float* src = samples + (size_t)y * s * row + (size_t)x * s * 3u;
  double r = 0.0, g = 0.0, b = 0.0;
  for (uint32_t j = 0; j < s; ++j) {
    const float* line = src + (size_t)j * row;
    for (uint32_t i = 0; i < s; ++i) {
      r += (double)line[3u * i + 0u];
      g += (double)line[3u * i + 1u];
      b += (double)line[3u * i + 2u];
    }
  }
  const double n = (double)(s * s);
  r /= n, g /= n, b /= n;
  if (rgb) {
    rgb[p * 3u + 0u] = (float)r;
    rgb[p * 3u + 1u] = (float)g;
    rgb[p * 3u + 2u] = (float)b;
  }
  if (rgba8) rgba8[p] = cast_u8(r) | (cast_u8(g) << 8) | (cast_u8(b) << 16) | 0xff000000u;
}

// The same average on a rank's compact tile buffer (s divides 8): each 32x8 tile of samples
// becomes a (32/s) x (8/s) tile of pixels, row-major inside the tile, tiles in the same order.
__global__ void __launch_bounds__(kBlock) resolve_tiles_kernel(uint64_t n_pixels, uint32_t s, const float* __restrict__ samples,
                                                               float* __restrict__ rgb, uint32_t* __restrict__ rgba8) {
  const uint64_t q = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (q >= n_pixels) return;
  const uint32_t tw = 32u / s, per_tile = 256u / (s * s);
  const uint64_t k = q / per_tile;
  const uint32_t r = (uint32_t)(q % per_tile), lx = r % tw, ly = r / tw;
  const float* src = samples + (k * 256u + (uint64_t)(ly * s) * 32u + lx * s) * 3u;
  double cr = 0.0, cg = 0.0, cb = 0.0;
  for (uint32_t j = 0; j < s; ++j)
    for (uint32_t i = 0; i < s; ++i) {
      const float* e = src + (j * 32u + i) * 3u;
      cr += (double)e[0];
      cg += (double)e[1];
      cb += (double)e[2];
    }
  const double n = (double)(s * s);
  cr /= n, cg /= n, cb /= n;
  if (rgb) {
    rgb[q * 3u + 0u] = (float)cr;
    rgb[q * 3u + 1u] = (float)cg;
    rgb[q * 3u + 2u] = (float)cb;
  }
  if (rgba8) rgba8[q] = cast_u8(cr) | (cast_u8(cg) << 8) | (cast_u8(cb) << 16) | 0xff000000u;
}

// De-interleave of such buffers: the frame's tiles are (32/s) x (8/s) pixels, numbered like the
// 32x8 tiles of the sample frame they came from.  One thread per pixel of the frame.
__global__ void __launch_bounds__(kBlock) untile_sampled_kernel(uint32_t width, uint32_t height, uint32_t s, uint32_t tiles_x,
                                                                uint32_t world, uint64_t rgb_stride, uint64_t px_stride,
                                                                const float* __restrict__ rgb_c, float* __restrict__ rgb_f,
                                                                const uint32_t* __restrict__ px_c, uint32_t* __restrict__ px_f) {
  const uint64_t p = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (p >= (uint64_t)width * height) return;
  const uint32_t x = (uint32_t)(p % width), y = (uint32_t)(p / width);
  const uint32_t tw = 32u / s, th = 8u / s, per_tile = tw * th;
  const uint32_t tile = (y / th) * tiles_x + x / tw;
  const uint32_t rank = tile % world, k = tile / world;
  const uint64_t q = (uint64_t)k * per_tile + (y % th) * tw + x % tw;
  if (px_c) px_f[p] = px_c[(uint64_t)rank * px_stride + q];
  if (rgb_c) {
    const float* src = rgb_c + (uint64_t)rank * rgb_stride + q * 3u;
    rgb_f[p * 3u + 0u] = src[0];
    rgb_f[p * 3u + 1u] = src[1];
    rgb_f[p * 3u + 2u] = src[2];
  }
}

// ------------------------------------------------------------------------------------
// launch wrappers (called from rtk_api.hip)

// WIDE kernels carry the sphere-per-lane sweep of opened clusters; scenes below kWideScene
// spheres never take it.
constexpr uint32_t kWideScene = 256;

template <bool WIDE, bool TRACK>
static hipError_t launch_render_lb(int lb, dim3 grid, hipStream_t st, const DevScene& S, const DevCamera& C, const RenderTarget& T,
                                   DevCounters* counters, unsigned int* work, const unsigned int* only_if) {
  switch (lb) {
    case 1: hipLaunchKernelGGL((render_kernel<1, WIDE, TRACK>), grid, dim3(kBlock), 0, st, S, C, T, counters, work, only_if); break;
    case 2: hipLaunchKernelGGL((render_kernel<2, WIDE, TRACK>), grid, dim3(kBlock), 0, st, S, C, T, counters, work, only_if); break;
    case 4: hipLaunchKernelGGL((render_kernel<4, WIDE, TRACK>), grid, dim3(kBlock), 0, st, S, C, T, counters, work, only_if); break;
    default: hipLaunchKernelGGL((render_kernel<8, WIDE, TRACK>), grid, dim3(kBlock), 0, st, S, C, T, counters, work, only_if); break;
  }
  return hipGetLastError();
}

int light_batch_for(uint32_t n_lights) {
  if (n_lights <= 1) return 1;
  if (n_lights == 2) return 2;
  if (n_lights <= 4) return 4;
  return 8;
}

hipError_t launch_prep_primary(hipStream_t st, const DevScene& S, const DevCamera& C, float4* prim, float4* prim_ball,
                               float4* prim_cluster, DevCounters* counters) {
  drop_stale_error();
  constexpr size_t kCounterBytes = sizeof(DevCounters) * kCounterSlots;
  static_assert(kCounterBytes % sizeof(unsigned long long) == 0, "counters are cleared as 64-bit words");
  if (S.n_spheres_padded == 0) return hipMemsetAsync(counters, 0, kCounterBytes, st);
  // n_clusters_padded <= n_spheres_padded: one grid covers both record arrays
  dim3 grid((S.n_spheres_padded + kBlock - 1) / kBlock);
  hipLaunchKernelGGL(prep_primary_kernel, grid, dim3(kBlock), 0, st, S, C, prim, prim_ball, prim_cluster,
                     reinterpret_cast<unsigned long long*>(counters), (uint32_t)(kCounterBytes / sizeof(unsigned long long)));
  return hipGetLastError();
}

hipError_t launch_render(hipStream_t st, const DevScene& S, const DevCamera& C, const RenderTarget& T, DevCounters* counters,
                         const WavefrontBuffers* wf, const float4* prim, const float4* prim_ball, const float4* prim_cluster) {
  drop_stale_error();
  if (T.n_local_tiles == 0) return hipSuccess;
  // TRACK kernels (some refractive index differs from the others) carry the entry before the winner
  // and the literal n1 evaluation; the others drop both
  const bool track = S.track_n1 != 0;
  if (S.n_objects <= kBruteObjects) {
    if (track)
      hipLaunchKernelGGL((render_small_kernel<true>), dim3(patch_groups(T.n_local_tiles)), dim3(64), 0, st, S, C, T, counters);
    else
      hipLaunchKernelGGL((render_small_kernel<false>), dim3(patch_groups(T.n_local_tiles)), dim3(64), 0, st, S, C, T, counters);
    return hipGetLastError();
  }
  // Level by level where the scene has secondary rays to queue; the depth-first kernel below then
  // only runs (behind the flag) for a launch whose children did not fit the queue.
  const unsigned int* only_if = nullptr;  // (scenes of a handful of objects, above, are never queued)
  if (wf && wf->cap_nodes >= kBlock * wf->nodes_per_pixel && prim && prim_ball && prim_cluster && S.any_secondary && S.max_depth >= 2) {
    const hipError_t we = launch_render_wavefront(st, S, C, prim, prim_ball, prim_cluster, T, *wf, counters);
    if (we != hipSuccess) return we;
    only_if = wavefront_overflow_flag(*wf);
  }
  // Persistent wavefronts draw 8x8 patches from the launch's work counter (the word behind the
  // counter shards): enough workgroups to fill the GPU
  // twice over at four wavefronts per SIMD, no more than there are tiles.
  unsigned int* work = reinterpret_cast<unsigned int*>(counters + kCounterShards);
  // a frame may be made of several launches that share the counters: each starts its patches at zero
  hipError_t e = hipMemsetAsync(work, 0, sizeof(unsigned int), st);
  if (e != hipSuccess) return e;
  dim3 grid(std::min<uint32_t>(T.n_local_tiles, kPersistentGroups));
  const int lb = light_batch_for(S.n_lights);
  if (S.n_spheres >= kWideScene)
    return track ? launch_render_lb<true, true>(lb, grid, st, S, C, T, counters, work, only_if)
                 : launch_render_lb<true, false>(lb, grid, st, S, C, T, counters, work, only_if);
  return track ? launch_render_lb<false, true>(lb, grid, st, S, C, T, counters, work, only_if)
               : launch_render_lb<false, false>(lb, grid, st, S, C, T, counters, work, only_if);
}

template <bool WIDE, bool TRACK>
static hipError_t launch_trace_rays_lb(int lb, dim3 grid, hipStream_t st, const DevScene& S, const double* origins,
                                       const double* directions, uint64_t n, double* colors, DevCounters* counters) {
  switch (lb) {
    case 1: hipLaunchKernelGGL((trace_rays_kernel<1, WIDE, TRACK>), grid, dim3(kBlock), 0, st, S, origins, directions, n, colors, counters); break;
    case 2: hipLaunchKernelGGL((trace_rays_kernel<2, WIDE, TRACK>), grid, dim3(kBlock), 0, st, S, origins, directions, n, colors, counters); break;
    case 4: hipLaunchKernelGGL((trace_rays_kernel<4, WIDE, TRACK>), grid, dim3(kBlock), 0, st, S, origins, directions, n, colors, counters); break;
    default: hipLaunchKernelGGL((trace_rays_kernel<8, WIDE, TRACK>), grid, dim3(kBlock), 0, st, S, origins, directions, n, colors, counters); break;
  }
  return hipGetLastError();
}

hipError_t launch_trace_rays(hipStream_t st, const DevScene& S, const double* origins, const double* directions, uint64_t n,
                             double* colors, DevCounters* counters) {
  drop_stale_error();
  if (n == 0) return hipSuccess;
  dim3 grid((unsigned)((n + kBlock - 1) / kBlock));
  const int lb = light_batch_for(S.n_lights);
  const bool track = S.track_n1 != 0;
  if (S.n_spheres >= kWideScene)
    return track ? launch_trace_rays_lb<true, true>(lb, grid, st, S, origins, directions, n, colors, counters)
                 : launch_trace_rays_lb<true, false>(lb, grid, st, S, origins, directions, n, colors, counters);
  return track ? launch_trace_rays_lb<false, true>(lb, grid, st, S, origins, directions, n, colors, counters)
               : launch_trace_rays_lb<false, false>(lb, grid, st, S, origins, directions, n, colors, counters);
}
hipError_t launch_untile(hipStream_t st, uint32_t width, uint32_t height, uint32_t world, uint64_t rgb_stride_floats,
                         uint64_t px_stride_pixels, const float* rgb_c, float* rgb_f, const uint8_t* px_c, uint8_t* px_f) {
  drop_stale_error();
  const uint32_t tiles_x = (width + 31u) / 32u, tiles_y = (height + 7u) / 8u;
  const uint32_t n_tiles = tiles_x * tiles_y;
  if (n_tiles == 0) return hipSuccess;
  hipLaunchKernelGGL(untile_kernel, dim3(n_tiles), dim3(kBlock), 0, st, width, height, tiles_x, n_tiles, world,
                     rgb_stride_floats, px_stride_pixels, rgb_c, rgb_f, reinterpret_cast<const uint32_t*>(px_c),
                     reinterpret_cast<uint32_t*>(px_f));
  return hipGetLastError();
}

hipError_t launch_resolve(hipStream_t st, uint32_t width, uint32_t height, uint32_t s, const float* samples, float* rgb,
                          uint8_t* rgba8) {
  drop_stale_error();
  const uint64_t pixels = (uint64_t)width * height;
  if (pixels == 0) return hipSuccess;
  hipLaunchKernelGGL(resolve_kernel, dim3((unsigned)((pixels + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, width, height, s,
                     samples, rgb, reinterpret_cast<uint32_t*>(rgba8));
  return hipGetLastError();
}

hipError_t launch_resolve_tiles(hipStream_t st, uint64_t n_tiles, uint32_t s, const float* samples, float* rgb, uint8_t* rgba8) {
  drop_stale_error();
  const uint64_t pixels = n_tiles * (256u / (s * s));
  if (pixels == 0) return hipSuccess;
  hipLaunchKernelGGL(resolve_tiles_kernel, dim3((unsigned)((pixels + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, pixels, s,
                     samples, rgb, reinterpret_cast<uint32_t*>(rgba8));
  return hipGetLastError();
}

hipError_t launch_untile_sampled(hipStream_t st, uint32_t width, uint32_t height, uint32_t s, uint32_t world,
                                 uint64_t rgb_stride_floats, uint64_t px_stride_pixels, const float* rgb_c, float* rgb_f,
                                 const uint8_t* px_c, uint8_t* px_f) {
  drop_stale_error();
  const uint64_t pixels = (uint64_t)width * height;
  if (pixels == 0) return hipSuccess;
  const uint32_t tiles_x = (uint32_t)(((uint64_t)width * s + 31u) / 32u);
  hipLaunchKernelGGL(untile_sampled_kernel, dim3((unsigned)((pixels + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, width, height,
                     s, tiles_x, world, rgb_stride_floats, px_stride_pixels, rgb_c, rgb_f,
                     reinterpret_cast<const uint32_t*>(px_c), reinterpret_cast<uint32_t*>(px_f));
  return hipGetLastError();
}

}  // namespace rtk
