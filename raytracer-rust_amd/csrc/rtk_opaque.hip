// rtk_opaque.hip -- the render path for scenes without reflective or transparent
// materials (every throughput configuration).  One pixel per lane, one 8x8 patch per
// wavefront, one 32x8 tile per workgroup (one patch per workgroup in the fused kernel).
//
//   render_uniform_kernel  every light has w = 0 (what scene files give): camera ray, the
//                    shadow rays of its hit towards all lights (batches of 4), shading and
//                    the store, in one kernel; nothing but the framebuffer crosses HBM
//   render_uniform_kernel<4, false>  the same when some lights have w = 1 (built from points):
//                    their shadow rays form cones with the apex at the light
//   primary_kernel + shadow_general_kernel + shade_kernel   when some light has another w:
//                    camera rays, shadow rays (per-ray cluster walk) and shading as three
//                    kernels with 13 bytes per pixel of hit records in between; shade_kernel
//                    rebuilds LightingData, samples the texture, sums Phong over the lights
//                    and stores FP32 RGB + RGBA8 (transposed through LDS)
//
// Arithmetic of the decisions is the same as in the single-kernel path (rtk_kernels.hip):
// conservative FP32 tests decide which (ray, object) pairs are looked at, then the
// reference's own f64 formulas with contraction off decide the outcome.
//
// Wavefront-level culling.  The 64 rays of a wavefront (an 8x8 pixel patch) are coherent:
// camera rays fill a narrow cone, shadow rays towards a w = 0 light are parallel lines
// through a small ball of shading points.  The spheres are ordered at scene upload into
// spatial clusters of 64 with one bounding ball each (rtk_scene_upload), and a sweep over
// the object list runs in three levels:
//   level 0  the lanes take 64 DIFFERENT clusters each and test their balls against the
//            patch's bundle (cone / fattened lines); a ballot gives the clusters to open;
//   level 1  for an opened cluster the lanes take its 64 member spheres, same test;
//   level 2  for the spheres that are left, the per-ray FP32 filter (one ray per lane, the
//            sphere record wavefront-uniform), then f64 on the candidates it keeps.
// Every level is conservative (it may keep too much, never too little), every (ray, sphere)
// pair of the object list is covered by one of the tests, and the f64 decisions are those of
// a plain sweep.  Records are read straight from global memory (they are wavefront-uniform
// or one coalesced line per wavefront): no LDS staging, no workgroup barriers.
//
// When every light of the scene has w = 0 (lights loaded from scene files,
// src/loader.rs:35-40) the shadow ray's line direction g_k = L_k/|L_k| is the same for
// every shading point, so the per-(sphere, light) part of the level-2 filter is tabulated at
// scene upload: K[j][k] = c_j.g_k - R_j'^2 rho_k.  With P_k = p.g_k per lane,
// h = u.g_k + R^2 rho_k = P_k - K[j][k]: two VALU instructions per (ray, sphere, light) plus
// the shared |p - c|^2 work.

#include "rtk_shading.h"
#include "rtk_sweeps.h"
#include "rtk_queue.h"
#include "rtk_kernels.h"

#pragma clang fp contract(off)

namespace rtk {

// Experiment builds (-DRTK_COUNT_EVENT=n): the exact-test counter counts one kind of event of the fused
// kernel instead -- 1 level-0 rounds, 2 opened clusters, 3 level-2 spheres, 4 exact-test visits of the
// shadow phase (per wavefront, not per lane); 5 exact-test visits, 6 level-2 spheres, 7 opened clusters
// of the camera phase.
#ifdef RTK_COUNT_EVENT
#define RTK_EVENT(n, wc) do { if (RTK_COUNT_EVENT == (n)) (wc).exact += 1u; } while (0)
#define RTK_EXACT(wc, n) do { if (RTK_COUNT_EVENT == 0) (wc).exact += (n); } while (0)
#else
#define RTK_EVENT(n, wc) do { } while (0)
#define RTK_EXACT(wc, n) do { (wc).exact += (n); } while (0)
#endif

// Hides a value from the optimiser so that what is computed from it is recomputed where
// it is used instead of being kept live across a sweep.
__device__ __forceinline__ uint32_t launder(uint32_t v) {
  asm volatile("" : "+v"(v));
  return v;
}

__device__ __forceinline__ bool wave_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }

// Per-lane values parked in LDS between the phases of the fused kernel (the register file is full
// at four wavefronts per SIMD; re-deriving a camera direction costs a square root and three
// divisions, fetching it back three 8-byte LDS reads).  One array of 256 doubles per component,
// so a wavefront's lanes read consecutive addresses.  A lane only ever reads what it wrote itself:
// no barrier is involved.
template <int N>  // threads of the workgroup
struct StashOf {
  double dir[3][N];     // unit camera-ray direction
  double over[3][N];    // over_position of the hit
  double normal[3][N];  // its (flipped) normal
};
using Stash = StashOf<kBlock>;
template <int N>
__device__ __forceinline__ void put3(double (*a)[N], V3 v) {
  a[0][threadIdx.x] = v.x, a[1][threadIdx.x] = v.y, a[2][threadIdx.x] = v.z;
}
template <int N>
__device__ __forceinline__ V3 get3(double (*a)[N]) { return {a[0][threadIdx.x], a[1][threadIdx.x], a[2][threadIdx.x]}; }

// Where a thread's pixel sits: local tile k (global tile k*world + rank), slot (lx, ly).
struct PixelRef {
  uint32_t x, y;   // frame coordinates
  uint32_t local;  // index into the per-launch intermediate buffers
  bool in_range;   // the tile exists
  bool valid;      // ... and the pixel lies inside the frame
};
__device__ __forceinline__ PixelRef pixel_ref(const RenderTarget& T, const DevCamera& C, uint32_t k, uint32_t wave, uint32_t lane) {
  PixelRef p;
  const uint32_t lx = (wave << 3) | (lane & 7u), ly = lane >> 3;
  p.in_range = k < T.n_local_tiles;
  const TileRef tr = tile_ref(T, k);
  p.x = tr.tx * 32u + lx;
  p.y = tr.ty * 8u + ly;
  p.valid = p.in_range && p.x < C.width && p.y < C.height;
  p.local = k * 256u + ly * 32u + lx;
  return p;
}
__device__ __forceinline__ PixelRef pixel_ref(const RenderTarget& T, const DevCamera& C, uint32_t k) {
  return pixel_ref(T, C, k, threadIdx.x >> 6, threadIdx.x & 63u);
}

// ------------------------------------------------------------------------------------
// Bundles.  All fields are wavefront-uniform (they live in SGPRs).

// Cone around the unit camera rays of one patch: axis a, tan and sec of the half-angle.
struct Cone {
  float ax, ay, az, tan_t, sec_t;
};
__device__ __forceinline__ Cone make_cone(float dx, float dy, float dz) {
  Cone c;
  // axis: between the directions of the two pixels either side of the patch's centre (lanes 27 and 36 of
  // the 8x8 patch).  Any axis is sound -- the half-angle below is the largest angle any lane's ray makes
  // with it -- and this one is within a hundredth of a pixel of the mean direction.
  const float sx = lane_value(dx, 27u) + lane_value(dx, 36u), sy = lane_value(dy, 27u) + lane_value(dy, 36u),
              sz = lane_value(dz, 27u) + lane_value(dz, 36u);
  // square roots and reciprocals are the hardware's (one instruction each, good to an ulp) instead of the
  // correctly rounded sequences (16 and 10 instructions): the slack below is a hundred ulps wide
  const float inv = __builtin_amdgcn_rsqf(sx * sx + sy * sy + sz * sz);
  c.ax = wave_uniform(sx * inv), c.ay = wave_uniform(sy * inv), c.az = wave_uniform(sz * inv);
  const float along = dx * c.ax + dy * c.ay + dz * c.az;
  const float qx = dx - along * c.ax, qy = dy - along * c.ay, qz = dz - along * c.az;
  const float across = __builtin_amdgcn_sqrtf(qx * qx + qy * qy + qz * qz);
  // the slack covers the FP32 rounding of the directions and of this computation
  const float t = along > 0.0f ? across * __builtin_amdgcn_rcpf(along) : 3.0e38f;
  c.tan_t = wave_uniform(wave_max_f(t) * (1.0f + 1.0e-5f) + 2.0e-6f);
  c.sec_t = wave_uniform(__builtin_amdgcn_sqrtf(1.0f + c.tan_t * c.tan_t) * (1.0f + 1.0e-6f));
  return c;
}
// Can some ray of the cone (apex at the camera) reach the ball {L, Rs}?  With s = L.a and
// d the distance of L from the axis: d cos(t) - s sin(t) <= Rs.  One sphere per lane.
__device__ __forceinline__ bool cone_reaches(const Cone& c, float4 ball) {
  const float s = __builtin_fmaf(ball.z, c.az, __builtin_fmaf(ball.y, c.ay, ball.x * c.ax));
  const float qx = __builtin_fmaf(-s, c.ax, ball.x), qy = __builtin_fmaf(-s, c.ay, ball.y), qz = __builtin_fmaf(-s, c.az, ball.z);
  const float d2 = __builtin_fmaf(qz, qz, __builtin_fmaf(qy, qy, qx * qx));
  const float rr = __builtin_fmaf(s, c.tan_t, ball.w * c.sec_t);
  return rr >= 0.0f && __builtin_fmaf(rr, rr, -d2) >= 0.0f;
}

// ------------------------------------------------------------------------------------
// Camera-ray phase of one pixel per lane; each wavefront owns one 8x8 patch.  Planes exact,
// spheres through the three levels (clusters, members, per-ray filter) and f64 on what is left.
template <typename StashT = Stash>
__device__ __forceinline__ void primary_phase(const DevScene& S, const DevCamera& C, const float4* __restrict__ prim,
                                              const float4* __restrict__ prim_ball,
                                              const float4* __restrict__ prim_cluster, const PixelRef& px, Closest& best,
                                              WaveCount& wc, StashT* stash = nullptr, bool track = false, bool* graze = nullptr) {
  // track: the entry before the winner is kept as well (scenes whose refractive indices differ), and *graze
  // is set when a sphere has both roots within 1e-5 of each other on the ray (trace_step's rule)
  const V3 origin = ld3(C.origin);
  const uint32_t lane = threadIdx.x & 63u;
  closest_init(best);
  float dx, dy, dz;
  {
    const V3 d = camera_ray_direction(C, px.x, px.y);
    if (stash) put3(stash->dir, d);
    dx = (float)d.x, dy = (float)d.y, dz = (float)d.z;
    for (uint32_t ip = 0; ip < S.n_planes; ++ip) {
      const uint32_t obj = S.plane_object[ip];
      double t0, t1;
      if (px.valid && intersect_object(S.objects[obj], origin, d, 0.0, t0, t1)) closest_consider(best, t0, obj, track);
    }
  }
  if (!wave_any(px.valid)) return;
  // pixels outside the frame still have well-defined directions; they only widen the cone a little
  const Cone cone = make_cone(dx, dy, dz);
#if defined(RTK_PHASES) && RTK_PHASES == 5  // experiment: the camera phase without its sweep (phases1 otherwise)
  if (cone.tan_t + cone.sec_t + cone.ax != 12345.678f) return;
#endif
  for (uint32_t cbase = 0; cbase < S.n_clusters_padded; cbase += kBundle) {
    // level 0: this lane's cluster against the cone
    unsigned long long cm = __builtin_amdgcn_ballot_w64(cone_reaches(cone, prim_cluster[cbase + lane]));
    for (; cm != 0ull; cm &= cm - 1ull) {
      const uint32_t first = (cbase + (uint32_t)__builtin_ctzll(cm)) * kBundle;
      if (first >= S.n_spheres_padded) break;  // cannot happen (padding clusters never pass); keeps a slip from reading out of bounds
      RTK_EVENT(7, wc);
      // level 1: this lane's sphere of the cluster against the cone
      unsigned long long m = __builtin_amdgcn_ballot_w64(cone_reaches(cone, prim_ball[first + lane]));
      // level 2: per-ray filter and exact test for the spheres level 1 kept
      for (; m != 0ull; m &= m - 1ull) {
        RTK_EVENT(6, wc);
        const uint32_t j = first + (uint32_t)__builtin_ctzll(m);
        const float4 r = prim[j];
        float t = r.x * dx;
        t = __builtin_fmaf(r.y, dy, t);
        t = __builtin_fmaf(r.z, dz, t);
        const bool pass = px.valid && __builtin_fmaf(t, t, -r.w) >= 0.0f;
        const unsigned n_pass = lanes_where(pass);
        if (n_pass) {
          RTK_EXACT(wc, n_pass);
          RTK_EVENT(5, wc);
          const RTK_RO SphRecord& rec = S.sph_record[j];
          if (pass) {
            double t0, t1;
            const V3 d = stash ? get3(stash->dir) : camera_ray_direction(C, launder(px.x), px.y);
            if (intersect_sphere_record(S, rec, origin, d, 0.0, t0, t1)) {
              closest_consider(best, t0, rec.object, track);
              closest_consider(best, t1, rec.object, track);
              if (track && fabs(t0 - t1) < 0.00001) *graze = true;
            }
          }
        }
      }
    }
  }
}

// Kernel 1 (scenes with some light of w != 0): camera rays only.  Workgroup b owns local
// tile n - 1 - b (bottom of the frame first).
__global__ void __launch_bounds__(kBlock, 4) primary_kernel(DevScene S, DevCamera C, const float4* __restrict__ prim,
                                                           const float4* __restrict__ prim_ball,
                                                           const float4* __restrict__ prim_cluster, RenderTarget T,
                                                           OpaqueBuffers B, DevCounters* counters) {
  const PixelRef px = pixel_ref(T, C, T.n_local_tiles - 1u - blockIdx.x);
  Closest best;
  WaveCount wc;
  primary_phase(S, C, prim, prim_ball, prim_cluster, px, best, wc);
  B.hit_obj[px.local] = px.valid ? best.obj : 0xffffffffu;
  B.hit_t[px.local] = best.t;
  wc.primary = lanes_where(px.valid);
  flush_counts(counter_shard(counters), wc);
}

// The shading point of a recorded hit: `over_position` of LightingData (src/scene/lighting.rs:50).
__device__ __forceinline__ V3 over_point(const DevScene& S, const DevCamera& C, const PixelRef& p, uint32_t obj, double t) {
  const V3 d = camera_ray_direction(C, p.x, p.y);
  return lighting_data(S.objects[obj], ld3(C.origin), d, t).over;
}

// One shadow ray against one object, exactly: the per-light body of `Scene::is_shadowed`
// (src/scene.rs:208-225).
__device__ __forceinline__ bool shadow_ray_blocked(const RTK_RO DevObject& ob, const RTK_RO DevLight& L, V3 p) {
  V3 lv = ld3(L.pos) - p;
  const double lw = L.pos[3] - 1.0;
  const double m = sqrt_f64(lv.x * lv.x + lv.y * lv.y + lv.z * lv.z + lw * lw);
  double dw;
  const V3 d = divide_by_length(lv, lw, m, dw);
  double t0, t1;
  const int n = intersect_object(ob, p, d, dw, t0, t1);
  return (n >= 1 && t0 > 0.0 && t0 < m) || (n == 2 && t1 > 0.0 && t1 < m);
}

// The same against a sphere given by its sweep-order record.
__device__ __forceinline__ bool shadow_ray_blocked(const DevScene& S, const RTK_RO SphRecord& rec, const RTK_RO DevLight& L, V3 p) {
  V3 lv = ld3(L.pos) - p;
  const double lw = L.pos[3] - 1.0;
  const double m = sqrt_f64(lv.x * lv.x + lv.y * lv.y + lv.z * lv.z + lw * lw);
  double dw;
  const V3 d = divide_by_length(lv, lw, m, dw);
  double t0, t1;
  const int n = intersect_sphere_record(S, rec, p, d, dw, t0, t1);
  return n == 2 && ((t0 > 0.0 && t0 < m) || (t1 > 0.0 && t1 < m));
}

// The tabulated filter of one pixel.  With the recentred, FP32-rounded shading point p,
// per light P_k = p.g_k, and per sphere the record {c, CC = kappa |c|^2 - R'^2} and the row
// K_k = c.g_k - R'^2 rho_k, the general test  (u.g_k + R'^2 rho_k)^2 - (kappa |u|^2 - R'^2) >= thr
// becomes
//     (P_k - K_k)^2 + (2 kappa p.c - CC) >= thr + kappa |p|^2  (= thrp),
// and since max_k (h_k^2 + e) = (max_k |h_k|)^2 + e the square moves out of the light loop:
// 6 VALU instructions per (pixel, sphere) plus 1.5 per (pixel, sphere, light).  |p|^2 and
// |c|^2 now cancel in FP32; the absolute slack in thr pays for that.
template <int KW>
struct PixelFilter {
  float x, y, z;  // recentred shading point; (0,0,0) when parked
  float P[KW];    // p.g_k
  float thrp;     // threshold; +3e38 when parked (nothing passes)
};

// A sphere BEHIND a shadow ray.  Most (pixel, sphere, light) triples the filter keeps are a sphere and a
// point on its own lit side: the line through the point does cross the sphere -- at negative
// parameters.  For a sphere of one radius R (inverse = sigma * identity plus a translation) the
// reference's object-space ray from p towards a w = 0 light at L is sigma u + t sigma (L - u)/m with
// u = p - centre (src/scene.rs:208-225, src/maths/rays.rs:31-41), so its quadratic
// (src/scene/shapes/spheres.rs:18-35) has c = sigma^2 |u|^2 - 1 and b = 2 sigma^2 u.(L - u)/m
// = 2 sigma^2 |L| (u.g + rho |u|^2)/m with g = L/|L|, rho = -1/|L|.  When the computed b and c are both
// >= 0 no root is positive: the computed discriminant b*b - 4*a*c is at most fl(b*b), whose correctly
// rounded square root is b itself, so sqrt(disc) <= b, both numerators -b -+ sqrt(disc) are <= 0, and
// `t > 0` fails for both quotients (or they are NaN, a = 0); a negative discriminant gives no roots at
// all.  The reference then finds the light unblocked by this sphere, and the exact test can be left out.
// Both signs are certified in FP32, for the records {c32, R'} of the wavefront-level test (R' covers
// the rounding of the centre) and u32 = f - c32 formed directly (the expanded form of the tabulated
// filter loses the 1e-4 by which a shading point stands off its surface):
//   outside:      |u32| (1 - 8 eps) - R' >= Q1
//                                   since |u| - R >= |u32| - R' - dp, dp the rounding of the recentred point;
//   moving away:  h_k - |rho_k| gap >= Q3,  gap >= |u32|^2 - R'^2
//                                   since h is within dh of u32.g + rho R'^2, and u.g + rho |u|^2 >=
//                                   u32.g + rho |u32|^2 - dd (1 + |rho| (2 |u32| + dd)) for dd = dp + the
//                                   rounding of the centre.
// Q1 and Q3 also hold the f64 rounding of the reference's own c and b (a few 1e-16 of the ABSOLUTE
// coordinates' magnitudes, the recentring offset included) with a factor of several hundred to spare.
// The margins are per-pixel quantities; the kernel uses their maxima over the wavefront.
struct BehindMargins {
  float q1, q3;
};
template <int KW>
__device__ __forceinline__ void park(PixelFilter<KW>& f) {
  f.x = f.y = f.z = 0.0f;
#pragma unroll
  for (int k = 0; k < KW; ++k) f.P[k] = 0.0f;
  f.thrp = 3.0e38f;
}
// Returns |pc| (FP32), the one per-pixel quantity the margins of the behind test depend on.
template <int KW>
__device__ __forceinline__ float arm(PixelFilter<KW>& f, const DevScene& S, V3 p, uint32_t kb, uint32_t nb) {
  const V3 pc = {p.x - S.center[0], p.y - S.center[1], p.z - S.center[2]};
  f.x = (float)pc.x, f.y = (float)pc.y, f.z = (float)pc.z;
  float rho_max = 0.0f;
#pragma unroll
  for (int k = 0; k < KW; ++k) {
    const float4 lf = S.light_filter[kb + min((uint32_t)k, nb - 1u)];  // a short batch repeats its last light
    f.P[k] = __builtin_fmaf(f.z, lf.z, __builtin_fmaf(f.y, lf.y, f.x * lf.x));
    rho_max = fmaxf(rho_max, fabsf(lf.w) * 1.0001f);
  }
  // slack: everything filter_threshold() covers, plus
  //  - h = P - K, a difference of rounded numbers: |dh| <= eps (6|p| + 2 c_max + 2B)
  //  - the expanded |u|^2: roundings of p.c, CC and the two fused multiply-adds,
  //    <= eps (5|p|^2 + 6 c_max^2 + 2 R_max^2)
  // The bounds are sums and products of non-negative terms: they are evaluated in FP32 (two dozen
  // operations, a few 1e-6 of relative error at most) and inflated by 1e-4, which keeps them above the
  // f64 expressions they stand for (tests/filter_model.py evaluates those); only |f|^2 in the threshold
  // itself needs all its digits.
  const float e32 = kEps32, up = 1.0f + 1.0e-4f;
  const float pn2 = __builtin_fmaf(f.z, f.z, __builtin_fmaf(f.y, f.y, f.x * f.x));  // |pc|^2 within a few eps
  const float pn = __builtin_amdgcn_sqrtf(pn2);
  const float B = S.r2_max * rho_max;
  const float dh = e32 * (6.0f * pn + 2.0f * S.c_max + 2.0f * B) * 1.01f;
  const float hmax = pn + S.c_max + B;
  const float extra = 2.0f * hmax * dh + dh * dh + 6.0f * e32 * (pn2 + S.c_max * S.c_max + S.r2_max) * 1.01f;
  const float dp = pn * e32 * 1.0001f;  // rounding of the recentred point (as in filter_threshold)
  const float r2s = 2.0f * S.r_max * dp + dp * dp;
  const float slack = r2s * (1.0f + 2.0f * hmax * rho_max) + 64.0f * e32 * (B * B + S.r2_max);
  const float thr = -((slack + extra) * 1.001f * up + 1.0e-30f);
  const double pp = (double)f.x * f.x + (double)f.y * f.y + (double)f.z * f.z;
  const double t = (double)kKappa * pp + (double)thr;
  // rounded to nearest and then lowered by at least two ulps (a lower threshold keeps more): below the
  // next FP32 value under t, which is what the bound needs
  const float tf = (float)t;
  f.thrp = tf - __builtin_fmaf(fabsf(tf), 0x1p-22f, 1.0e-30f);
  return pn;
}
// Margins of the behind test (see BehindMargins) for a pixel whose recentred point has length pn, towards
// lights of |rho| <= rho_max.  Every term is a sum or product of non-negative numbers and FP32 rounding is
// monotone, so the value at the largest pn of a wavefront is at least the value at each of its pixels: the
// kernel evaluates the margins once, at that maximum.
__device__ __forceinline__ BehindMargins behind_margins(const DevScene& S, float pn, float rho_max) {
  const float e32 = kEps32, up = 1.0f + 1.0e-4f;
  const float B = S.r2_max * rho_max;
  const float dh = e32 * (6.0f * pn + 2.0f * S.c_max + 2.0f * B) * 1.01f;
  const float hmax = pn + S.c_max + B;
  const float dp = pn * e32 * 1.0001f;              // rounding of the recentred point (as in filter_threshold)
  const float reach = pn * (1.0f + e32) + S.c_max;  // >= |u32| for every sphere
  const float absolute = reach + (float)fabs(S.center[0]) + (float)fabs(S.center[1]) + (float)fabs(S.center[2]);  // >= |p| + |centre| in world coordinates
  const float dd = dp + e32 * S.c_max * 1.01f;      // rounding of the recentred point and of a recentred centre
  BehindMargins bm;
  bm.q1 = (dp * 1.01f + 1.0e-12f * (absolute + reach)) * up + 1.0e-30f;
  bm.q3 = (dh * 1.01f + dd * (1.0f + rho_max * (2.0f * reach + dd)) * 1.01f +
           1.01f * e32 * (hmax + rho_max * (2.0f * reach * reach + S.r2_max)) +
           1.0e-12f * reach * (1.0f + 2.0f * rho_max * absolute)) * up + 1.0e-30f;
  return bm;
}
constexpr float kTwoKappa = 2.0f * kKappa;

// Ball around the live shading points of one patch (recentred FP32 coordinates).
struct Ball {
  float mx, my, mz, radius;
  bool any;
};
__device__ __forceinline__ Ball make_ball(bool live, float x, float y, float z) {
  Ball b;
  const float n = (float)__builtin_popcountll(__builtin_amdgcn_ballot_w64(live));
  b.any = n > 0.0f;
  const float inv = __builtin_amdgcn_rcpf(fmaxf(n, 1.0f));  // any centre will do: the radius below is measured from it
  b.mx = wave_uniform(wave_sum_f(live ? x : 0.0f) * inv);
  b.my = wave_uniform(wave_sum_f(live ? y : 0.0f) * inv);
  b.mz = wave_uniform(wave_sum_f(live ? z : 0.0f) * inv);
  const float ex = x - b.mx, ey = y - b.my, ez = z - b.mz;
  const float far = live ? __builtin_amdgcn_sqrtf(ex * ex + ey * ey + ez * ez) : 0.0f;  // an ulp: the 1e-5 below covers it
  b.radius = wave_uniform(wave_max_f(far) * (1.0f + 1.0e-5f));
  return b;
}

// ------------------------------------------------------------------------------------
// Shadow phase of one pixel per lane when every light has w = 0: `Scene::is_shadowed`
// (src/scene.rs:208-225) for the shading point p.  KW = lights per batch (4 or 8).
//
// Level 0/1 for one light: the patch's shadow lines are parallel to g and pass through the
// ball {m, rho_w}.  A line through p can touch sphere {c, R} only if its distance from c is
// at most R_e = R sqrt(1 + 2 |rho| (-s) + R^2 rho^2), s = (p - c).g (the reference's growing
// ball, see PixelFilter).  Over the ball of points: distance(c, line through m) <=
// rho_w + R + R |rho| max(rho_w - s_m, 0) + R^3 rho^2 / 2.  For a cluster {C, Rc} around its
// members' balls the same holds with R -> Rc in the reach, s_m -> s_m - Rc, and the growth
// terms taken with the scene's largest sphere radius.
template <int KW>
__device__ __forceinline__ bool shadow_phase_uniform(const DevScene& S, bool need, V3 p, WaveCount& wc) {
  constexpr int KV = KW / 4;  // float4 per table row
  const uint32_t lane = threadIdx.x & 63u;
  bool in_shadow = false;
  for (uint32_t kb = 0, batch = 0; kb < S.n_lights; kb += KW, ++batch) {
    if (!wave_any(need)) break;
    const uint32_t nb = min((uint32_t)KW, S.n_lights - kb);
    uint32_t occl = 0;
    PixelFilter<KW> f;
    park(f);
    float pn = 0.0f;  // |recentred point| of a live pixel
    bool live = false;
    if (need) {
      for (uint32_t ip = 0; ip < S.n_planes; ++ip) {
        const RTK_RO DevObject& plane = S.objects[S.plane_object[ip]];
        const PlaneSide side = plane_side(plane, p);
        for (uint32_t r = 0; r < nb; ++r)
          if (plane_may_block(side, S.plane_light[ip * S.n_lights + kb + r]) && shadow_ray_blocked(plane, S.lights[kb + r], p))
            occl |= 1u << r;
      }
      live = !(occl & 1u);
      if (live) pn = arm(f, S, p, kb, nb);
    }
    const RTK_RO float4* ktab = S.k_table + (size_t)batch * S.n_spheres_padded * KV;
    // bundle of the patch; the slack covers the FP32 rounding of points and centres
    const Ball ball = make_ball(live, f.x, f.y, f.z);
    float lgx[KW], lgy[KW], lgz[KW], lrho[KW];
    float grow = 0.0f;
#pragma unroll
    for (int k = 0; k < KW; ++k) {
      const float4 lf = S.light_filter[kb + min((uint32_t)k, nb - 1u)];
      lgx[k] = lf.x, lgy[k] = lf.y, lgz[k] = lf.z, lrho[k] = fabsf(lf.w) * 1.0001f;
      grow = fmaxf(grow, 0.5f * S.r_max * S.r2_max * lf.w * lf.w);
    }
    const float pad = ball.radius + grow * 1.01f +
                      16.0f * kEps32 * (fabsf(ball.mx) + fabsf(ball.my) + fabsf(ball.mz) + ball.radius + S.c_max);
    // the margins of the behind test at the wavefront's largest |point|: at least every live pixel's own
    float rho_top = 0.0f;
#pragma unroll
    for (int k = 0; k < KW; ++k) rho_top = fmaxf(rho_top, lrho[k]);
    const BehindMargins bm = behind_margins(S, wave_uniform(wave_max_f(pn)), rho_top);
    const float Q1 = bm.q1, Q3 = bm.q3;

    // rec = {centre, radius} of a sphere (t_extra = 0, growth radius = its own) or of a
    // cluster (t_extra = its radius, growth radius = the scene's largest sphere radius).
    // The squared distance of the centre from the line along g_k through m is formed as |u|^2 - s^2: one
    // multiply-add per light instead of six for |u - s g_k|^2.  That difference carries the roundings of
    // |u|^2 and of s and the length of the rounded g_k to first order, 13 eps |u|^2 at most; `room` allows
    // 32.  The factors that cover the roundings of t and of the reach are folded into the per-light
    // constant and into the reach (each part of rr stays inflated by more than its own roundings), and a
    // record passes no light when its reach is negative (padding records), so that is tested once.
    float lgrow[KW];
#pragma unroll
    for (int k = 0; k < KW; ++k) lgrow[k] = lrho[k] * ((1.0f + 8.0f * kEps32) * (1.0f + 4.0f * kEps32) * (1.0f + 4.0f * kEps32));
    auto lines_reach = [&](float4 rec, float t_extra, float r_grow) {
      const float ux = ball.mx - rec.x, uy = ball.my - rec.y, uz = ball.mz - rec.z;
      const float uu = __builtin_fmaf(uz, uz, __builtin_fmaf(uy, uy, ux * ux));
      const float reach = (rec.w + pad) * (1.0f + 4.0f * kEps32);  // R + rho_w + slack
      const float tb = ball.radius + t_extra;
      float best = -3.0e38f;  // max over the lights of rr^2 - (|u|^2 - s^2)
#pragma unroll
      for (int k = 0; k < KW; ++k) {
        const float s = __builtin_fmaf(uz, lgz[k], __builtin_fmaf(uy, lgy[k], ux * lgx[k]));
        const float t = fmaxf(tb - s, 0.0f);
        const float rr = __builtin_fmaf(t, r_grow * lgrow[k], reach);
        best = fmaxf(best, __builtin_fmaf(rr, rr, __builtin_fmaf(s, s, -uu)));
      }
      return reach >= 0.0f && __builtin_fmaf(uu, 32.0f * kEps32, best) >= 0.0f;
    };

    bool wave_live = wave_any(live);
#if defined(RTK_PHASES) && RTK_PHASES == 3  // experiment: the shadow phase's set-up and accounting without its sweep
    wave_live = wave_live && Q1 + Q3 + pad + f.thrp + lgrow[0] == 12345.678f;
#endif
    for (uint32_t cbase = 0; wave_live && cbase < S.n_clusters_padded; cbase += kBundle) {
      // level 0: this lane's cluster against the patch's line bundles
      RTK_EVENT(1, wc);
      const float4 cl = S.cluster_ball[cbase + lane];
      unsigned long long cm = __builtin_amdgcn_ballot_w64(lines_reach(cl, cl.w, S.r_max));
      for (; wave_live && cm != 0ull; cm &= cm - 1ull) {
        const uint32_t first = (cbase + (uint32_t)__builtin_ctzll(cm)) * kBundle;
        if (first >= S.n_spheres_padded) break;  // cannot happen (padding clusters never pass); keeps a slip from reading out of bounds
        // level 1: this lane's sphere of the cluster
        RTK_EVENT(2, wc);
        const float4 rec = S.sph_ball[first + lane];
        unsigned long long m = __builtin_amdgcn_ballot_w64(lines_reach(rec, 0.0f, rec.w));
        const unsigned long long plain = S.cluster_plain[first / kBundle];
        bool changed = false;
        // level 2: per-ray tabulated filter, then the exact test
        for (; m != 0ull; m &= m - 1ull) {
          const uint32_t member = (uint32_t)__builtin_ctzll(m), j = first + member;
          RTK_EVENT(3, wc);
          const float4 c = S.sph_tab[j];
          const float pc = __builtin_fmaf(f.z, c.z, __builtin_fmaf(f.y, c.y, f.x * c.x));
          const float e = __builtin_fmaf(pc, kTwoKappa, -c.w);
          // behind test (see BehindMargins): the point's distance from the centre against R', and an
          // upper bound of |u32|^2 - R'^2
          const float rj = lane_value(rec.w, member);
          const float ux = f.x - c.x, uy = f.y - c.y, uz = f.z - c.z;
          const float uu = __builtin_fmaf(uz, uz, __builtin_fmaf(uy, uy, ux * ux));
          const bool outside = ((plain >> member) & 1ull) != 0ull && __builtin_fmaf(__builtin_amdgcn_sqrtf(uu), 1.0f - 8.0f * kEps32, -rj) >= Q1;
          const float gap = __builtin_fmaf(uu, 1.0f + 8.0f * kEps32, -(rj * rj) * (1.0f - 1.0e-6f)) * (1.0f + 4.0f * kEps32);
          uint32_t mask = 0;
#pragma unroll
          for (int v = 0; v < KV; ++v) {
            const float4 row = ktab[(size_t)j * KV + v];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float h = f.P[v * 4 + i] - (&row.x)[i];
              const bool behind = outside && __builtin_fmaf(-lrho[v * 4 + i], gap, h) >= Q3;
              if (__builtin_fmaf(h, h, e) >= f.thrp && !behind) mask |= 1u << (v * 4 + i);
            }
          }
          mask &= ~occl;
          if (wave_any(mask != 0)) {
            const RTK_RO SphRecord& rec = S.sph_record[j];
            for (uint32_t r = 0; r < nb; ++r) {  // one light after the other, for the lanes that kept it
              const bool run = ((mask >> r) & 1u) != 0u;
              const unsigned n_run = lanes_where(run);
              if (n_run == 0u) continue;
              RTK_EXACT(wc, n_run);
              RTK_EVENT(4, wc);
              if (run && shadow_ray_blocked(S, rec, S.lights[kb + r], p)) occl |= 1u << r;
            }
            changed = true;
          }
        }
        if (changed) {
          // a pixel whose first light is blocked needs nothing more: park it
          if (live && (occl & 1u)) {
            live = false;
            park(f);
          }
          wave_live = wave_any(live);
        }
      }
    }
    // the reference stops at the first blocked light (src/scene.rs:208-225): a pixel traces the lights
    // of the batch up to and including that one
    unsigned traced = 0;
    if (need) {
      occl &= (nb >= 32u) ? 0xffffffffu : ((1u << nb) - 1u);
      traced = nb;
      if (occl) {
        traced = (unsigned)__builtin_ctz(occl) + 1u;
        in_shadow = true;
        need = false;
      }
    }
    wc.shadow += lanes_total<bits_for(KW)>(traced);
  }
  return in_shadow;
}

// Kernel 2b: shadow rays for arbitrary lights (some w != 0): one pixel per lane, per-lane
// line directions, LB lights fused.
template <int LB>
__global__ void __launch_bounds__(kBlock, 4) shadow_general_kernel(DevScene S, DevCamera C, RenderTarget T, OpaqueBuffers B,
                                                                  DevCounters* counters) {
  const uint32_t k = T.n_local_tiles - 1u - blockIdx.x;
  const PixelRef px = pixel_ref(T, C, k);
  uint32_t obj = 0xffffffffu;
  if (px.valid) obj = B.hit_obj[px.local];
  const bool hit = obj != 0xffffffffu;
  V3 p = {0, 0, 0};
  if (hit) p = over_point(S, C, px, obj, B.hit_t[px.local]);
  WaveCount wc;
  const bool in_shadow = shadow_sweeps<LB, false>(S, hit, p, wc);
  B.shadow[px.local] = in_shadow ? 1u : 0u;
  flush_counts(counter_shard(counters), wc);
}

// ------------------------------------------------------------------------------------
// Shadow phase when every light has w == 0 or w == 1 but not all have w == 0.  The shadow
// rays of a patch towards a w == 1 light are segments from the ball {m, rho_w} of shading
// points to the light: a cone with its apex AT THE LIGHT (the camera-ray test with the light
// in place of the camera).  Level 2 uses the general per-ray form with the line direction
// rebuilt per lane: g = L/|L| for w == 0, (L - p)/|L - p| for w == 1 (rho = 0).
template <int KW>
__device__ __forceinline__ bool shadow_phase_bundled(const DevScene& S, bool need, V3 p, WaveCount& wc) {
  const uint32_t lane = threadIdx.x & 63u;
  bool in_shadow = false;
  for (uint32_t kb = 0; kb < S.n_lights; kb += KW) {
    if (!wave_any(need)) break;
    const uint32_t nb = min((uint32_t)KW, S.n_lights - kb);
    uint32_t occl = 0;
    bool live = false;
    if (need) {
      for (uint32_t ip = 0; ip < S.n_planes; ++ip) {
        const RTK_RO DevObject& plane = S.objects[S.plane_object[ip]];
        const PlaneSide side = plane_side(plane, p);
        for (uint32_t r = 0; r < nb; ++r)
          if (plane_may_block(side, S.plane_light[ip * S.n_lights + kb + r]) && shadow_ray_blocked(plane, S.lights[kb + r], p))
            occl |= 1u << r;
      }
      live = !(occl & 1u);
    }
    const V3 pc = {p.x - S.center[0], p.y - S.center[1], p.z - S.center[2]};
    const float fx = (float)pc.x, fy = (float)pc.y, fz = (float)pc.z;
    const Ball ball = make_ball(live, fx, fy, fz);

    // per light (wavefront-uniform): line direction or cone; a short batch repeats its last light
    float ax[KW], ay[KW], az[KW];  // w == 0: g; w == 1: cone axis
    float bx[KW], by[KW], bz[KW];  // w == 1: apex (the light, recentred)
    float lt[KW], ls[KW];          // w == 0: |rho|, unused; w == 1: tan, sec of the cone's half-angle
    float rho[KW];                 // signed rho of the per-ray filter (0 for w == 1)
    uint32_t cones = 0, open_cones = 0;  // bit k: light k has w == 1 / its cone is everything (light inside the ball)
    float grow = 0.0f, rho_max = 0.0f, apex_max = 0.0f;
#pragma unroll
    for (int k = 0; k < KW; ++k) {
      const uint32_t li = kb + min((uint32_t)k, nb - 1u);
      const float4 lf = S.light_filter[li];
      if (S.light_kind[li] == 1u) {
        cones |= 1u << k;
        bx[k] = lf.x, by[k] = lf.y, bz[k] = lf.z;
        const float ex = ball.mx - lf.x, ey = ball.my - lf.y, ez = ball.mz - lf.z;
        const float D = sqrtf(ex * ex + ey * ey + ez * ez);
        const float rw = ball.radius * (1.0f + 1.0e-5f) + 8.0f * kEps32 * (fabsf(lf.x) + fabsf(lf.y) + fabsf(lf.z) + D);
        const float inv = 1.0f / fmaxf(D, 1.0e-30f);
        ax[k] = ex * inv, ay[k] = ey * inv, az[k] = ez * inv;
        const float leg2 = D * D - rw * rw;  // squared tangent length from the light to the ball
        if (!(leg2 > 1.0e-6f * D * D)) open_cones |= 1u << k;
        const float tn = rw / sqrtf(fmaxf(leg2, 1.0e-30f)) * (1.0f + 1.0e-5f) + 2.0e-6f;
        lt[k] = tn;
        ls[k] = sqrtf(1.0f + tn * tn) * (1.0f + 1.0e-6f);
        rho[k] = 0.0f;
        apex_max = fmaxf(apex_max, fabsf(lf.x) + fabsf(lf.y) + fabsf(lf.z));
      } else {
        ax[k] = lf.x, ay[k] = lf.y, az[k] = lf.z;
        bx[k] = by[k] = bz[k] = 0.0f;
        lt[k] = fabsf(lf.w) * 1.0001f;
        ls[k] = 0.0f;
        rho[k] = lf.w;
        rho_max = fmaxf(rho_max, fabsf(lf.w) * 1.0001f);
        grow = fmaxf(grow, 0.5f * S.r_max * S.r2_max * lf.w * lf.w);
      }
    }
    const float slack = 16.0f * kEps32 * (fabsf(ball.mx) + fabsf(ball.my) + fabsf(ball.mz) + ball.radius + S.c_max + apex_max);
    const float pad = ball.radius + grow * 1.01f + slack;
    // per-lane threshold of the level-2 filter
    const float thr = need ? filter_threshold(S, pc, rho_max) : 3.0e38f;

    // rec = {centre, radius} of a sphere (t_extra = 0, growth radius = its own) or of a
    // cluster (t_extra = its radius, growth radius = the scene's largest sphere radius)
    auto bundles_reach = [&](float4 rec, float t_extra, float r_grow) {
      bool pass = false;
#pragma unroll
      for (int k = 0; k < KW; ++k) {
        if ((cones >> k) & 1u) {
          if ((open_cones >> k) & 1u) {
            pass |= rec.w >= 0.0f;  // every real record; padding records carry a negative radius
          } else {
            const float lx = rec.x - bx[k], ly = rec.y - by[k], lz = rec.z - bz[k];
            const float s = __builtin_fmaf(lz, az[k], __builtin_fmaf(ly, ay[k], lx * ax[k]));
            const float qx = __builtin_fmaf(-s, ax[k], lx), qy = __builtin_fmaf(-s, ay[k], ly), qz = __builtin_fmaf(-s, az[k], lz);
            const float d2 = __builtin_fmaf(qz, qz, __builtin_fmaf(qy, qy, qx * qx));
            const float rr = __builtin_fmaf(s, lt[k], (rec.w + slack) * ls[k]) * (1.0f + 4.0f * kEps32);
            pass |= rr >= 0.0f && __builtin_fmaf(rr, rr, -d2) >= 0.0f;
          }
        } else {
          const float ux = ball.mx - rec.x, uy = ball.my - rec.y, uz = ball.mz - rec.z;
          const float s = __builtin_fmaf(uz, az[k], __builtin_fmaf(uy, ay[k], ux * ax[k]));
          const float qx = __builtin_fmaf(-s, ax[k], ux), qy = __builtin_fmaf(-s, ay[k], uy), qz = __builtin_fmaf(-s, az[k], uz);
          const float d2 = __builtin_fmaf(qz, qz, __builtin_fmaf(qy, qy, qx * qx));
          const float t = fmaxf(ball.radius + t_extra - s, 0.0f) * (1.0f + 8.0f * kEps32);
          const float rr = __builtin_fmaf(t, r_grow * lt[k], rec.w + pad) * (1.0f + 4.0f * kEps32);
          pass |= rr >= 0.0f && __builtin_fmaf(rr, rr, -d2) >= 0.0f;
        }
      }
      return pass;
    };

    bool wave_live = wave_any(live);
    for (uint32_t cbase = 0; wave_live && cbase < S.n_clusters_padded; cbase += kBundle) {
      const float4 cl = S.cluster_ball[cbase + lane];
      unsigned long long cm = __builtin_amdgcn_ballot_w64(bundles_reach(cl, cl.w, S.r_max));
      for (; wave_live && cm != 0ull; cm &= cm - 1ull) {
        const uint32_t first = (cbase + (uint32_t)__builtin_ctzll(cm)) * kBundle;
        if (first >= S.n_spheres_padded) break;  // cannot happen (padding clusters never pass); keeps a slip from reading out of bounds
        const float4 rec = S.sph_ball[first + lane];
        unsigned long long m = __builtin_amdgcn_ballot_w64(bundles_reach(rec, 0.0f, rec.w));
        bool changed = false;
        // level 2: the general per-ray filter (see sweep_filter in rtk_sweeps.h), then the exact test
        for (; m != 0ull; m &= m - 1ull) {
          const uint32_t j = first + (uint32_t)__builtin_ctzll(m);
          const float4 c = S.sph_filter[j];
          const float ux = fx - c.x, uy = fy - c.y, uz = fz - c.z;
          const float uu = __builtin_fmaf(uz, uz, __builtin_fmaf(uy, uy, ux * ux));
          const float cf = __builtin_fmaf(uu, kKappa, -c.w);
          uint32_t mask = 0;
#pragma unroll
          for (int k = 0; k < KW; ++k) {
            float h;
            if ((cones >> k) & 1u) {
              // direction from the f64 difference, normalised in FP32 (rsq: 1 ulp); the factor on h^2
              // below covers the direction's length being off by a few ulp
              const RTK_RO DevLight& L = S.lights[kb + min((uint32_t)k, nb - 1u)];
              const float gx = (float)(L.pos[0] - p.x), gy = (float)(L.pos[1] - p.y), gz = (float)(L.pos[2] - p.z);
              const float gg = __builtin_fmaf(gz, gz, __builtin_fmaf(gy, gy, gx * gx));
              h = __builtin_fmaf(uz, gz, __builtin_fmaf(uy, gy, ux * gx)) * __builtin_amdgcn_rsqf(gg);
              if (!(gg > 0.0f && gg < 3.0e38f)) h = 3.0e18f;  // light at the shading point, or out of FP32 range: leave it to the exact test
            } else {
              h = __builtin_fmaf(uz, az[k], __builtin_fmaf(uy, ay[k], __builtin_fmaf(ux, ax[k], c.w * rho[k])));
            }
            if (__builtin_fmaf(h * (1.0f + 16.0f * kEps32), h, -cf) >= thr) mask |= 1u << k;
          }
          if (!live) mask = 0;
          mask &= ~occl;
          if (wave_any(mask != 0)) {
            const RTK_RO SphRecord& rec = S.sph_record[j];
            for (uint32_t r = 0; r < nb; ++r) {  // one light after the other, for the lanes that kept it
              const bool run = ((mask >> r) & 1u) != 0u;
              const unsigned n_run = lanes_where(run);
              if (n_run == 0u) continue;
              RTK_EXACT(wc, n_run);
              RTK_EVENT(4, wc);
              if (run && shadow_ray_blocked(S, rec, S.lights[kb + r], p)) occl |= 1u << r;
            }
            changed = true;
          }
        }
        if (changed) {
          if (live && (occl & 1u)) live = false;
          wave_live = wave_any(live);
        }
      }
    }
    // the reference stops at the first blocked light (src/scene.rs:208-225): a pixel traces the lights
    // of the batch up to and including that one
    unsigned traced = 0;
    if (need) {
      occl &= (nb >= 32u) ? 0xffffffffu : ((1u << nb) - 1u);
      traced = nb;
      if (occl) {
        traced = (unsigned)__builtin_ctz(occl) + 1u;
        in_shadow = true;
        need = false;
      }
    }
    wc.shadow += lanes_total<bits_for(KW)>(traced);
  }
  return in_shadow;
}

// ------------------------------------------------------------------------------------
// `apply_lighting` (src/scene.rs:172-205) for a hit on an opaque material: the reflected and
// refracted parts are BLACK, what is left is ambient + the Phong sum over the lights.
__device__ __forceinline__ V3 shade_hit(const DevScene& S, const DevCamera& C, const PixelRef& px, uint32_t obj, double t,
                                        bool in_shadow) {
  const RTK_RO DevObject& ob = S.objects[obj];
  const V3 d = camera_ray_direction(C, px.x, px.y);
  const HitGeom g = lighting_data(ob, ld3(C.origin), d, t);
  const V3 tex = sample_texture_at(S, ob, obj, g.objp);
  V3 surface = ld3(S.ambient);
  for (uint32_t l = 0; l < S.n_lights; ++l)
    surface = surface + phong_lighting(S.lights[l], ob, tex, g.over, g.eye, g.normal, in_shadow);
  return surface;
}

// Kernel 3: shading and store from the hit records of the ray kernel(s).
__global__ void __launch_bounds__(kBlock) shade_kernel(DevScene S, DevCamera C, RenderTarget T, OpaqueBuffers B,
                                                      DevCounters* counters) {
  __shared__ float s_rgb[768];
  __shared__ uint32_t s_px[256];
  const uint32_t k = blockIdx.x;
  const PixelRef px = pixel_ref(T, C, k);
  V3 color = {0, 0, 0};
  if (px.valid) {
    color = ld3(S.ambient);  // a miss, or trace_inner at depth >= MAX_DEPTH, returns ambient_color
    const uint32_t obj = B.hit_obj[px.local];
    if (obj != 0xffffffffu) color = shade_hit(S, C, px, obj, B.hit_t[px.local], B.shadow[px.local] != 0u);
  }
  store_tile(C, T, k, color, s_rgb, s_px);
}

// `apply_lighting` for the fused kernel: the hit geometry comes back from the stash
// (LightingData::calculate ran once, in the shadow set-up) instead of being derived again.
template <typename StashT>
__device__ __forceinline__ V3 shade_hit_stashed(const DevScene& S, uint32_t obj, bool in_shadow, StashT* stash) {
  const RTK_RO DevObject& ob = S.objects[obj];
  const V3 over = get3(stash->over), normal = get3(stash->normal), eye = -get3(stash->dir);
  const V3 objp = xform(ob.inv, over, 1.0);  // as LightingData computes it (src/scene/lighting.rs:52)
  const V3 tex = sample_texture_at(S, ob, obj, objp);
  V3 surface = ld3(S.ambient);
#if defined(RTK_PHASES) && RTK_PHASES == 4  // experiment: shading without the sum over the lights
  return surface + tex + normal + eye;
#endif
  for (uint32_t l = 0; l < S.n_lights; ++l)
    surface = surface + phong_lighting(S.lights[l], ob, tex, over, eye, normal, in_shadow);
  return surface;
}

// All three steps in one kernel (every light has w = 0 or 1): no hit records cross HBM.
// Five wavefronts per SIMD: with the object records in scalar registers the kernel needs 96 VGPRs
// (built without machine LICM, see the Makefile).
// ONE wavefront per workgroup (an 8x8 patch, a quarter of a tile).  The four wavefronts of a tile differ
// in what their patches cost; as one workgroup the fast ones waited for the slow one -- at the barrier
// of a store transposed through LDS, and then for the workgroup's registers and LDS to be released
// together (a sixth of a wavefront's lifetime on C4; a 4096^2 C5 frame 1.07 -> 0.89 ms without).  Here
// a wavefront stores its own 64 pixels (rows of 8: 32 bytes of RGBA8, 96 of FP32) and leaves.
// Workgroups go round the eight XCDs, each with an L2 of its own, so the quarters of a tile are given
// to consecutive workgroups OF ONE XCD: its L2 sees the whole 128- and 384-byte tile rows before it
// writes them back.
template <int KW, bool TABULATED>
__global__ void __launch_bounds__(64, 5) render_uniform_kernel(DevScene S, DevCamera C, const float4* __restrict__ prim,
                                                            const float4* __restrict__ prim_ball,
                                                            const float4* __restrict__ prim_cluster, RenderTarget T,
                                                            DevCounters* counters) {
  __shared__ StashOf<64> stash;
  const PatchOfGroup pg = patch_of_group();
  const uint32_t quarter = pg.quarter, tile = pg.tile;
  if (tile >= T.n_local_tiles) return;
  const uint32_t k = T.n_local_tiles - 1u - tile;  // expensive tiles (bottom of the frame) first
  const PixelRef px = pixel_ref(T, C, k, quarter, threadIdx.x);
  Closest best;
  WaveCount wc;
  wc.primary = lanes_where(px.valid);
  primary_phase(S, C, prim, prim_ball, prim_cluster, px, best, wc, &stash);
  const uint32_t obj = px.valid ? best.obj : 0xffffffffu;
  const bool hit = obj != 0xffffffffu;
  V3 p = {0, 0, 0};
  if (hit) {
    const HitGeom g = lighting_data(S.objects[obj], ld3(C.origin), get3(stash.dir), best.t);
    put3(stash.over, g.over);
    put3(stash.normal, g.normal);
    p = g.over;
  }
#if defined(RTK_PHASES) && (RTK_PHASES == 1 || RTK_PHASES == 5)  // experiment: the camera phase and the hit geometry only
  store_patch(C, T, k, quarter, V3{best.t, p.x, p.y});
  flush_counts(counters + (tile % kCounterShards), wc);
  return;
#endif
  const bool in_shadow = TABULATED ? shadow_phase_uniform<KW>(S, hit, p, wc) : shadow_phase_bundled<KW>(S, hit, p, wc);
#if defined(RTK_PHASES) && RTK_PHASES == 2  // experiment: ... and the shadow phase
  store_patch(C, T, k, quarter, V3{best.t, p.x, in_shadow ? 1.0 : 0.0});
  flush_counts(counters + (tile % kCounterShards), wc);
  return;
#endif
  V3 color = {0, 0, 0};
  if (px.valid) color = hit ? shade_hit_stashed(S, launder(obj), in_shadow, &stash) : ld3(S.ambient);
  store_patch(C, T, k, quarter, color);
  flush_counts(counters + (tile % kCounterShards), wc);
}

// Level 0 of the level-by-level order (rtk_wavefront.hip) on the bundle tests of the fused kernel:
// the camera rays of the pass's tiles, their shadow rays and their shading exactly as above, but
// the pixel is not stored -- what the camera ray's node adds to it goes to the node's record, and the
// reflected and refracted children of the hit are queued.  For scenes whose lights all have w = 0
// (TABULATED) or w = 0 or 1, as for the fused kernel; one workgroup per tile, node = workgroup * 256 + thread.  TRACK: the scene's refractive indices differ,
// so a camera ray that hits a reflecting or transmitting surface needs n1, the material of the entry before
// the winner on its line -- something BEHIND the camera, which the cone tests never look at: those lanes walk
// the scene's cell grid from the camera against the ray's direction (grid_march), and a ray on which some
// sphere has both roots within 1e-5 of each other takes the literal evaluation, as in find_hit.
#ifndef RTK_QP_OCC
#define RTK_QP_OCC 5
#endif
#ifndef RTK_QP_TRACK_OCC
#define RTK_QP_TRACK_OCC 5  // (a few spilled registers at five: 1-2 % faster than four from 4K on, the same at 1080p)
#endif
template <int KW, bool TRACK, bool TABULATED>
__global__ void __launch_bounds__(kBlock, TRACK ? RTK_QP_TRACK_OCC : RTK_QP_OCC) queue_primary_kernel(DevScene S, DevCamera C, const float4* __restrict__ prim,
                                                                 const float4* __restrict__ prim_ball,
                                                                 const float4* __restrict__ prim_cluster, RenderTarget T,
                                                                 uint32_t tile_base, WavefrontBuffers B) {
  __shared__ Stash stash;
  __shared__ uint32_t s_counts[10];
  const PixelRef px = pixel_ref(T, C, tile_base + blockIdx.x);
  Closest best;
  WaveCount wc;
  wc.primary = lanes_where(px.valid);
  bool graze = false;
  primary_phase(S, C, prim, prim_ball, prim_cluster, px, best, wc, &stash, TRACK, &graze);
  const uint32_t obj = px.valid ? best.obj : 0xffffffffu;
  const bool hit = obj != 0xffffffffu;
  double n1 = 1.0;
  if constexpr (TRACK) {
    bool need = false;
    if (hit) {
      const RTK_RO DevObject& ob = S.objects[obj];
      need = !is_approx_zero(ob.reflectivity) || !is_approx_zero(ob.transparency);
    }
    if (__any(need)) {
      const V3 o = ld3(C.origin), d = get3(stash.dir);
      auto exact = [&](uint32_t at) -> double {
        const RTK_RO SphRecord& rec = S.sph_record[at];
        double t0, t1;
        if (intersect_sphere_record(S, rec, o, d, 0.0, t0, t1)) {
          closest_consider(best, t0, rec.object, true);
          closest_consider(best, t1, rec.object, true);
          if (fabs(t0 - t1) < 0.00001) graze = true;
        }
        return -best.prev_t;
      };
      grid_march(S, need, o, d, true, -best.prev_t, wc, exact);
      if (need) {
        if (graze)
          n1 = literal_n1(S, o, d, obj, best.t);
        else if (best.prev_obj != 0xffffffffu)
          n1 = S.objects[best.prev_obj].refractivity;
      }
    }
  }
  QueueSink<4> sink(S, B, gridDim.x * kBlock, gridDim.x * kBlock, 1u, s_counts);
  V3 p = {0, 0, 0};
  if (hit) {
    const RTK_RO DevObject& ob = S.objects[obj];
    const HitGeom g = lighting_data(ob, ld3(C.origin), get3(stash.dir), best.t);
    put3(stash.over, g.over);
    put3(stash.normal, g.normal);
    p = g.over;
    emit_children(S, ob, g, n1, 1.0, 0, sink);  // children first: nothing of them stays live across the shadow phase
  }
  sink.children_done();
  const bool in_shadow = TABULATED ? shadow_phase_uniform<KW>(S, hit, p, wc) : shadow_phase_bundled<KW>(S, hit, p, wc);
  if (px.valid) {
    if (hit)
      sink.term(shade_hit_stashed(S, launder(obj), in_shadow, &stash), 1.0);
    else
      sink.term(ld3(S.ambient), 1.0);  // trace_inner: miss -> ambient_color
  }
  sink.store(blockIdx.x * kBlock + threadIdx.x);
  flush_counts(counter_shard(B.counters), wc);
}

// Scenes whose lights all have w = 0 or 1; where refractive indices differ, the scene must have its cell grid and the
// camera must stand where a walk can start (grid_walk_applies: within 200,000 cells of the grid; half of that here).
bool queue_primary_bundled_applies(const DevScene& S, const DevCamera& C) {
  if (!(S.lights_uniform || S.lights_bundled) || S.max_depth < 2) return false;
  if (!S.track_n1) return true;
  if (!S.trace_grid) return false;
  for (int a = 0; a < 3; ++a)
    if (!(std::fabs((C.origin[a] - S.tg_lo[a]) * S.tg_inv_h) <= 1.0e5)) return false;
  return true;
}

hipError_t launch_queue_primary_bundled(hipStream_t st, const DevScene& S, const DevCamera& C, const float4* prim,
                                        const float4* prim_ball, const float4* prim_cluster, const RenderTarget& T,
                                        uint32_t tile_base, uint32_t n_tiles, const WavefrontBuffers& B) {
  const dim3 grid(n_tiles), block(kBlock);
  if (S.track_n1 && S.lights_uniform)
    hipLaunchKernelGGL((queue_primary_kernel<4, true, true>), grid, block, 0, st, S, C, prim, prim_ball, prim_cluster, T, tile_base, B);
  else if (S.track_n1)
    hipLaunchKernelGGL((queue_primary_kernel<4, true, false>), grid, block, 0, st, S, C, prim, prim_ball, prim_cluster, T, tile_base, B);
  else if (S.lights_uniform)
    hipLaunchKernelGGL((queue_primary_kernel<4, false, true>), grid, block, 0, st, S, C, prim, prim_ball, prim_cluster, T, tile_base, B);
  else
    hipLaunchKernelGGL((queue_primary_kernel<4, false, false>), grid, block, 0, st, S, C, prim, prim_ball, prim_cluster, T, tile_base, B);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------
hipError_t launch_render_opaque(hipStream_t st, const DevScene& S, const DevCamera& C, const float4* prim,
                                const float4* prim_ball, const float4* prim_cluster, const RenderTarget& T,
                                const OpaqueBuffers& B, DevCounters* counters, hipEvent_t after_rays) {
  drop_stale_error();
  if (T.n_local_tiles == 0) return after_rays ? hipEventRecord(after_rays, st) : hipSuccess;
  const dim3 tiles(T.n_local_tiles), block(kBlock);
  hipError_t e = hipSuccess;
  if (S.max_depth > 0 && (S.lights_uniform || S.lights_bundled)) {
    // everything in one kernel; lights are swept in batches of 4 (the cluster levels cost next
    // to nothing per batch, and 128 VGPRs hold the 4-light filter state without spills).  Lights
    // built from points (w == 1) add cones with their apex at the light.
    const dim3 patches(patch_groups(T.n_local_tiles)), wavefront(64);
    if (S.lights_uniform)
      hipLaunchKernelGGL((render_uniform_kernel<4, true>), patches, wavefront, 0, st, S, C, prim, prim_ball, prim_cluster, T, counters);
    else
      hipLaunchKernelGGL((render_uniform_kernel<4, false>), patches, wavefront, 0, st, S, C, prim, prim_ball, prim_cluster, T, counters);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    return after_rays ? hipEventRecord(after_rays, st) : hipSuccess;
  }
  if (S.max_depth > 0) {
    hipLaunchKernelGGL(primary_kernel, tiles, block, 0, st, S, C, prim, prim_ball, prim_cluster, T, B, counters);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    switch (light_batch_for(S.n_lights)) {
      case 1: hipLaunchKernelGGL((shadow_general_kernel<1>), tiles, block, 0, st, S, C, T, B, counters); break;
      case 2: hipLaunchKernelGGL((shadow_general_kernel<2>), tiles, block, 0, st, S, C, T, B, counters); break;
      case 4: hipLaunchKernelGGL((shadow_general_kernel<4>), tiles, block, 0, st, S, C, T, B, counters); break;
      default: hipLaunchKernelGGL((shadow_general_kernel<8>), tiles, block, 0, st, S, C, T, B, counters); break;
    }
    if ((e = hipGetLastError()) != hipSuccess) return e;
  } else {
    // MAX_DEPTH 0: trace_inner returns the ambient colour without intersecting anything
    if ((e = hipMemsetAsync(B.hit_obj, 0xff, (size_t)T.n_local_tiles * 256u * sizeof(uint32_t), st)) != hipSuccess) return e;
  }
  if (after_rays && (e = hipEventRecord(after_rays, st)) != hipSuccess) return e;
  hipLaunchKernelGGL(shade_kernel, tiles, block, 0, st, S, C, T, B, counters);
  return hipGetLastError();
}

// Whether launch_render_opaque(S, ...) runs the three-kernel path and needs the per-pixel records of B.
bool opaque_needs_records(const DevScene& S) { return !(S.max_depth > 0 && (S.lights_uniform || S.lights_bundled)); }

}  // namespace rtk
