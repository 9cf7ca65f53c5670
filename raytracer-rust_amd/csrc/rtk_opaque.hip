// rtk_opaque.hip -- the render path for scenes without reflective or transparent
// materials (every throughput configuration): three lean kernels instead of one.
//
//   primary_kernel   two pixels per lane; camera-ray sweep over all spheres (FP32 filter,
//                    f64 on candidates), planes exact; writes the hit (object, t)
//   shadow_*_kernel  rebuilds the shading point of each hit, sweeps all spheres for all
//                    lights at once, writes the reference's shadow flag and ray count
//   shade_kernel     rebuilds LightingData, samples the texture, sums Phong over the
//                    lights and stores FP32 RGB + RGBA8 (transposed through LDS)
//
// Nothing of the f64 shading state is live across a sweep, so the sweep kernels fit in 64
// VGPRs and run 8 wavefronts per SIMD.  The intermediate records (13 bytes per pixel) make
// one round trip through HBM; that is noise next to the sweeps (tens of ms at 4K).
//
// Arithmetic of the decisions is unchanged from the single-kernel path (rtk_kernels.hip):
// conservative FP32 filter, then the reference's own f64 formulas with contraction off.
//
// Wavefront-level early-out.  The 64 rays of a wavefront patch (8x8 pixels) are coherent:
// camera rays fill a narrow cone, shadow rays towards a w = 0 light are parallel lines
// through a small ball of shading points.  Each sweep over the object list therefore runs
// in two levels:
//   level 1  the lanes take 64 DIFFERENT spheres each and test them against the patch's
//            bundle (cone / fattened line), about 14 VALU instructions per 64 spheres per
//            bundle; a ballot gives the spheres some ray of the patch might touch;
//   level 2  only for those, the per-ray FP32 filter (one ray per lane, the sphere
//            broadcast), then f64 on the candidates it keeps.
// Both levels are conservative, every (ray, sphere) pair of the object list is covered by
// one of the tests, and the f64 decisions are the same as before.
//
// When every light of the scene has w = 0 (lights loaded from scene files,
// src/loader.rs:35-40) the shadow ray's line direction g_k = L_k/|L_k| is the same for
// every shading point, so the per-(sphere, light) part of the filter is tabulated at scene
// upload: K[j][k] = c_j.g_k - R_j'^2 rho_k.  With P_k = p.g_k per lane,
// h = u.g_k + R^2 rho_k = P_k - K[j][k]: two VALU instructions per (ray, sphere) plus a
// seventh of the shared |p - c|^2 work.  Two pixels per lane keep the LDS reads (32 bytes
// per sphere) below the VALU time.

#include "rtk_shading.h"
#include "rtk_sweeps.h"
#include "rtk_kernels.h"

#pragma clang fp contract(off)

namespace rtk {


// Hides a value from the optimiser so that what is computed from it is recomputed where
// it is used instead of being kept live across a sweep.
__device__ __forceinline__ uint32_t launder(uint32_t v) {
  asm volatile("" : "+v"(v));
  return v;
}

__device__ __forceinline__ bool wave_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }

// Where a thread's pixel sits: local tile k (global tile k*world + rank), slot (lx, ly).
struct PixelRef {
  uint32_t x, y;   // frame coordinates
  uint32_t local;  // index into the per-launch intermediate buffers
  bool in_range;   // the tile exists
  bool valid;      // ... and the pixel lies inside the frame
};
__device__ __forceinline__ PixelRef pixel_ref(const RenderTarget& T, const DevCamera& C, uint32_t k) {
  PixelRef p;
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  const uint32_t lx = (wave << 3) | (lane & 7u), ly = lane >> 3;
  p.in_range = k < T.n_local_tiles;
  const uint32_t tile = k * T.world + T.rank;
  p.x = (tile % T.tiles_x) * 32u + lx;
  p.y = (tile / T.tiles_x) * 8u + ly;
  p.valid = p.in_range && p.x < C.width && p.y < C.height;
  p.local = k * 256u + ly * 32u + lx;
  return p;
}

// ------------------------------------------------------------------------------------
// Bundles.  All fields are wavefront-uniform (they live in SGPRs).

// Cone around the unit camera rays of one patch: axis a, tan and sec of the half-angle.
struct Cone {
  float ax, ay, az, tan_t, sec_t;
};
__device__ __forceinline__ Cone make_cone(float dx, float dy, float dz) {
  Cone c;
  float sx = wave_sum_f(dx), sy = wave_sum_f(dy), sz = wave_sum_f(dz);
  const float inv = 1.0f / sqrtf(sx * sx + sy * sy + sz * sz);
  c.ax = wave_uniform(sx * inv), c.ay = wave_uniform(sy * inv), c.az = wave_uniform(sz * inv);
  const float along = dx * c.ax + dy * c.ay + dz * c.az;
  const float qx = dx - along * c.ax, qy = dy - along * c.ay, qz = dz - along * c.az;
  const float across = sqrtf(qx * qx + qy * qy + qz * qz);
  // the slack covers the FP32 rounding of the directions and of this computation
  const float t = along > 0.0f ? across / along : 3.0e38f;
  c.tan_t = wave_uniform(wave_max_f(t) * (1.0f + 1.0e-5f) + 2.0e-6f);
  c.sec_t = wave_uniform(sqrtf(1.0f + c.tan_t * c.tan_t) * (1.0f + 1.0e-6f));
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
// Kernel 1: camera rays.  Workgroup b owns local tiles 2b' and 2b'+1 (bottom of the frame
// first); each lane traces the same slot of both tiles, each wavefront two 8x8 patches.
__global__ void __launch_bounds__(kBlock, 4) primary_kernel(DevScene S, DevCamera C, const float4* __restrict__ prim,
                                                           const float4* __restrict__ prim_ball,
                                                           const float4* __restrict__ prim_cluster, RenderTarget T,
                                                           OpaqueBuffers B, DevCounters* counters) {
  const uint32_t pairs = (T.n_local_tiles + 1u) / 2u;
  const uint32_t pr = pairs - 1u - blockIdx.x;
  const PixelRef pa = pixel_ref(T, C, 2u * pr), pb = pixel_ref(T, C, 2u * pr + 1u);
  const V3 origin = ld3(C.origin);
  const uint32_t lane = threadIdx.x & 63u;

  Closest best_a, best_b;
  closest_init(best_a);
  closest_init(best_b);
  float ax, ay, az, bx, by, bz;
  unsigned n_exact = 0;
  {
    const V3 da = camera_ray_direction(C, pa.x, pa.y), db = camera_ray_direction(C, pb.x, pb.y);
    ax = (float)da.x, ay = (float)da.y, az = (float)da.z;
    bx = (float)db.x, by = (float)db.y, bz = (float)db.z;
    for (uint32_t ip = 0; ip < S.n_planes; ++ip) {
      const uint32_t obj = S.plane_object[ip];
      double t0, t1;
      if (pa.valid && intersect_object(S.objects[obj], origin, da, 0.0, t0, t1)) closest_consider(best_a, t0, obj, false);
      if (pb.valid && intersect_object(S.objects[obj], origin, db, 0.0, t0, t1)) closest_consider(best_b, t0, obj, false);
    }
  }
  // pixels outside the frame still have well-defined directions; they only widen the cone a little
  const Cone cone_a = make_cone(ax, ay, az), cone_b = make_cone(bx, by, bz);
  const bool live_a = wave_any(pa.valid), live_b = wave_any(pb.valid);

  if (live_a || live_b) {
    for (uint32_t cbase = 0; cbase < S.n_clusters_padded; cbase += kBundle) {
      // level 0: this lane's cluster against both cones
      const float4 cl = prim_cluster[cbase + lane];
      const unsigned long long ca = live_a ? __builtin_amdgcn_ballot_w64(cone_reaches(cone_a, cl)) : 0ull;
      const unsigned long long cb = live_b ? __builtin_amdgcn_ballot_w64(cone_reaches(cone_b, cl)) : 0ull;
      for (unsigned long long cm = ca | cb; cm != 0ull; cm &= cm - 1ull) {
        const uint32_t cbit = (uint32_t)__builtin_ctzll(cm);
        const uint32_t first = (cbase + cbit) * kBundle;
        // level 1: this lane's sphere of the cluster against the cones that reach the cluster
        const float4 ball = prim_ball[first + lane];
        const unsigned long long ma = ((ca >> cbit) & 1ull) ? __builtin_amdgcn_ballot_w64(cone_reaches(cone_a, ball)) : 0ull;
        const unsigned long long mb = ((cb >> cbit) & 1ull) ? __builtin_amdgcn_ballot_w64(cone_reaches(cone_b, ball)) : 0ull;
        // level 2: per-ray filter and exact test for the spheres level 1 kept
        for (unsigned long long m = ma | mb; m != 0ull; m &= m - 1ull) {
          const uint32_t bit = (uint32_t)__builtin_ctzll(m);
          const uint32_t j = first + bit;
          const float4 r = prim[j];
          float ta = r.x * ax;
          ta = __builtin_fmaf(r.y, ay, ta);
          ta = __builtin_fmaf(r.z, az, ta);
          float tb = r.x * bx;
          tb = __builtin_fmaf(r.y, by, tb);
          tb = __builtin_fmaf(r.z, bz, tb);
          const bool pass_a = pa.valid && ((ma >> bit) & 1ull) && __builtin_fmaf(ta, ta, -r.w) >= 0.0f;
          const bool pass_b = pb.valid && ((mb >> bit) & 1ull) && __builtin_fmaf(tb, tb, -r.w) >= 0.0f;
          if (wave_any(pass_a || pass_b)) {
            const uint32_t obj = S.sph_object[j];
            double t0, t1;
            if (pass_a) {
              ++n_exact;
              const V3 d = camera_ray_direction(C, launder(pa.x), pa.y);
              if (intersect_object(S.objects[obj], origin, d, 0.0, t0, t1)) {
                closest_consider(best_a, t0, obj, false);
                closest_consider(best_a, t1, obj, false);
              }
            }
            if (pass_b) {
              ++n_exact;
              const V3 d = camera_ray_direction(C, launder(pb.x), pb.y);
              if (intersect_object(S.objects[obj], origin, d, 0.0, t0, t1)) {
                closest_consider(best_b, t0, obj, false);
                closest_consider(best_b, t1, obj, false);
              }
            }
          }
        }
      }
    }
  }
  if (pa.in_range) {
    B.hit_obj[pa.local] = pa.valid ? best_a.obj : 0xffffffffu;
    B.hit_t[pa.local] = best_a.t;
  }
  if (pb.in_range) {
    B.hit_obj[pb.local] = pb.valid ? best_b.obj : 0xffffffffu;
    B.hit_t[pb.local] = best_b.t;
  }
  unsigned n_primary = wave_sum((pa.valid ? 1u : 0u) + (pb.valid ? 1u : 0u));
  n_exact = wave_sum(n_exact);
  if (lane == 0) {
    atomicAdd(&counter_shard(counters)->rays_primary, (unsigned long long)n_primary);
    atomicAdd(&counter_shard(counters)->exact_tests, (unsigned long long)n_exact);
  }
}

// The shading point of a recorded hit: `over_position` of LightingData (src/scene/lighting.rs:50).
__device__ __forceinline__ V3 over_point(const DevScene& S, const DevCamera& C, const PixelRef& p, uint32_t obj, double t) {
  const V3 d = camera_ray_direction(C, p.x, p.y);
  return lighting_data(S.objects[obj], ld3(C.origin), d, t).over;
}

// One shadow ray against one object, exactly: the per-light body of `Scene::is_shadowed`
// (src/scene.rs:208-225).
__device__ __forceinline__ bool shadow_ray_blocked(const DevObject& ob, const DevLight& L, V3 p) {
  V3 lv = ld3(L.pos) - p;
  const double lw = L.pos[3] - 1.0;
  const double m = sqrt(lv.x * lv.x + lv.y * lv.y + lv.z * lv.z + lw * lw);
  const V3 d = {lv.x / m, lv.y / m, lv.z / m};
  double t0, t1;
  const int n = intersect_object(ob, p, d, lw / m, t0, t1);
  return (n >= 1 && t0 > 0.0 && t0 < m) || (n == 2 && t1 > 0.0 && t1 < m);
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
template <int KW>
__device__ __forceinline__ void park(PixelFilter<KW>& f) {
  f.x = f.y = f.z = 0.0f;
#pragma unroll
  for (int k = 0; k < KW; ++k) f.P[k] = 0.0f;
  f.thrp = 3.0e38f;
}
template <int KW>
__device__ __forceinline__ void arm(PixelFilter<KW>& f, const DevScene& S, V3 p, uint32_t kb, uint32_t nb) {
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
  const double pn2 = dot3(pc, pc), pn = sqrt(pn2);
  const double B = (double)S.r2_max * (double)rho_max;
  const double dh = (double)kEps32 * (6.0 * pn + 2.0 * (double)S.c_max + 2.0 * B) * 1.01;
  const double hmax = pn + (double)S.c_max + B;
  const double cm2 = (double)S.c_max * (double)S.c_max;
  const double extra = 2.0 * hmax * dh + dh * dh + 6.0 * (double)kEps32 * (pn2 + cm2 + (double)S.r2_max) * 1.01;
  const double thr = (double)filter_threshold(S, pc, rho_max) - extra * 1.001;
  const double pp = (double)f.x * f.x + (double)f.y * f.y + (double)f.z * f.z;
  const double t = (double)kKappa * pp + thr;
  float tf = (float)t;
  if ((double)tf > t) tf = nextafterf(tf, -INFINITY);
  f.thrp = nextafterf(tf, -INFINITY);
}
constexpr float kTwoKappa = 2.0f * kKappa;

// Ball around the live shading points of one patch (recentred FP32 coordinates).
struct Ball {
  float mx, my, mz, radius;
  bool any;
};
__device__ __forceinline__ Ball make_ball(bool live, float x, float y, float z) {
  Ball b;
  const float n = wave_sum_f(live ? 1.0f : 0.0f);
  b.any = n > 0.0f;
  const float inv = 1.0f / fmaxf(n, 1.0f);
  b.mx = wave_uniform(wave_sum_f(live ? x : 0.0f) * inv);
  b.my = wave_uniform(wave_sum_f(live ? y : 0.0f) * inv);
  b.mz = wave_uniform(wave_sum_f(live ? z : 0.0f) * inv);
  const float ex = x - b.mx, ey = y - b.my, ez = z - b.mz;
  const float far = live ? sqrtf(ex * ex + ey * ey + ez * ez) : 0.0f;
  b.radius = wave_uniform(wave_max_f(far) * (1.0f + 1.0e-5f));
  return b;
}

// ------------------------------------------------------------------------------------
// Kernel 2a: shadow rays when every light has w = 0.  KW = lights per batch (4 or 8).
//
// Level 1 for one light: the patch's shadow lines are parallel to g and pass through the
// ball {m, rho_w}.  A line through p can touch sphere {c, R} only if its distance from c is
// at most R_e = R sqrt(1 + 2 |rho| (-s) + R^2 rho^2), s = (p - c).g (the reference's growing
// ball, see PixelFilter).  Over the ball of points: distance(c, line through m) <=
// rho_w + R + R |rho| max(rho_w - s_m, 0) + R^3 rho^2 / 2.
template <int KW>
__global__ void __launch_bounds__(kBlock, KW == 4 ? 4 : 3) shadow_uniform_kernel(DevScene S, DevCamera C, RenderTarget T,
                                                                                OpaqueBuffers B, DevCounters* counters) {
  constexpr int KV = KW / 4;  // float4 per table row
  const uint32_t pairs = (T.n_local_tiles + 1u) / 2u;
  const uint32_t pr = pairs - 1u - blockIdx.x;
  const PixelRef pa = pixel_ref(T, C, 2u * pr), pb = pixel_ref(T, C, 2u * pr + 1u);
  const uint32_t lane = threadIdx.x & 63u;

  uint32_t obj_a = 0xffffffffu, obj_b = 0xffffffffu;
  if (pa.valid) obj_a = B.hit_obj[pa.local];
  if (pb.valid) obj_b = B.hit_obj[pb.local];
  bool need_a = obj_a != 0xffffffffu, need_b = obj_b != 0xffffffffu;
  // The f64 shading points are needed only by the exact tests; they are parked in HBM and
  // re-read there so that they do not occupy 12 VGPRs across the sweep.
  const size_t stride = (size_t)T.n_local_tiles * 256u;
  if (need_a) {
    const V3 p = over_point(S, C, pa, obj_a, B.hit_t[pa.local]);
    B.over[pa.local] = p.x, B.over[stride + pa.local] = p.y, B.over[2 * stride + pa.local] = p.z;
  }
  if (need_b) {
    const V3 p = over_point(S, C, pb, obj_b, B.hit_t[pb.local]);
    B.over[pb.local] = p.x, B.over[stride + pb.local] = p.y, B.over[2 * stride + pb.local] = p.z;
  }
  auto load_p = [&](const PixelRef& px) {
    const uint32_t i = launder(px.local);
    return V3{B.over[i], B.over[stride + i], B.over[2 * stride + i]};
  };

  unsigned rays_a = 0, rays_b = 0, n_exact = 0;
  bool shadow_a = false, shadow_b = false;

#if RTK_EXP == 1
  need_a = need_b = false;
#endif
  for (uint32_t kb = 0, batch = 0; kb < S.n_lights; kb += KW, ++batch) {
    if (!wave_any(need_a || need_b)) break;
    const uint32_t nb = min((uint32_t)KW, S.n_lights - kb);
    uint32_t occl_a = 0, occl_b = 0;
    PixelFilter<KW> fa, fb;
    park(fa);
    park(fb);
    bool live_a = false, live_b = false;
    if (need_a) {
      const V3 p = load_p(pa);
      for (uint32_t r = 0; r < nb; ++r)
        for (uint32_t ip = 0; ip < S.n_planes; ++ip)
          if (plane_may_block(S.objects[S.plane_object[ip]], S.lights[kb + r], p) &&
              shadow_ray_blocked(S.objects[S.plane_object[ip]], S.lights[kb + r], p))
            occl_a |= 1u << r;
      live_a = !(occl_a & 1u);
      if (live_a) arm(fa, S, p, kb, nb);
    }
    if (need_b) {
      const V3 p = load_p(pb);
      for (uint32_t r = 0; r < nb; ++r)
        for (uint32_t ip = 0; ip < S.n_planes; ++ip)
          if (plane_may_block(S.objects[S.plane_object[ip]], S.lights[kb + r], p) &&
              shadow_ray_blocked(S.objects[S.plane_object[ip]], S.lights[kb + r], p))
            occl_b |= 1u << r;
      live_b = !(occl_b & 1u);
      if (live_b) arm(fb, S, p, kb, nb);
    }
    const float4* __restrict__ ktab = S.k_table + (size_t)batch * S.n_spheres_padded * KV;

    // bundles of the two patches; the slack covers the FP32 rounding of points and centres
    const Ball ball_a = make_ball(live_a, fa.x, fa.y, fa.z), ball_b = make_ball(live_b, fb.x, fb.y, fb.z);
    float lgx[KW], lgy[KW], lgz[KW], lrho[KW];
    float grow = 0.0f;
#pragma unroll
    for (int k = 0; k < KW; ++k) {
      const float4 lf = S.light_filter[kb + min((uint32_t)k, nb - 1u)];
      lgx[k] = lf.x, lgy[k] = lf.y, lgz[k] = lf.z, lrho[k] = fabsf(lf.w) * 1.0001f;
      grow = fmaxf(grow, 0.5f * S.r_max * S.r2_max * lf.w * lf.w);
    }
    const float pad_a = ball_a.radius + grow * 1.01f +
                        16.0f * kEps32 * (fabsf(ball_a.mx) + fabsf(ball_a.my) + fabsf(ball_a.mz) + ball_a.radius + S.c_max);
    const float pad_b = ball_b.radius + grow * 1.01f +
                        16.0f * kEps32 * (fabsf(ball_b.mx) + fabsf(ball_b.my) + fabsf(ball_b.mz) + ball_b.radius + S.c_max);

    // rec = {centre, radius} of a sphere (t_extra = 0, growth radius = its own) or of a
    // cluster (t_extra = its radius, growth radius = the scene's largest sphere radius)
    auto lines_reach = [&](const Ball& bl, float pad, float4 rec, float t_extra, float r_grow) {
      const float ux = bl.mx - rec.x, uy = bl.my - rec.y, uz = bl.mz - rec.z;
      const float reach = rec.w + pad;  // R + rho_w + slack
      const float tb = bl.radius + t_extra;
      bool pass = false;
#pragma unroll
      for (int k = 0; k < KW; ++k) {
        const float s = __builtin_fmaf(uz, lgz[k], __builtin_fmaf(uy, lgy[k], ux * lgx[k]));
        const float qx = __builtin_fmaf(-s, lgx[k], ux), qy = __builtin_fmaf(-s, lgy[k], uy), qz = __builtin_fmaf(-s, lgz[k], uz);
        const float d2 = __builtin_fmaf(qz, qz, __builtin_fmaf(qy, qy, qx * qx));
        const float t = fmaxf(tb - s, 0.0f) * (1.0f + 8.0f * kEps32);
        const float rr = __builtin_fmaf(t, r_grow * lrho[k], reach) * (1.0f + 4.0f * kEps32);
        pass |= rr >= 0.0f && __builtin_fmaf(rr, rr, -d2) >= 0.0f;
      }
      return pass;
    };

    bool wave_live = wave_any(live_a || live_b);
#if RTK_EXP == 2
    wave_live = false;
#endif
    for (uint32_t cbase = 0; wave_live && cbase < S.n_clusters_padded; cbase += kBundle) {
      // level 0: this lane's cluster against the line bundles of both patches
      const float4 cl = S.cluster_ball[cbase + lane];
      const unsigned long long ca =
          wave_any(live_a) ? __builtin_amdgcn_ballot_w64(lines_reach(ball_a, pad_a, cl, cl.w, S.r_max)) : 0ull;
      const unsigned long long cb =
          wave_any(live_b) ? __builtin_amdgcn_ballot_w64(lines_reach(ball_b, pad_b, cl, cl.w, S.r_max)) : 0ull;
      for (unsigned long long cm = ca | cb; wave_live && cm != 0ull; cm &= cm - 1ull) {
        const uint32_t cbit = (uint32_t)__builtin_ctzll(cm);
        const uint32_t first = (cbase + cbit) * kBundle;
        // level 1: this lane's sphere of the cluster
        const float4 rec = S.sph_ball[first + lane];
        const unsigned long long ma =
            ((ca >> cbit) & 1ull) && wave_any(live_a) ? __builtin_amdgcn_ballot_w64(lines_reach(ball_a, pad_a, rec, 0.0f, rec.w)) : 0ull;
        const unsigned long long mb =
            ((cb >> cbit) & 1ull) && wave_any(live_b) ? __builtin_amdgcn_ballot_w64(lines_reach(ball_b, pad_b, rec, 0.0f, rec.w)) : 0ull;
        bool changed = false;
        // level 2: per-ray tabulated filter, then the exact test
        for (unsigned long long m = ma | mb; m != 0ull; m &= m - 1ull) {
          const uint32_t bit = (uint32_t)__builtin_ctzll(m);
          const uint32_t j = first + bit;
          const float4 c = S.sph_tab[j];
          const float pca = __builtin_fmaf(fa.z, c.z, __builtin_fmaf(fa.y, c.y, fa.x * c.x));
          const float pcb = __builtin_fmaf(fb.z, c.z, __builtin_fmaf(fb.y, c.y, fb.x * c.x));
          const float ea = __builtin_fmaf(pca, kTwoKappa, -c.w), eb = __builtin_fmaf(pcb, kTwoKappa, -c.w);
          uint32_t mask_a = 0, mask_b = 0;
#pragma unroll
          for (int v = 0; v < KV; ++v) {
            const float4 row = ktab[(size_t)j * KV + v];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int k = v * 4 + e;
              const float kv = (&row.x)[e];
              const float ha = fa.P[k] - kv, hb = fb.P[k] - kv;
              if (__builtin_fmaf(ha, ha, ea) >= fa.thrp) mask_a |= 1u << k;
              if (__builtin_fmaf(hb, hb, eb) >= fb.thrp) mask_b |= 1u << k;
            }
          }
          if (!((ma >> bit) & 1ull)) mask_a = 0;
          if (!((mb >> bit) & 1ull)) mask_b = 0;
          mask_a &= ~occl_a;
          mask_b &= ~occl_b;
          if (wave_any((mask_a | mask_b) != 0)) {
            const DevObject& ob = S.objects[S.sph_object[j]];
            if (mask_a) {
              const V3 p = load_p(pa);
              for (uint32_t r = 0; r < nb; ++r)
                if ((mask_a >> r) & 1u) {
                  ++n_exact;
                  if (shadow_ray_blocked(ob, S.lights[kb + r], p)) occl_a |= 1u << r;
                }
            }
            if (mask_b) {
              const V3 p = load_p(pb);
              for (uint32_t r = 0; r < nb; ++r)
                if ((mask_b >> r) & 1u) {
                  ++n_exact;
                  if (shadow_ray_blocked(ob, S.lights[kb + r], p)) occl_b |= 1u << r;
                }
            }
            changed = true;
          }
        }
        if (changed) {
          // a pixel whose first light is blocked needs nothing more: park it
          if (live_a && (occl_a & 1u)) {
            live_a = false;
            park(fa);
          }
          if (live_b && (occl_b & 1u)) {
            live_b = false;
            park(fb);
          }
          wave_live = wave_any(live_a || live_b);
        }
      }
    }
    const uint32_t lights_mask = (nb >= 32u) ? 0xffffffffu : ((1u << nb) - 1u);
    if (need_a) {
      occl_a &= lights_mask;
      if (occl_a) {
        rays_a += (unsigned)__builtin_ctz(occl_a) + 1u;
        shadow_a = true;
        need_a = false;
      } else {
        rays_a += nb;
      }
    }
    if (need_b) {
      occl_b &= lights_mask;
      if (occl_b) {
        rays_b += (unsigned)__builtin_ctz(occl_b) + 1u;
        shadow_b = true;
        need_b = false;
      } else {
        rays_b += nb;
      }
    }
  }
  if (pa.in_range) B.shadow[pa.local] = (uint8_t)((shadow_a ? 0x80u : 0u) | rays_a);
  if (pb.in_range) B.shadow[pb.local] = (uint8_t)((shadow_b ? 0x80u : 0u) | rays_b);
  n_exact = wave_sum(n_exact);
  if (lane == 0) atomicAdd(&counter_shard(counters)->exact_tests, (unsigned long long)n_exact);
}

// Kernel 2b: shadow rays for arbitrary lights (some w != 0): one pixel per lane, per-lane
// line directions, LB lights fused.
template <int LB>
__global__ void __launch_bounds__(kBlock, 4) shadow_general_kernel(DevScene S, DevCamera C, RenderTarget T, OpaqueBuffers B,
                                                                  DevCounters* counters) {
  __shared__ float4 s_buf[kChunk];
  const uint32_t k = T.n_local_tiles - 1u - blockIdx.x;
  const PixelRef px = pixel_ref(T, C, k);
  uint32_t obj = 0xffffffffu;
  if (px.valid) obj = B.hit_obj[px.local];
  const bool hit = obj != 0xffffffffu;
  V3 p = {0, 0, 0};
  if (hit) p = over_point(S, C, px, obj, B.hit_t[px.local]);
  unsigned rays = 0, n_exact = 0;
  const bool in_shadow = shadow_sweeps<LB>(S, s_buf, hit, p, rays, n_exact);
  B.shadow[px.local] = (uint8_t)((in_shadow ? 0x80u : 0u) | rays);
  n_exact = wave_sum(n_exact);
  if ((threadIdx.x & 63u) == 0) atomicAdd(&counter_shard(counters)->exact_tests, (unsigned long long)n_exact);
}

// ------------------------------------------------------------------------------------
// Kernel 3: `apply_lighting` (src/scene.rs:172-205) for opaque materials and the store.
__global__ void __launch_bounds__(kBlock) shade_kernel(DevScene S, DevCamera C, RenderTarget T, OpaqueBuffers B,
                                                      DevCounters* counters) {
  __shared__ float s_rgb[768];
  __shared__ uint32_t s_px[256];
  const uint32_t k = blockIdx.x;
  const PixelRef px = pixel_ref(T, C, k);
  const V3 ambient = ld3(S.ambient);
  V3 color = {0, 0, 0};
  unsigned n_shadow = 0;
  if (px.valid) {
    color = ambient;  // a miss, or trace_inner at depth >= MAX_DEPTH, returns ambient_color
    const uint32_t obj = B.hit_obj[px.local];
    if (obj != 0xffffffffu) {
      const uint8_t sh = B.shadow[px.local];
      const bool in_shadow = (sh & 0x80u) != 0;
      n_shadow = sh & 0x7fu;
      const DevObject& ob = S.objects[obj];
      const V3 d = camera_ray_direction(C, px.x, px.y);
      const HitGeom g = lighting_data(ob, ld3(C.origin), d, B.hit_t[px.local]);
      const V3 tex = sample_texture(ob, g.objp);
      V3 surface = ambient;
      for (uint32_t l = 0; l < S.n_lights; ++l)
        surface = surface + phong_lighting(S.lights[l], ob, tex, g.over, g.eye, g.normal, in_shadow);
      color = surface;  // the reflected and refracted parts are BLACK for these materials
    }
  }
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  const uint32_t lx = (wave << 3) | (lane & 7u), ly = lane >> 3;
  const uint32_t slot = ly * 32u + lx;
  s_rgb[slot * 3 + 0] = (float)color.x;
  s_rgb[slot * 3 + 1] = (float)color.y;
  s_rgb[slot * 3 + 2] = (float)color.z;
  s_px[slot] = cast_u8(color.x) | (cast_u8(color.y) << 8) | (cast_u8(color.z) << 16) | 0xff000000u;
  __syncthreads();
  {
    const uint32_t tile = k * T.world + T.rank;
    const uint32_t tx = tile % T.tiles_x, ty = tile / T.tiles_x;
    const uint32_t rx = threadIdx.x & 31u, ry = threadIdx.x >> 5;
    const uint32_t gx = tx * 32u + rx, gy = ty * 8u + ry;
    if (T.rgba8) {
      uint32_t* out = reinterpret_cast<uint32_t*>(T.rgba8);
      if (T.compact)
        out[(size_t)k * 256u + threadIdx.x] = s_px[threadIdx.x];
      else if (gx < C.width && gy < C.height)
        out[(size_t)gx + (size_t)gy * C.width] = s_px[threadIdx.x];
    }
    if (T.rgb) {
#pragma unroll
      for (uint32_t pass = 0; pass < 3; ++pass) {
        const uint32_t e = threadIdx.x + pass * 256u;
        const uint32_t row = e / 96u, col = e % 96u;
        if (T.compact) {
          T.rgb[(size_t)k * 768u + e] = s_rgb[e];
        } else {
          const uint32_t fx = tx * 32u + col / 3u, fy = ty * 8u + row;
          if (fx < C.width && fy < C.height) T.rgb[((size_t)fx + (size_t)fy * C.width) * 3u + col % 3u] = s_rgb[e];
        }
      }
    }
  }
  n_shadow = wave_sum(n_shadow);
  if (lane == 0) atomicAdd(&counter_shard(counters)->rays_shadow, (unsigned long long)n_shadow);
}

// ------------------------------------------------------------------------------------
hipError_t launch_render_opaque(hipStream_t st, const DevScene& S, const DevCamera& C, const float4* prim,
                                const float4* prim_ball, const float4* prim_cluster, const RenderTarget& T,
                                const OpaqueBuffers& B, DevCounters* counters) {
  if (T.n_local_tiles == 0) return hipSuccess;
  const dim3 pairs((T.n_local_tiles + 1u) / 2u), tiles(T.n_local_tiles), block(kBlock);
  if (S.max_depth > 0) {
    hipLaunchKernelGGL(primary_kernel, pairs, block, 0, st, S, C, prim, prim_ball, prim_cluster, T, B, counters);
    if (S.lights_uniform) {
      if (S.k_width == 4)
        hipLaunchKernelGGL((shadow_uniform_kernel<4>), pairs, block, 0, st, S, C, T, B, counters);
      else
        hipLaunchKernelGGL((shadow_uniform_kernel<8>), pairs, block, 0, st, S, C, T, B, counters);
    } else {
      switch (light_batch_for(S.n_lights)) {
        case 1: hipLaunchKernelGGL((shadow_general_kernel<1>), tiles, block, 0, st, S, C, T, B, counters); break;
        case 2: hipLaunchKernelGGL((shadow_general_kernel<2>), tiles, block, 0, st, S, C, T, B, counters); break;
        case 4: hipLaunchKernelGGL((shadow_general_kernel<4>), tiles, block, 0, st, S, C, T, B, counters); break;
        default: hipLaunchKernelGGL((shadow_general_kernel<8>), tiles, block, 0, st, S, C, T, B, counters); break;
      }
    }
  } else {
    // MAX_DEPTH 0: trace_inner returns the ambient colour without intersecting anything
    hipError_t e = hipMemsetAsync(B.hit_obj, 0xff, (size_t)T.n_local_tiles * 256u * sizeof(uint32_t), st);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(shade_kernel, tiles, block, 0, st, S, C, T, B, counters);
  return hipGetLastError();
}

}  // namespace rtk
