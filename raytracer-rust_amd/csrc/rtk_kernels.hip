// rtk_kernels.hip -- gfx950 kernels of the ray-trace core.
//
// One thread traces one pixel (reference loop: src/scene/cameras.rs:72-85); one
// 256-thread workgroup owns a 32x8 pixel tile, its four wavefronts own 8x8 sub-tiles so
// that the 64 rays of a wavefront stay coherent.
//
// Every "ray" (one `Scene::intersect` call of the reference, src/scene.rs:156-169) is a
// brute-force sweep over ALL objects:
//   1. planes are few: each is intersected exactly in f64;
//   2. spheres go through a conservative FP32 filter fed from LDS.  The filter answers
//      "can the ray's line touch the sphere's bounding ball at all?" with enough slack to
//      cover every rounding error, so it never rejects a real hit.  Eight records are
//      tested per wavefront branch; only when some lane keeps a candidate does the
//      wavefront leave the fast loop;
//   3. kept candidates are re-evaluated in f64 with the reference's own formulas
//      (object-space ray, unit sphere quadratic), so every decision -- closest hit,
//      tie-break, shadow -- is taken on f64 values like the reference takes it.
// Shading (LightingData, Phong, reflection, refraction, Schlick) is O(1) per ray and
// runs in f64 in registers; the result is narrowed to FP32 / RGBA8 at the store.
//
// VALU-bound by design: no MFMA (nothing here is a dense contraction), HBM traffic is
// the framebuffer store plus one pass over the 16-byte records per workgroup sweep.

#include "rtk_device.h"
#include "rtk_kernels.h"

#include <float.h>

// The f64 code below restates the reference's arithmetic operation for operation.  rustc
// never fuses a multiply with an add, so contraction stays off here as well: the exact
// tests then take bit-identical decisions even in degenerate cases (tangent rays, points
// lying on a surface).  The FP32 filter spells out its fused multiply-adds explicitly.
#pragma clang fp contract(off)

namespace rtk {

// ------------------------------------------------------------------------------------
// small f64 vector helpers (xyz only; w is carried explicitly where the reference's
// 4-component arithmetic makes it matter)
struct V3 {
  double x, y, z;
};
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
__device__ __forceinline__ V3 operator*(V3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ V3 operator*(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
__device__ __forceinline__ double dot3(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 ld3(const double* p) { return {p[0], p[1], p[2]}; }

// rows 0..2 of an affine 4x4 times (v, w): the 4-term rows of src/maths/vectors.rs:205-223
__device__ __forceinline__ V3 xform(const double* m, V3 v, double w) {
  return {m[0] * v.x + m[1] * v.y + m[2] * v.z + m[3] * w, m[4] * v.x + m[5] * v.y + m[6] * v.z + m[7] * w,
          m[8] * v.x + m[9] * v.y + m[10] * v.z + m[11] * w};
}

// ------------------------------------------------------------------------------------
// Exact f64 intersection of one object: `SceneNode::intersect` (src/scene.rs:75-84) with
// `Sphere::intersect` (src/scene/shapes/spheres.rs:18-35) or `Plane::intersect`
// (src/scene/shapes/planes.rs:22-28).  The ray origin has w = 1; dw is the direction's w
// (0 for camera and secondary rays, non-zero for shadow rays towards a w != 1 light).
__device__ __forceinline__ int intersect_object(const DevObject& ob, V3 o, V3 d, double dw, double& t0, double& t1) {
  const double* m = ob.inv;
  if (ob.kind == 1u) {
    double oy = m[4] * o.x + m[5] * o.y + m[6] * o.z + m[7];
    double dy = m[4] * d.x + m[5] * d.y + m[6] * d.z + m[7] * dw;
    if (fabs(dy) < 0.00001) return 0;
    t0 = -oy / dy;
    return 1;
  }
  V3 oo = xform(m, o, 1.0);
  V3 dd = xform(m, d, dw);
  // direction.w stays dw (row 3 = 0 0 0 1) and takes part in `a`; sphere_to_ray.w = 1 - 1 = 0
  double a = dd.x * dd.x + dd.y * dd.y + dd.z * dd.z + dw * dw;
  double b = 2.0 * (oo.x * dd.x + oo.y * dd.y + oo.z * dd.z);
  double c = (oo.x * oo.x + oo.y * oo.y + oo.z * oo.z) - 1.0;
  double disc = b * b - 4.0 * a * c;
  if (disc >= 0.0) {
    double s = sqrt(disc);
    t0 = (-b - s) / (2.0 * a);
    t1 = (-b + s) / (2.0 * a);
    return 2;
  }
  return 0;
}

// Closest-hit bookkeeping.  The reference sorts all hits stably by distance and takes the
// first strictly smallest positive one (src/scene.rs:164-166, 325-340): ties go to the
// lowest object index.  `prev` is the entry just before the winner in that sorted list
// (largest t <= 0, ties to the LAST in list order); its material gives n1
// (src/scene/lighting.rs:81-116).
struct Closest {
  double t;
  uint32_t obj;
  double prev_t;
  uint32_t prev_obj;
};
__device__ __forceinline__ void closest_init(Closest& c) {
  c.t = DBL_MAX;
  c.obj = 0xffffffffu;
  c.prev_t = -DBL_MAX;
  c.prev_obj = 0xffffffffu;
}
__device__ __forceinline__ void closest_consider(Closest& c, double t, uint32_t obj, bool track_prev) {
  if (t > 0.0) {
    if (t < c.t || (t == c.t && obj < c.obj)) {
      c.t = t;
      c.obj = obj;
    }
  } else if (track_prev && t <= 0.0) {
    if (t > c.prev_t || (t == c.prev_t && obj >= c.prev_obj)) {
      c.prev_t = t;
      c.prev_obj = obj;
    }
  }
}

// ------------------------------------------------------------------------------------
// LDS staging of one chunk of 16-byte filter records (all 256 threads take part).
__device__ __forceinline__ void stage_chunk(float4* s_buf, const float4* __restrict__ src, uint32_t base, uint32_t count) {
  __syncthreads();  // previous chunk fully consumed
  for (uint32_t i = threadIdx.x; i < count; i += kBlock) s_buf[i] = src[base + i];
  __syncthreads();
}

// ------------------------------------------------------------------------------------
// Primary rays: all rays of a frame leave from the camera origin, so the per-sphere part
// of the test is hoisted into the record {L = c - o, CC = |L|^2 (1 - slack) - R'^2}.
// With a unit direction d the line can touch the bounding ball only if (L.d)^2 >= CC:
// 3 multiply-adds, one fused square-and-subtract and a share of a max per test.
template <typename ExactFn>
__device__ __forceinline__ void sweep_primary(const DevScene& S, const float4* __restrict__ prim, float4* s_buf, bool active,
                                              float dx, float dy, float dz, ExactFn&& exact) {
  for (uint32_t base = 0; base < S.n_spheres_padded; base += kChunk) {
    const uint32_t count = min((uint32_t)kChunk, S.n_spheres_padded - base);
    stage_chunk(s_buf, prim, base, count);
    if (!__any(active)) continue;  // wavefront has no live ray: only help with staging
    for (uint32_t j = 0; j < count; j += kGroup) {
      float4 q[kGroup];
#pragma unroll
      for (int g = 0; g < kGroup; ++g) q[g] = s_buf[j + g];
      float m = -1.0f;
#pragma unroll
      for (int g = 0; g < kGroup; ++g) {
        float t = q[g].x * dx;
        t = __builtin_fmaf(q[g].y, dy, t);
        t = __builtin_fmaf(q[g].z, dz, t);
        m = fmaxf(m, __builtin_fmaf(t, t, -q[g].w));
      }
      if (__any(active && m >= 0.0f)) {
        // leave the fast loop: find which of the eight records kept a candidate
#pragma unroll 1
        for (int g = 0; g < kGroup; ++g) {
          float4 r = s_buf[j + g];
          float t = r.x * dx;
          t = __builtin_fmaf(r.y, dy, t);
          t = __builtin_fmaf(r.z, dz, t);
          bool pass = active && (__builtin_fmaf(t, t, -r.w) >= 0.0f);
          if (__any(pass)) {
            uint32_t obj = S.sph_object[base + j + g];
            if (pass) exact(obj);
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------
// General filter for NR rays that share the origin p (shadow rays of one shading point
// towards NR lights, or a single secondary ray).  For a ray with unit line direction g
// and "growth" rho (see make_filter_ray) a real intersection needs
//     (u.g + R^2 rho)^2 - (|u|^2 - R^2) >= 0,   u = p - c.
// The FP32 evaluation keeps a candidate when the computed value is >= thr (<= 0), where
// kappa (relative, folded into the multiply-add) and thr (absolute, per lane) cover the
// rounding of inputs and arithmetic.
constexpr float kKappa = 1.0f - 24.0f * kEps32;

template <int NR, typename ExactFn, typename DoneFn>
__device__ __forceinline__ void sweep_filter(const DevScene& S, float4* s_buf, bool& active, float px, float py, float pz,
                                             const float (&gx)[NR], const float (&gy)[NR], const float (&gz)[NR],
                                             const float (&rho)[NR], float thr, ExactFn&& exact, DoneFn&& done) {
  bool wave_live = __any(active);
  for (uint32_t base = 0; base < S.n_spheres_padded; base += kChunk) {
    const uint32_t count = min((uint32_t)kChunk, S.n_spheres_padded - base);
    stage_chunk(s_buf, S.sph_filter, base, count);
    if (!wave_live) continue;
    for (uint32_t j = 0; j < count; j += kGroup) {
      float4 q[kGroup];
#pragma unroll
      for (int g = 0; g < kGroup; ++g) q[g] = s_buf[j + g];
      float m = -3.0e38f;
#pragma unroll
      for (int g = 0; g < kGroup; ++g) {
        float ux = px - q[g].x, uy = py - q[g].y, uz = pz - q[g].z;
        float uu = __builtin_fmaf(uz, uz, __builtin_fmaf(uy, uy, ux * ux));
        float cf = __builtin_fmaf(uu, kKappa, -q[g].w);
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          float h = __builtin_fmaf(uz, gz[r], __builtin_fmaf(uy, gy[r], __builtin_fmaf(ux, gx[r], q[g].w * rho[r])));
          m = fmaxf(m, __builtin_fmaf(h, h, -cf));
        }
      }
      if (__any(active && m >= thr)) {
#pragma unroll 1
        for (int g = 0; g < kGroup; ++g) {
          float4 c = s_buf[j + g];
          float ux = px - c.x, uy = py - c.y, uz = pz - c.z;
          float uu = __builtin_fmaf(uz, uz, __builtin_fmaf(uy, uy, ux * ux));
          float cf = __builtin_fmaf(uu, kKappa, -c.w);
          uint32_t mask = 0;
#pragma unroll
          for (int r = 0; r < NR; ++r) {
            float h = __builtin_fmaf(uz, gz[r], __builtin_fmaf(uy, gy[r], __builtin_fmaf(ux, gx[r], c.w * rho[r])));
            if (__builtin_fmaf(h, h, -cf) >= thr) mask |= 1u << r;
          }
          if (!active) mask = 0;
          if (__any(mask != 0)) {
            uint32_t obj = S.sph_object[base + j + g];
            if (mask) exact(obj, mask);
          }
        }
        // lanes that need no further test drop out; the wavefront stops once all have
        if (active && done()) active = false;
        wave_live = __any(active);
        if (!wave_live) break;
      }
    }
  }
}

// Per-lane set-up of one filter ray.  For a light at L (w = L.w) seen from p the
// reference's shadow ray has direction (L - p)/m with a w component (L.w - 1)/m that the
// object's inverse transform turns into a per-object shift (src/scene.rs:208-225,
// src/maths/rays.rs:31-41).  In the sphere's frame the ray runs along
// G = L.xyz - L.w p through u = p - c, and a root exists only if
// (u.G^ + R^2 rho)^2 >= |u|^2 - R^2 with G^ = G/|G|, rho = (L.w - 1)/|G|.
// A plain ray (direction d, w = 0) is the case G = d, rho = 0.
__device__ __forceinline__ void make_filter_ray(V3 G, double omega, float& gx, float& gy, float& gz, float& rho) {
  double inv = 1.0 / sqrt(dot3(G, G));
  gx = (float)(G.x * inv);
  gy = (float)(G.y * inv);
  gz = (float)(G.z * inv);
  rho = (float)(omega * inv);
}

// Absolute slack of the filter for one lane: covers the FP32 rounding of the recentred
// origin (treated as growing every bounding ball by dp), and the terms of the rounding
// analysis that do not scale with |u|^2.  Returned as the (non-positive) threshold.
__device__ __forceinline__ float filter_threshold(const DevScene& S, V3 pc, float rho_max) {
  double pn = sqrt(dot3(pc, pc));
  double dp = pn * (double)kEps32 * 1.0001;
  double B = (double)S.r2_max * (double)rho_max;
  double r2s = 2.0 * (double)S.r_max * dp + dp * dp;
  double hmax = pn + (double)S.c_max + B;
  double slack = r2s * (1.0 + 2.0 * hmax * (double)rho_max) + 64.0 * (double)kEps32 * (B * B + (double)S.r2_max);
  return -(float)(slack * 1.001) - 1e-30f;
}

// ------------------------------------------------------------------------------------
// Shading, all f64, following the reference line by line.

// `Texture::sample_at` (src/scene/materials.rs:14-19) and the four patterns
// (src/graphics/patterns.rs:57-65, 81-88, 105-116, 132-140).  Rust's `%` is fmod.
__device__ __forceinline__ V3 sample_texture(const DevObject& ob, V3 p) {
  V3 a = ld3(ob.color_a), b = ld3(ob.color_b);
  switch (ob.texture_kind) {
    case 1u:  // checker
      return (fmod(floor(p.x) + floor(p.y) + floor(p.z), 2.0) == 0.0) ? a : b;
    case 2u: {  // gradient
      V3 distance = b - a;
      double fraction = p.x - floor(p.x);
      return a + distance * fraction;
    }
    case 3u: {  // ring
      double x2 = p.x * p.x, z2 = p.z * p.z;
      return (fmod(floor(sqrt(x2 + z2)), 2.0) == 0.0) ? a : b;
    }
    case 4u:  // striped
      return (fmod(floor(p.x), 2.0) == 0.0) ? a : b;
    default:
      return a;
  }
}

// `LightingData::calculate` (src/scene/lighting.rs:40-78) without the refractive indices.
struct HitGeom {
  V3 over, under, normal, eye, reflect, objp;
};
__device__ __forceinline__ HitGeom lighting_data(const DevObject& ob, V3 o, V3 d, double t) {
  HitGeom h;
  V3 world = o + d * t;
  h.eye = -d;
  V3 n;
  if (ob.kind == 1u) {
    n = ld3(ob.plane_normal);  // never transformed (src/scene/shapes/planes.rs:30-32)
  } else {
    // src/scene/shapes/spheres.rs:37-44: inverse^T * (object_point - origin), w := 0, normalize
    V3 op = xform(ob.inv, world, 1.0);
    const double* m = ob.inv;
    n = {m[0] * op.x + m[4] * op.y + m[8] * op.z, m[1] * op.x + m[5] * op.y + m[9] * op.z,
         m[2] * op.x + m[6] * op.y + m[10] * op.z};
    double mag = sqrt(n.x * n.x + n.y * n.y + n.z * n.z);
    n = {n.x / mag, n.y / mag, n.z / mag};
  }
  // offsets, texture point and reflection use the normal BEFORE the inside flip
  h.over = world + n * 0.0001;
  h.under = world - n * 0.0001;
  h.objp = xform(ob.inv, h.over, 1.0);
  h.reflect = d - (n * 2.0) * dot3(d, n);
  if (dot3(n, h.eye) < 0.0) n = -n;
  h.normal = n;
  return h;
}

// `phong_lighting` (src/scene/lighting.rs:120-154) for one light; `tex` is the sampled
// texture colour.  The light vector is normalised over four components: its w is
// L.w - 1 (the shading point has w = 1).
__device__ __forceinline__ V3 phong_lighting(const DevLight& L, const DevObject& ob, V3 tex, V3 over, V3 eye, V3 normal,
                                             bool in_shadow) {
  V3 intensity = ld3(L.intensity);
  V3 effective = tex * intensity;
  V3 ambient = effective * ob.ambient;
  if (in_shadow) return ambient;
  V3 lv = ld3(L.pos) - over;
  double lw = L.pos[3] - 1.0;
  double mag = sqrt(lv.x * lv.x + lv.y * lv.y + lv.z * lv.z + lw * lw);
  V3 ld = {lv.x / mag, lv.y / mag, lv.z / mag};
  V3 diffuse = {0, 0, 0}, specular = {0, 0, 0};
  double ldn = dot3(ld, normal);
  if (ldn >= 0.0) {
    diffuse = (effective * ob.diffuse) * ldn;
    V3 rd = -(ld - (normal * 2.0) * ldn);
    double rde = dot3(rd, eye);
    if (rde >= 0.0) {
      double factor = pow(rde, ob.shininess);
      specular = (intensity * ob.specular) * factor;
    }
  }
  return (ambient + diffuse) + specular;
}

// `Scene::shlick` (src/scene.rs:275-295); powi(5) by repeated squaring.
__device__ __forceinline__ double shlick(double n1, double n2, double cosv) {
  if (n1 > n2) {
    double n = n1 / n2;
    double sin_t2 = n * n * (1.0 - cosv * cosv);
    if (sin_t2 > 1.0) return 1.0;
    cosv = sqrt(1.0 - sin_t2);
  }
  double r0 = (n1 - n2) / (n1 + n2);
  double r02 = r0 * r0;
  double x = 1.0 - cosv;
  double x2 = x * x;
  return r02 + (1.0 - r02) * (x * (x2 * x2));
}

__device__ __forceinline__ bool is_approx_zero(double v) { return fabs(v) < 0.00001; }  // src/maths.rs:24-28

// Rust's `as u8`: truncate toward zero, saturate, NaN -> 0 (src/graphics.rs:65-69)
__device__ __forceinline__ uint32_t cast_u8(double c) {
  double v = c * 255.0;
  if (!(v > 0.0)) return 0u;
  if (v >= 255.0) return 255u;
  return (uint32_t)v;
}

__device__ __forceinline__ unsigned wave_sum(unsigned v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// ------------------------------------------------------------------------------------
// Shadow test for one shading point: `Scene::is_shadowed` (src/scene.rs:208-225).
// Lights are swept in batches of LB fused rays; the reference stops at the first blocked
// light, so `n_rays` = index of the first blocked light + 1 (or the light count).
template <int LB>
__device__ __forceinline__ bool shadow_sweeps(const DevScene& S, float4* s_buf, bool active, V3 p, unsigned& n_rays,
                                              unsigned& n_exact) {
  bool in_shadow = false;
  unsigned rays = 0;
  bool need = active;
  for (uint32_t kb = 0; kb < S.n_lights; kb += LB) {
    if (!__syncthreads_or(need)) break;  // nobody in the tile still needs a light
    const uint32_t nb = min((uint32_t)LB, S.n_lights - kb);
    uint32_t occl = 0;
    // planes: exact, per light (the shadow ray is rebuilt once per light)
    if (need) {
      for (uint32_t r = 0; r < nb; ++r) {
        const DevLight& L = S.lights[kb + r];
        V3 lv = ld3(L.pos) - p;
        double lw = L.pos[3] - 1.0;
        double m = sqrt(lv.x * lv.x + lv.y * lv.y + lv.z * lv.z + lw * lw);
        V3 d = {lv.x / m, lv.y / m, lv.z / m};
        double dw = lw / m;
        for (uint32_t ip = 0; ip < S.n_planes; ++ip) {
          double t0, t1;
          int n = intersect_object(S.objects[S.plane_object[ip]], p, d, dw, t0, t1);
          if (n >= 1 && t0 > 0.0 && t0 < m) occl |= 1u << r;
        }
      }
    }
    // spheres: fused FP32 filter, exact f64 on the candidates
    float gx[LB], gy[LB], gz[LB], rho[LB];
    float rho_max = 0.0f;
    bool exotic = false;
#pragma unroll
    for (int r = 0; r < LB; ++r) {
      const DevLight& L = S.lights[kb + min((uint32_t)r, nb - 1)];  // pad with the last real light
      double lw = L.pos[3];
      V3 G = ld3(L.pos) - p * lw;
      make_filter_ray(G, lw - 1.0, gx[r], gy[r], gz[r], rho[r]);
      rho_max = fmaxf(rho_max, fabsf(rho[r]) * 1.0001f);
      exotic |= !(lw >= 0.0 && lw <= 1.0);
    }
    V3 pc = {p.x - S.center[0], p.y - S.center[1], p.z - S.center[2]};
    float thr = exotic ? -3.0e38f : filter_threshold(S, pc, rho_max);
    bool live = need && !(occl & 1u);
    auto exact = [&](uint32_t obj, uint32_t mask) {
      const DevObject& ob = S.objects[obj];
      for (uint32_t r = 0; r < nb; ++r) {
        if (!((mask >> r) & 1u) || ((occl >> r) & 1u)) continue;
        const DevLight& L = S.lights[kb + r];
        V3 lv = ld3(L.pos) - p;
        double lw = L.pos[3] - 1.0;
        double m = sqrt(lv.x * lv.x + lv.y * lv.y + lv.z * lv.z + lw * lw);
        V3 d = {lv.x / m, lv.y / m, lv.z / m};
        double t0, t1;
        int n = intersect_object(ob, p, d, lw / m, t0, t1);
        ++n_exact;
        if (n == 2 && ((t0 > 0.0 && t0 < m) || (t1 > 0.0 && t1 < m))) occl |= 1u << r;
      }
    };
    auto done = [&]() { return (occl & 1u) != 0; };
    sweep_filter<LB>(S, s_buf, live, (float)pc.x, (float)pc.y, (float)pc.z, gx, gy, gz, rho, thr, exact, done);
    if (need) {
      occl &= (1u << nb) - 1u;
      if (occl) {
        rays += (unsigned)__builtin_ctz(occl) + 1u;
        in_shadow = true;
        need = false;
      } else {
        rays += nb;
      }
    }
  }
  n_rays += rays;
  return in_shadow;
}

// ------------------------------------------------------------------------------------
// Closest hit of a general ray (origin o, direction d with w = 0, not necessarily unit):
// `Scene::intersect` + `HitList::closest_hit` (src/scene.rs:156-169, 325-340).
__device__ __forceinline__ void closest_general(const DevScene& S, float4* s_buf, bool active, V3 o, V3 d, Closest& best,
                                                unsigned& n_exact) {
  closest_init(best);
  const bool track = S.track_n1 != 0;
  if (active) {
    for (uint32_t ip = 0; ip < S.n_planes; ++ip) {
      uint32_t obj = S.plane_object[ip];
      double t0, t1;
      if (intersect_object(S.objects[obj], o, d, 0.0, t0, t1)) closest_consider(best, t0, obj, track);
    }
  }
  float gx[1], gy[1], gz[1], rho[1];
  make_filter_ray(d, 0.0, gx[0], gy[0], gz[0], rho[0]);
  V3 pc = {o.x - S.center[0], o.y - S.center[1], o.z - S.center[2]};
  float thr = filter_threshold(S, pc, 0.0f);
  auto exact = [&](uint32_t obj, uint32_t) {
    double t0, t1;
    ++n_exact;
    if (intersect_object(S.objects[obj], o, d, 0.0, t0, t1)) {
      closest_consider(best, t0, obj, track);
      closest_consider(best, t1, obj, track);
    }
  };
  auto done = [&]() { return false; };
  bool live = active;
  sweep_filter<1>(S, s_buf, live, (float)pc.x, (float)pc.y, (float)pc.z, gx, gy, gz, rho, thr, exact, done);
}

// Closest hit of a camera ray (unit direction, shared origin).
__device__ __forceinline__ void closest_primary(const DevScene& S, const float4* __restrict__ prim, float4* s_buf, bool active,
                                                V3 o, V3 d, Closest& best, unsigned& n_exact) {
  closest_init(best);
  const bool track = S.track_n1 != 0;
  if (active) {
    for (uint32_t ip = 0; ip < S.n_planes; ++ip) {
      uint32_t obj = S.plane_object[ip];
      double t0, t1;
      if (intersect_object(S.objects[obj], o, d, 0.0, t0, t1)) closest_consider(best, t0, obj, track);
    }
  }
  auto exact = [&](uint32_t obj) {
    double t0, t1;
    ++n_exact;
    if (intersect_object(S.objects[obj], o, d, 0.0, t0, t1)) {
      closest_consider(best, t0, obj, track);
      closest_consider(best, t1, obj, track);
    }
  };
  sweep_primary(S, prim, s_buf, active, (float)d.x, (float)d.y, (float)d.z, exact);
}

// ------------------------------------------------------------------------------------
// `Scene::trace` (src/scene.rs:136-296) for one ray per thread, evaluated iteratively:
// child colours enter their parent linearly with scalar factors (reflectivity,
// transparency, Schlick reflectance), so each node adds weight * surface to the pixel and
// pushes its children with the product weight.  SECONDARY = false removes the stack for
// scenes without reflective or transparent materials.
struct StackEntry {
  double ox, oy, oz, dx, dy, dz, weight;
  int depth;
};

template <bool SECONDARY, bool PRIMARY_FILTER, int LB>
__device__ __forceinline__ V3 trace_ray(const DevScene& S, const float4* __restrict__ prim, float4* s_buf, bool valid, V3 o,
                                        V3 d, unsigned& c_primary, unsigned& c_shadow, unsigned& c_secondary,
                                        unsigned& c_exact) {
  V3 color = {0, 0, 0};
  const V3 scene_ambient = ld3(S.ambient);

  StackEntry stack[SECONDARY ? kMaxDepth + 1 : 1];
  int sp = 0;
  // current ray
  V3 ro = o, rd = d;
  double weight = 1.0;
  int depth = 0;
  bool have = valid && S.max_depth > 0;
  if (valid && S.max_depth == 0) color = scene_ambient;  // trace_inner(depth 0 >= MAX_DEPTH)

  for (int iter = 0;; ++iter) {
    if (SECONDARY) {
      if (iter > 0) {
        have = sp > 0;
        if (have) {
          --sp;
          const StackEntry& e = stack[sp];
          ro = {e.ox, e.oy, e.oz};
          rd = {e.dx, e.dy, e.dz};
          weight = e.weight;
          depth = e.depth;
        }
      }
      if (!__syncthreads_or(have)) break;
    } else if (iter > 0) {
      break;
    }

    if (have) {
      if (depth == 0)
        ++c_primary;
      else
        ++c_secondary;
    }
    Closest best;
    if (PRIMARY_FILTER && (!SECONDARY || iter == 0))
      closest_primary(S, prim, s_buf, have, ro, rd, best, c_exact);
    else
      closest_general(S, s_buf, have, ro, rd, best, c_exact);

    const bool hit = have && best.obj != 0xffffffffu;
    if (have && !hit) color = color + scene_ambient * weight;  // trace_inner: miss -> ambient_color

    // --- apply_lighting (src/scene.rs:172-205) ---
    V3 over = {0, 0, 0}, normal = {0, 0, 1}, eye = {0, 0, 1}, tex = {0, 0, 0};
    uint32_t obj = hit ? best.obj : 0u;
    if (hit) {
      const DevObject& ob = S.objects[obj];
      HitGeom g = lighting_data(ob, ro, rd, best.t);
      over = g.over;
      normal = g.normal;
      eye = g.eye;
      tex = sample_texture(ob, g.objp);
      if (SECONDARY) {
        // children first, so that nothing of them stays live across the shadow sweep
        const double reflectivity = ob.reflectivity, transparency = ob.transparency;
        const bool has_refl = !is_approx_zero(reflectivity);   // reflected_color early-out
        const bool has_refr = !is_approx_zero(transparency);   // refracted_color early-out
        if (has_refl || has_refr) {
          double n2 = ob.refractivity;
          // n1: material of the entry just before the hit in the sorted list, else 1.0.  When
          // every refractivity in the scene is 1.0 (track_n1 == 0) that is 1.0 either way.
          double n1 = 1.0;
          if (S.track_n1 && best.prev_obj != 0xffffffffu) n1 = S.objects[best.prev_obj].refractivity;
          const double cos_i = dot3(eye, normal);
          double w_refl = weight * reflectivity, w_refr = weight * transparency;
          if (reflectivity > 0.0 && transparency > 0.0) {
            double reflectance = shlick(n1, n2, cos_i);
            w_refl = weight * (reflectivity * reflectance);
            w_refr = weight * (transparency * (1.0 - reflectance));
          }
          const bool child_traces = (uint32_t)(depth + 1) < S.max_depth;
          if (has_refr) {  // src/scene.rs:244-272
            double n_ratio = n1 / n2;
            double sin_t2 = n_ratio * n_ratio * (1.0 - cos_i * cos_i);
            if (!(sin_t2 > 1.0)) {
              double cos_t = sqrt(1.0 - sin_t2);
              V3 dir = normal * (n_ratio + cos_i - cos_t) - eye * n_ratio;  // the reference's `+`
              if (child_traces) {
                StackEntry& e = stack[sp++];
                e.ox = g.under.x, e.oy = g.under.y, e.oz = g.under.z;
                e.dx = dir.x, e.dy = dir.y, e.dz = dir.z;
                e.weight = w_refr;
                e.depth = depth + 1;
              } else {
                color = color + scene_ambient * w_refr;  // trace_inner(depth >= MAX_DEPTH)
              }
            }
          }
          if (has_refl) {  // src/scene.rs:228-241
            if (child_traces) {
              StackEntry& e = stack[sp++];
              e.ox = g.over.x, e.oy = g.over.y, e.oz = g.over.z;
              e.dx = g.reflect.x, e.dy = g.reflect.y, e.dz = g.reflect.z;
              e.weight = w_refl;
              e.depth = depth + 1;
            } else {
              color = color + scene_ambient * w_refl;
            }
          }
        }
      }
    }

    const bool in_shadow = shadow_sweeps<LB>(S, s_buf, hit, over, c_shadow, c_exact);

    if (hit) {
      const DevObject& ob = S.objects[obj];
      V3 surface = scene_ambient;
      for (uint32_t k = 0; k < S.n_lights; ++k)
        surface = surface + phong_lighting(S.lights[k], ob, tex, over, eye, normal, in_shadow);
      color = color + surface * weight;
    }
  }
  return color;
}

// ------------------------------------------------------------------------------------
// `Camera::render` (src/scene/cameras.rs:72-85): ray generation (cameras.rs:52-69 with the
// matrix inverse hoisted to the host), trace, and the framebuffer store
// (src/graphics.rs:44-46, 59-74).
template <bool SECONDARY, int LB>
__global__ void __launch_bounds__(kBlock) render_kernel(DevScene S, DevCamera C, const float4* __restrict__ prim,
                                                        RenderTarget T, DevCounters* counters) {
  __shared__ float4 s_buf[kChunk];

  // expensive tiles (ground, bottom of the frame) first, cheap sky tiles last
  const uint32_t k = T.n_local_tiles - 1u - blockIdx.x;
  const uint32_t tile = k * T.world + T.rank;
  const uint32_t tx = tile % T.tiles_x, ty = tile / T.tiles_x;
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  const uint32_t lx = (wave << 3) | (lane & 7u), ly = lane >> 3;
  const uint32_t x = tx * 32u + lx, y = ty * 8u + ly;
  const bool valid = x < C.width && y < C.height;

  // cameras.rs:52-69
  const double x_offset = ((double)x + 0.5) * C.pixel_size;
  const double y_offset = ((double)y + 0.5) * C.pixel_size;
  const double world_x = C.half_width - x_offset;
  const double world_y = C.half_height - y_offset;
  V3 pixel = xform(C.inv, V3{world_x, world_y, -1.0}, 1.0);
  V3 origin = ld3(C.origin);
  V3 dv = pixel - origin;
  double mag = sqrt(dv.x * dv.x + dv.y * dv.y + dv.z * dv.z);
  V3 dir = {dv.x / mag, dv.y / mag, dv.z / mag};

  unsigned c_primary = 0, c_shadow = 0, c_secondary = 0, c_exact = 0;
  V3 color = trace_ray<SECONDARY, true, LB>(S, prim, s_buf, valid, origin, dir, c_primary, c_shadow, c_secondary, c_exact);

  // ---- framebuffer store, transposed through LDS so that every wavefront writes whole
  // 128-byte (RGBA8) / 384-byte (FP32 RGB) tile rows
  __syncthreads();
  float* s_rgb = reinterpret_cast<float*>(s_buf);              // [256][3]
  uint32_t* s_px = reinterpret_cast<uint32_t*>(s_buf) + 768;  // [256]
  const uint32_t slot = ly * 32u + lx;
  s_rgb[slot * 3 + 0] = (float)color.x;
  s_rgb[slot * 3 + 1] = (float)color.y;
  s_rgb[slot * 3 + 2] = (float)color.z;
  s_px[slot] = cast_u8(color.x) | (cast_u8(color.y) << 8) | (cast_u8(color.z) << 16) | 0xff000000u;
  __syncthreads();
  {
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
        const uint32_t e = threadIdx.x + pass * 256u;  // element of the tile's [8][96] float image
        const uint32_t row = e / 96u, col = e % 96u;
        if (T.compact) {
          T.rgb[(size_t)k * 768u + e] = s_rgb[e];
        } else {
          const uint32_t px = tx * 32u + col / 3u, py = ty * 8u + row;
          if (px < C.width && py < C.height) T.rgb[((size_t)px + (size_t)py * C.width) * 3u + col % 3u] = s_rgb[e];
        }
      }
    }
  }

  // ray counters: one atomic per wavefront
  c_primary = wave_sum(c_primary);
  c_shadow = wave_sum(c_shadow);
  c_secondary = wave_sum(c_secondary);
  c_exact = wave_sum(c_exact);
  if (lane == 0) {
    atomicAdd(&counters->rays_primary, (unsigned long long)c_primary);
    atomicAdd(&counters->rays_shadow, (unsigned long long)c_shadow);
    atomicAdd(&counters->rays_secondary, (unsigned long long)c_secondary);
    atomicAdd(&counters->exact_tests, (unsigned long long)c_exact);
  }
}

// `Scene::trace` for caller-supplied rays (src/scene.rs:136-138): n x 4 doubles each.
template <int LB>
__global__ void __launch_bounds__(kBlock) trace_rays_kernel(DevScene S, const double* __restrict__ origins,
                                                            const double* __restrict__ directions, uint64_t n,
                                                            double* __restrict__ colors, DevCounters* counters) {
  __shared__ float4 s_buf[kChunk];
  const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  const bool valid = i < n;
  V3 o = {0, 0, 0}, d = {0, 0, 1};
  if (valid) {
    o = ld3(origins + 4 * i);
    d = ld3(directions + 4 * i);
  }
  unsigned c_primary = 0, c_shadow = 0, c_secondary = 0, c_exact = 0;
  V3 color = trace_ray<true, false, LB>(S, nullptr, s_buf, valid, o, d, c_primary, c_shadow, c_secondary, c_exact);
  if (valid) {
    colors[3 * i + 0] = color.x;
    colors[3 * i + 1] = color.y;
    colors[3 * i + 2] = color.z;
  }
  c_primary = wave_sum(c_primary);
  c_shadow = wave_sum(c_shadow);
  c_secondary = wave_sum(c_secondary);
  c_exact = wave_sum(c_exact);
  if ((threadIdx.x & 63u) == 0) {
    atomicAdd(&counters->rays_primary, (unsigned long long)c_primary);
    atomicAdd(&counters->rays_shadow, (unsigned long long)c_shadow);
    atomicAdd(&counters->rays_secondary, (unsigned long long)c_secondary);
    atomicAdd(&counters->exact_tests, (unsigned long long)c_exact);
  }
}

// Per-camera record of the primary filter: one thread per (padded) sphere.
__global__ void __launch_bounds__(kBlock) prep_primary_kernel(DevScene S, DevCamera C, float4* __restrict__ prim) {
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= S.n_spheres_padded) return;
  // padding records repeat the last real sphere (see rtk_scene_upload)
  const double4 s = S.sph_exact[min(i, S.n_spheres - 1u)];
  const double Lx = s.x - C.origin[0], Ly = s.y - C.origin[1], Lz = s.z - C.origin[2];
  const double LL = Lx * Lx + Ly * Ly + Lz * Lz;
  const double eps = (double)kEps32;
  // rounding L to FP32 moves the centre by at most eps|L|: grow the ball by that much
  const double R = s.w * (1.0 + 1e-12) + 1.01 * eps * sqrt(LL);
  // 12 eps |L|^2 covers the dot product (3 roundings), the rounded direction and this record's own rounding
  double cc = LL * (1.0 - 12.0 * eps) - R * R;
  float ccf = (float)cc;
  if ((double)ccf > cc) ccf = nextafterf(ccf, -INFINITY);
  ccf = nextafterf(ccf, -INFINITY);
  prim[i] = make_float4((float)Lx, (float)Ly, (float)Lz, ccf);
}

// Rebuilds row-major frames from `world` compact tile buffers laid end to end.
__global__ void __launch_bounds__(kBlock) untile_kernel(uint32_t width, uint32_t height, uint32_t tiles_x, uint32_t n_tiles,
                                                        uint32_t world, uint64_t slot_pixels, const float* __restrict__ rgb_c,
                                                        float* __restrict__ rgb_f, const uint32_t* __restrict__ px_c,
                                                        uint32_t* __restrict__ px_f) {
  const uint32_t tile = blockIdx.x;
  if (tile >= n_tiles) return;
  const uint32_t rank = tile % world, k = tile / world;
  const uint32_t tx = tile % tiles_x, ty = tile / tiles_x;
  const size_t src = (size_t)rank * slot_pixels + (size_t)k * 256u;
  const uint32_t rx = threadIdx.x & 31u, ry = threadIdx.x >> 5;
  const uint32_t gx = tx * 32u + rx, gy = ty * 8u + ry;
  if (px_c && gx < width && gy < height) px_f[(size_t)gx + (size_t)gy * width] = px_c[src + threadIdx.x];
  if (rgb_c) {
#pragma unroll
    for (uint32_t pass = 0; pass < 3; ++pass) {
      const uint32_t e = threadIdx.x + pass * 256u;
      const uint32_t row = e / 96u, col = e % 96u;
      const uint32_t px = tx * 32u + col / 3u, py = ty * 8u + row;
      if (px < width && py < height) rgb_f[((size_t)px + (size_t)py * width) * 3u + col % 3u] = rgb_c[src * 3u + e];
    }
  }
}

// ------------------------------------------------------------------------------------
// launch wrappers (called from rtk_api.hip)

template <bool SECONDARY>
static hipError_t launch_render_lb(int lb, dim3 grid, hipStream_t st, const DevScene& S, const DevCamera& C,
                                   const float4* prim, const RenderTarget& T, DevCounters* counters) {
  switch (lb) {
    case 1: hipLaunchKernelGGL((render_kernel<SECONDARY, 1>), grid, dim3(kBlock), 0, st, S, C, prim, T, counters); break;
    case 2: hipLaunchKernelGGL((render_kernel<SECONDARY, 2>), grid, dim3(kBlock), 0, st, S, C, prim, T, counters); break;
    case 4: hipLaunchKernelGGL((render_kernel<SECONDARY, 4>), grid, dim3(kBlock), 0, st, S, C, prim, T, counters); break;
    default: hipLaunchKernelGGL((render_kernel<SECONDARY, 8>), grid, dim3(kBlock), 0, st, S, C, prim, T, counters); break;
  }
  return hipGetLastError();
}

int light_batch_for(uint32_t n_lights) {
  if (n_lights <= 1) return 1;
  if (n_lights == 2) return 2;
  if (n_lights <= 4) return 4;
  return 8;
}

hipError_t launch_prep_primary(hipStream_t st, const DevScene& S, const DevCamera& C, float4* prim) {
  if (S.n_spheres_padded == 0) return hipSuccess;
  dim3 grid((S.n_spheres_padded + kBlock - 1) / kBlock);
  hipLaunchKernelGGL(prep_primary_kernel, grid, dim3(kBlock), 0, st, S, C, prim);
  return hipGetLastError();
}

hipError_t launch_render(hipStream_t st, const DevScene& S, const DevCamera& C, const float4* prim, const RenderTarget& T,
                         DevCounters* counters) {
  if (T.n_local_tiles == 0) return hipSuccess;
  dim3 grid(T.n_local_tiles);
  const int lb = light_batch_for(S.n_lights);
  if (S.any_secondary) return launch_render_lb<true>(lb, grid, st, S, C, prim, T, counters);
  return launch_render_lb<false>(lb, grid, st, S, C, prim, T, counters);
}

hipError_t launch_trace_rays(hipStream_t st, const DevScene& S, const double* origins, const double* directions, uint64_t n,
                             double* colors, DevCounters* counters) {
  if (n == 0) return hipSuccess;
  dim3 grid((unsigned)((n + kBlock - 1) / kBlock));
  switch (light_batch_for(S.n_lights)) {
    case 1: hipLaunchKernelGGL((trace_rays_kernel<1>), grid, dim3(kBlock), 0, st, S, origins, directions, n, colors, counters); break;
    case 2: hipLaunchKernelGGL((trace_rays_kernel<2>), grid, dim3(kBlock), 0, st, S, origins, directions, n, colors, counters); break;
    case 4: hipLaunchKernelGGL((trace_rays_kernel<4>), grid, dim3(kBlock), 0, st, S, origins, directions, n, colors, counters); break;
    default: hipLaunchKernelGGL((trace_rays_kernel<8>), grid, dim3(kBlock), 0, st, S, origins, directions, n, colors, counters); break;
  }
  return hipGetLastError();
}

hipError_t launch_untile(hipStream_t st, uint32_t width, uint32_t height, uint32_t world, uint64_t slot_pixels,
                         const float* rgb_c, float* rgb_f, const uint8_t* px_c, uint8_t* px_f) {
  const uint32_t tiles_x = (width + 31u) / 32u, tiles_y = (height + 7u) / 8u;
  const uint32_t n_tiles = tiles_x * tiles_y;
  if (n_tiles == 0) return hipSuccess;
  hipLaunchKernelGGL(untile_kernel, dim3(n_tiles), dim3(kBlock), 0, st, width, height, tiles_x, n_tiles, world, slot_pixels,
                     rgb_c, rgb_f, reinterpret_cast<const uint32_t*>(px_c), reinterpret_cast<uint32_t*>(px_f));
  return hipGetLastError();
}

}  // namespace rtk
