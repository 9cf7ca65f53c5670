// rtk_shading.h -- device-side building blocks shared by the kernels: f64 vector helpers,
// the exact object intersection, closest-hit bookkeeping, the reference's shading model,
// and the set-up of the FP32 candidate filter.  Device code only (included by .hip files).
#ifndef RTK_SHADING_H
#define RTK_SHADING_H

#include "rtk_device.h"

#include <float.h>

// The f64 code below restates the reference's arithmetic operation for operation.  rustc
// never fuses a multiply with an add, so contraction stays off here as well: the exact
// tests then take bit-identical decisions even in degenerate cases (tangent rays, points
// lying on a surface).  The FP32 filter spells out its fused multiply-adds explicitly.
#pragma clang fp contract(off)

namespace rtk {

// relative slack of the general filter: |u|^2 is scaled by this before R'^2 is subtracted
constexpr float kKappa = 1.0f - 24.0f * kEps32;

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
template <typename P>
__device__ __forceinline__ V3 ld3(P p) {  // P: pointer to doubles in any address space
  return {p[0], p[1], p[2]};
}

// rows 0..2 of an affine 4x4 times (v, w): the 4-term rows of src/maths/vectors.rs:205-223
template <typename P>
__device__ __forceinline__ V3 xform(P m, V3 v, double w) {
  return {m[0] * v.x + m[1] * v.y + m[2] * v.z + m[3] * w, m[4] * v.x + m[5] * v.y + m[6] * v.z + m[7] * w,
          m[8] * v.x + m[9] * v.y + m[10] * v.z + m[11] * w};
}

// ------------------------------------------------------------------------------------
// Several IEEE divisions by one divisor.  The compiler's f64 division scales both operands away from
// the ends of the exponent range (v_div_scale), refines a hardware reciprocal of the divisor by two
// Newton steps, forms q0 = n r and one correction q = q0 + (n - d q0) r (v_div_fmas), and patches the
// special cases (v_div_fixup).  For operands well inside the exponent range nothing is scaled or
// patched: the quotient is exactly the sequence below, and the refined reciprocal -- 9 of its 13
// instructions, among them the quarter-rate v_rcp_f64 -- depends on the divisor alone.  The functions
// below share it between the numerators when every operand is in that range and divide one by one
// otherwise (zeros, infinities, NaNs, very small and very large values): bit for bit the quotients of
// `/` either way.
__device__ __forceinline__ bool div_plain(double x) { return fabs(x) >= 0x1p-400 && fabs(x) < 0x1p400; }
__device__ __forceinline__ double refined_reciprocal(double d) {
  double r = __builtin_amdgcn_rcp(d);
  double e = __builtin_fma(-d, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-d, r, 1.0);
  return __builtin_fma(r, e, r);
}
__device__ __forceinline__ double quotient(double n, double d, double r) {
  const double q = n * r;
  return __builtin_fma(__builtin_fma(-d, q, n), r, q);
}
// {n0 / d, n1 / d}
__device__ __forceinline__ void divide2(double n0, double n1, double d, double& q0, double& q1) {
  if (div_plain(d) && div_plain(n0) && div_plain(n1)) {
    const double r = refined_reciprocal(d);
    q0 = quotient(n0, d, r), q1 = quotient(n1, d, r);
  } else {
    q0 = n0 / d, q1 = n1 / d;
  }
}
// v / m for m = the length of a vector v is a part of: |v.i| <= m (1 + a few ulp), so a lower bound on
// the components and an upper bound on m keep every operand in range
__device__ __forceinline__ V3 divide_by_length(V3 v, double m) {
  if (fabs(v.x) >= 0x1p-400 && fabs(v.y) >= 0x1p-400 && fabs(v.z) >= 0x1p-400 && m < 0x1p400) {
    const double r = refined_reciprocal(m);
    return {quotient(v.x, m, r), quotient(v.y, m, r), quotient(v.z, m, r)};
  }
  return {v.x / m, v.y / m, v.z / m};
}
// ... and a fourth component w (|w| <= m as well; exactly zero for lights built from points)
__device__ __forceinline__ V3 divide_by_length(V3 v, double w, double m, double& qw) {
  if (fabs(v.x) >= 0x1p-400 && fabs(v.y) >= 0x1p-400 && fabs(v.z) >= 0x1p-400 && m < 0x1p400 && (w == 0.0 || fabs(w) >= 0x1p-400)) {
    const double r = refined_reciprocal(m);
    qw = w == 0.0 ? w : quotient(w, m, r);  // m > 0: a zero keeps its sign
    return {quotient(v.x, m, r), quotient(v.y, m, r), quotient(v.z, m, r)};
  }
  qw = w / m;
  return {v.x / m, v.y / m, v.z / m};
}

// f64 square root.  The compiler expands sqrt() into: scale the argument up when it is below 2^-767,
// a hardware reciprocal square root refined by the ten operations below, scale the result back, and
// return the argument itself for zeros and infinities.  For an argument in [2^-767, infinity) nothing is
// scaled and nothing is patched, so the ten operations ARE sqrt(): the same instructions on the same
// values.  They are issued alone when every lane of the wavefront that is here holds such an argument
// (a wavefront-uniform branch, no selects); anything else -- zeros, subnormals and tiny values, negative
// numbers, infinities, NaNs -- takes the library's sequence.
__device__ __forceinline__ double sqrt_f64(double x) {
  if (__builtin_amdgcn_ballot_w64(!(x >= 0x1p-767 && x < INFINITY)) == 0ull) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    double h = y * 0.5;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    const double d0 = __builtin_fma(-g, g, x);
    h = __builtin_fma(h, r, h);
    g = __builtin_fma(d0, h, g);
    const double d1 = __builtin_fma(-g, g, x);
    return __builtin_fma(d1, h, g);
  }
  return sqrt(x);
}

// `Sphere::intersect` (src/scene/shapes/spheres.rs:18-35) on an object-space ray: the unit-sphere
// quadratic.  direction.w stays dw (row 3 of the transforms is 0 0 0 1) and takes part in `a`;
// sphere_to_ray.w = 1 - 1 = 0.
__device__ __forceinline__ int unit_sphere_roots(V3 oo, V3 dd, double dw, double& t0, double& t1) {
  double a = dd.x * dd.x + dd.y * dd.y + dd.z * dd.z + dw * dw;
  double b = 2.0 * (oo.x * dd.x + oo.y * dd.y + oo.z * dd.z);
  double c = (oo.x * oo.x + oo.y * oo.y + oo.z * oo.z) - 1.0;
  double disc = b * b - 4.0 * a * c;
  if (disc >= 0.0) {
    double s = sqrt_f64(disc);
    divide2(-b - s, -b + s, 2.0 * a, t0, t1);
    return 2;
  }
  return 0;
}

// ------------------------------------------------------------------------------------
// Exact f64 intersection of one object: `SceneNode::intersect` (src/scene.rs:75-84) with
// `Sphere::intersect` (src/scene/shapes/spheres.rs:18-35) or `Plane::intersect`
// (src/scene/shapes/planes.rs:22-28).  The ray origin has w = 1; dw is the direction's w
// (0 for camera and secondary rays, non-zero for shadow rays towards a w != 1 light).
__device__ __forceinline__ int intersect_object(const RTK_RO DevObject& ob, V3 o, V3 d, double dw, double& t0, double& t1) {
  const RTK_RO double* m = ob.inv;
  if (ob.kind == 1u) {
    double oy = m[4] * o.x + m[5] * o.y + m[6] * o.z + m[7];
    double dy = m[4] * d.x + m[5] * d.y + m[6] * d.z + m[7] * dw;
    if (fabs(dy) < 0.00001) return 0;
    t0 = -oy / dy;
    return 1;
  }
  V3 oo = xform(m, o, 1.0);
  V3 dd = xform(m, d, dw);
  return unit_sphere_roots(oo, dd, dw, t0, t1);
}

// The same for a sphere given by its sweep-order record.  Diagonal inverse: the row sums of
// inverse * origin and inverse * direction with their exact-zero terms left out.
__device__ __forceinline__ int intersect_sphere_record(const DevScene& S, const RTK_RO SphRecord& r, V3 o, V3 d, double dw,
                                                       double& t0, double& t1) {
  if (r.diagonal == 0u) return intersect_object(S.objects[r.object], o, d, dw, t0, t1);
  const V3 oo = {r.scale[0] * o.x + r.shift[0], r.scale[1] * o.y + r.shift[1], r.scale[2] * o.z + r.shift[2]};
  const V3 dd = {r.scale[0] * d.x + r.shift[0] * dw, r.scale[1] * d.y + r.shift[1] * dw, r.scale[2] * d.z + r.shift[2] * dw};
  return unit_sphere_roots(oo, dd, dw, t0, t1);
}

// Can the reference's shadow ray from p towards light L meet a plane at a positive parameter?
// `Plane::intersect` (src/scene/shapes/planes.rs:22-28) returns t = -o'.y / d'.y for the object-space
// origin o' and direction d' = inverse * (L - p, L.w - 1)/m.  o'.y is formed here exactly as
// intersect_object forms it.  d'.y m is row 1 of the inverse applied to (L - p, L.w - 1): in real
// numbers G = row.(L, L.w - 1) - row.(p, 0), and both the reference's value and the difference g formed
// here lie within a few ulp of the sum of the magnitudes of all those terms from G.  The light's part
// and its magnitudes come per (plane, light) from scene upload (DevScene::plane_light), the point's
// part once per (plane, point): two additions per light.  When g is clear of 1e-12 times the
// magnitudes, d'.y has its sign; when o'.y and d'.y share a sign, t < 0, and when o'.y == 0, t == 0:
// `is_shadowed` (src/scene.rs:208-225) wants t > 0 either way, so the square root and the divisions
// can be skipped.  plane_may_block returns false only then.
struct PlaneSide {
  double oy;     // o'.y
  double q;      // row.(p, 0)
  double noise;  // 1e-12 x the magnitudes of q's terms
};
__device__ __forceinline__ PlaneSide plane_side(const RTK_RO DevObject& ob, V3 p) {
  const RTK_RO double* m = ob.inv;
  const double a = m[4] * p.x, b = m[5] * p.y, c = m[6] * p.z;
  PlaneSide s;
  s.q = a + b + c;
  s.oy = s.q + m[7];  // ((a + b) + c) + m[7], as in intersect_object
  s.noise = 1.0e-12 * (fabs(a) + fabs(b) + fabs(c));
  return s;
}
__device__ __forceinline__ bool plane_may_block(const PlaneSide& s, double2 plane_light) {
  const double g = plane_light.x - s.q;
  const double noise = plane_light.y + s.noise;
  if (s.oy == 0.0) return false;
  if (s.oy > 0.0 && g > noise) return false;
  if (s.oy < 0.0 && g < -noise) return false;
  return true;
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
// The slack is a sum of products of non-negative terms: it is evaluated in FP32 from the rounded
// point (a few 1e-7 of relative error per operation) and inflated by 1e-4, which keeps it above the
// f64 expression the soundness tests model (tests/filter_model.py).
__device__ __forceinline__ float filter_threshold(const DevScene& S, V3 pc, float rho_max) {
  const float x = (float)pc.x, y = (float)pc.y, z = (float)pc.z;
  const float pn = __builtin_amdgcn_sqrtf(__builtin_fmaf(z, z, __builtin_fmaf(y, y, x * x)));
  const float dp = pn * kEps32 * 1.0001f;
  const float B = S.r2_max * rho_max;
  const float r2s = 2.0f * S.r_max * dp + dp * dp;
  const float hmax = pn + S.c_max + B;
  const float slack = r2s * (1.0f + 2.0f * hmax * rho_max) + 64.0f * kEps32 * (B * B + S.r2_max);
  return -(slack * 1.001f * (1.0f + 1.0e-4f)) - 1e-30f;
}

// ------------------------------------------------------------------------------------
// Shading, all f64, following the reference line by line.

// `x % 2.0 == 0.0` for an x that came out of floor() (or is a sum of such values): Rust's `%` is fmod,
// and for an integer-valued x the remainder is zero exactly when x / 2 is an integer too (every double
// from 2^53 on is even); fmod of an infinity is NaN, which equals nothing.
__device__ __forceinline__ bool even_integer(double x) {
  const double h = x * 0.5;
  return fabs(x) < INFINITY && floor(h) == h;
}

// `Texture::sample_at` (src/scene/materials.rs:14-19) and the four patterns
// (src/graphics/patterns.rs:57-65, 81-88, 105-116, 132-140).
__device__ __forceinline__ V3 sample_texture(const RTK_RO DevObject& ob, V3 p) {
  V3 a = ld3(ob.color_a), b = ld3(ob.color_b);
  switch (ob.texture_kind & 0xffu) {
    case 1u:  // checker
      return even_integer(floor(p.x) + floor(p.y) + floor(p.z)) ? a : b;
    case 2u: {  // gradient
      V3 distance = b - a;
      double fraction = p.x - floor(p.x);
      return a + distance * fraction;
    }
    case 3u: {  // ring
      double x2 = p.x * p.x, z2 = p.z * p.z;
      return even_integer(floor(sqrt_f64(x2 + z2))) ? a : b;
    }
    case 4u:  // striped
      return even_integer(floor(p.x)) ? a : b;
    default:
      return a;
  }
}

// `TransformPattern::sample_at` (src/graphics/patterns.rs:27-35) around the pattern: the
// object-space point (w = 1) goes through the inverse pattern transform first.  The inverse
// is formed once at scene upload with the reference's cofactor method; a singular transform
// is stored as "no transform".
// A pattern inside several TransformPatterns: every wrapper's inverse in turn, outermost first, as
// full 4x4 products (src/maths/vectors.rs:205-223) -- the transforms need not be affine, so w travels.
__device__ __forceinline__ V3 sample_texture_at(const DevScene& S, const RTK_RO DevObject& ob, uint32_t obj, V3 objp) {
  const uint32_t tk = ob.texture_kind;
  if (tk & 0x80000000u) {
    const uint32_t levels = (tk >> 8) & 0xfu;
    if (levels == 0u) {
      objp = xform(S.pattern_inv + 12 * (size_t)obj, objp, 1.0);
    } else {
      double w = 1.0;
      const RTK_RO double* m = S.pattern_chain + 16 * (size_t)((tk >> 12) & 0x7ffffu);
      for (uint32_t k = 0; k < levels; ++k, m += 16) {
        const V3 p = objp;
        objp = {m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3] * w, m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7] * w,
                m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11] * w};
        w = m[12] * p.x + m[13] * p.y + m[14] * p.z + m[15] * w;
      }
    }
  }
  return sample_texture(ob, objp);
}

// `LightingData::calculate` (src/scene/lighting.rs:40-78) without the refractive indices.
struct HitGeom {
  V3 over, under, normal, eye, reflect, objp;
};
__device__ __forceinline__ HitGeom lighting_data(const RTK_RO DevObject& ob, V3 o, V3 d, double t) {
  HitGeom h;
  V3 world = o + d * t;
  h.eye = -d;
  V3 n;
  if (ob.kind == 1u) {
    n = ld3(ob.plane_normal);  // never transformed (src/scene/shapes/planes.rs:30-32)
  } else {
    // src/scene/shapes/spheres.rs:37-44: inverse^T * (object_point - origin), w := 0, normalize
    V3 op = xform(ob.inv, world, 1.0);
    const RTK_RO double* m = ob.inv;
    n = {m[0] * op.x + m[4] * op.y + m[8] * op.z, m[1] * op.x + m[5] * op.y + m[9] * op.z,
         m[2] * op.x + m[6] * op.y + m[10] * op.z};
    double mag = sqrt_f64(n.x * n.x + n.y * n.y + n.z * n.z);
    n = divide_by_length(n, mag);
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

// `reflect_dot_eye.powf(material.shininess)` (src/scene/lighting.rs:143-147).  The reference's powf is the C
// library's pow: the correctly rounded power in all but a few cases per ten thousand.  The device library's
// pow is 260 instructions (a logarithm and an exponential in extended precision) and only good to an ulp.
// For an integer-valued exponent n in 1 .. 4095 -- every scene file's shininess -- the power is formed by
// squaring and multiplying, left to right over the bits of n, on an unevaluated sum h + l: every product is
// split exactly into its rounded value and its rounding error with one fused multiply-add, and the error
// terms ride along (the l^2 terms left out are below 2^-80 of the result).  The rounded sum h + l is the
// correctly rounded power (tests/test_powi_model.py restates the sequence against exact arithmetic), seven
// rounds of 7 f64 operations for n = 200.  A skipped multiplication is a multiplication by 1.0, which
// changes neither h nor l.  Below 2^-900 the error terms go subnormal and the last bits may differ from
// the library's; such a factor reaches no pixel (it is below FP32's range, and zero after the 8-bit cast).
// Other exponents (fractions, zero, negatives, 4096 and more) go to the library's pow.
__device__ __forceinline__ double highlight_power(double x, double shininess) {
  const double nf = floor(shininess);
  if (!(nf == shininess && shininess >= 1.0 && shininess < 4096.0 && x >= 0.0 && x < 2.0)) return pow(x, shininess);
  const uint32_t n = (uint32_t)nf;
  double h = x, l = 0.0;
  // Every lane that is here usually raises to the same power (one shininess for all spheres).  Then the
  // loop runs on the scalar unit and the multiplications of the exponent's zero bits are left out
  // altogether -- a multiplication by 1.0 changes neither h nor l -- which is 34 operations instead of
  // 112 for shininess 200.
  const uint32_t n_all = (uint32_t)__builtin_amdgcn_readfirstlane((int)n);
  if (__builtin_amdgcn_ballot_w64(n != n_all) == 0ull) {
    for (int b = 30 - (int)__builtin_clz(n_all); b >= 0; --b) {
      double p = h * h;
      double e = __builtin_fma(h, h, -p);
      e = __builtin_fma(h + h, l, e);
      h = p, l = e;
      if ((n_all >> b) & 1u) {
        p = h * x;
        e = __builtin_fma(h, x, -p);
        e = __builtin_fma(l, x, e);
        h = p, l = e;
      }
    }
    return h + l;
  }
  for (int b = 30 - (int)__builtin_clz(n); b >= 0; --b) {
    double p = h * h;
    double e = __builtin_fma(h, h, -p);
    e = __builtin_fma(h + h, l, e);
    h = p, l = e;
    const double m = ((n >> b) & 1u) ? x : 1.0;
    p = h * m;
    e = __builtin_fma(h, m, -p);
    e = __builtin_fma(l, m, e);
    h = p, l = e;
  }
  return h + l;
}

// `phong_lighting` (src/scene/lighting.rs:120-154) for one light; `tex` is the sampled
// texture colour.  The light vector is normalised over four components: its w is
// L.w - 1 (the shading point has w = 1).
__device__ __forceinline__ V3 phong_lighting(const RTK_RO DevLight& L, const RTK_RO DevObject& ob, V3 tex, V3 over, V3 eye, V3 normal,
                                             bool in_shadow) {
  V3 intensity = ld3(L.intensity);
  V3 effective = tex * intensity;
  V3 ambient = effective * ob.ambient;
  if (in_shadow) return ambient;
  V3 lv = ld3(L.pos) - over;
  double lw = L.pos[3] - 1.0;
  double mag = sqrt_f64(lv.x * lv.x + lv.y * lv.y + lv.z * lv.z + lw * lw);
  V3 ld = divide_by_length(lv, mag);
  V3 diffuse = {0, 0, 0}, specular = {0, 0, 0};
  double ldn = dot3(ld, normal);
  if (ldn >= 0.0) {
    diffuse = (effective * ob.diffuse) * ldn;
    V3 rd = -(ld - (normal * 2.0) * ldn);
    double rde = dot3(rd, eye);
    if (rde >= 0.0) {
      // The highlight is added last: where it is below a quarter ulp of every channel of
      // ambient + diffuse the f64 sum does not change, and `pow` (the costliest call of the
      // shading model) is skipped.  FP32 estimate of log2 of the highlight, with 2 bits of margin.
      const V3 base = ambient + diffuse;
      const double base_min = fmin(base.x, fmin(base.y, base.z));
      const double gain = fmax(fabs(intensity.x), fmax(fabs(intensity.y), fabs(intensity.z))) * fabs(ob.specular);
      const float sh = (float)ob.shininess;
      const float lg_factor = sh * __log2f((float)rde) + 2.0f;        // -inf for rde == 0
      // (the hardware reciprocal is good to an ulp; the margin is two bits)
      const float lg_room = __log2f((float)base_min * __builtin_amdgcn_rcpf((float)gain)) - 55.0f;  // NaN or -inf when there is no room
      const bool negligible = sh > 0.0f && sh < 1.0e6f && rde <= 1.0 && base_min > 0.0 && lg_factor < lg_room;
      if (!negligible) {
        double factor = highlight_power(rde, ob.shininess);
        specular = (intensity * ob.specular) * factor;
      }
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
    cosv = sqrt_f64(1.0 - sin_t2);
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

__device__ __forceinline__ float wave_uniform(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); }

// value of lane `src` (wavefront-uniform index) in every lane, through an SGPR: no memory round trip
__device__ __forceinline__ float lane_value(float v, uint32_t src) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), (int)src));
}

// Reductions over the 64 lanes of a wavefront on the data-parallel-primitive path of the vector ALU
// (no LDS round trips): an inclusive scan inside each row of 16 lanes (row_shr 1, 2, 4, 8), the row
// totals carried into the rows above (row_bcast 15 into rows 1 and 3, row_bcast 31 into rows 2 and 3),
// and the total read from lane 63.  A lane with no source inside its row, or outside the enabled rows,
// receives `identity`.  Every lane of the wavefront must be active; the result is wavefront-uniform.
template <int CTRL, int ROWS>
__device__ __forceinline__ int dpp_from(int identity, int v) {
  return __builtin_amdgcn_update_dpp(identity, v, CTRL, ROWS, 0xf, false);
}
template <typename T, typename Op>
__device__ __forceinline__ T wave_reduce(T value, T identity, Op op) {
  static_assert(sizeof(T) == 4, "one dword per lane");
  const int id = __builtin_bit_cast(int, identity);
  auto step = [&](int moved) { value = op(value, __builtin_bit_cast(T, moved)); };
  step(dpp_from<0x111, 0xf>(id, __builtin_bit_cast(int, value)));
  step(dpp_from<0x112, 0xf>(id, __builtin_bit_cast(int, value)));
  step(dpp_from<0x114, 0xf>(id, __builtin_bit_cast(int, value)));
  step(dpp_from<0x118, 0xf>(id, __builtin_bit_cast(int, value)));
  step(dpp_from<0x142, 0xa>(id, __builtin_bit_cast(int, value)));
  step(dpp_from<0x143, 0xc>(id, __builtin_bit_cast(int, value)));
  return __builtin_bit_cast(T, __builtin_amdgcn_readlane(__builtin_bit_cast(int, value), 63));
}
__device__ __forceinline__ float wave_sum_f(float v) {
  return wave_reduce(v, 0.0f, [](float a, float b) { return a + b; });
}
__device__ __forceinline__ float wave_max_f(float v) {
  return wave_reduce(v, -3.0e38f, [](float a, float b) { return fmaxf(a, b); });
}

// Ray and test counts of one wavefront.  The numbers are wavefront-uniform (they live in scalar
// registers): a site adds the NUMBER OF LANES for which its predicate holds -- a ballot and a
// population count -- instead of every lane incrementing a register of its own that is summed over
// the lanes at the end.  A lane that never took part (a pixel outside the frame, a node slot without
// a node, a lane masked off around the site) contributes nothing by construction, whatever its
// vector registers or spill slots hold.  Every addition must stand in wavefront-uniform control flow
// (the predicate carries the per-lane condition), or the sum would differ from lane to lane.
struct WaveCount {
  unsigned primary = 0, shadow = 0, secondary = 0, exact = 0;
};
__device__ __forceinline__ unsigned lanes_where(bool p) { return (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(p)); }
// Sum over the lanes of a small per-lane amount (< 2^BITS; 0 for lanes that do not take part):
// one ballot per bit.
template <int BITS>
__device__ __forceinline__ unsigned lanes_total(unsigned amount) {
  unsigned total = 0;
#pragma unroll
  for (int b = 0; b < BITS; ++b) total += lanes_where(((amount >> b) & 1u) != 0u) << b;
  return total;
}
// bits needed for an amount of at most n
constexpr int bits_for(unsigned n) {
  int b = 0;
  for (; n != 0u; n >>= 1) ++b;
  return b;
}
// One lane of the wavefront adds the counts to its shard of the launch's counters.
__device__ __forceinline__ void flush_counts(DevCounters* shard, const WaveCount& c) {
  if ((threadIdx.x & 63u) == 0) {
    if (c.primary) atomicAdd(&shard->rays_primary, (unsigned long long)c.primary);
    if (c.shadow) atomicAdd(&shard->rays_shadow, (unsigned long long)c.shadow);
    if (c.secondary) atomicAdd(&shard->rays_secondary, (unsigned long long)c.secondary);
    if (c.exact) atomicAdd(&shard->exact_tests, (unsigned long long)c.exact);
  }
}


// `Camera::ray_for_pixel` (src/scene/cameras.rs:52-69) with the matrix inverse hoisted to
// the host: the unit direction of the ray through pixel (x, y); the origin is C.origin.
__device__ __forceinline__ V3 camera_ray_direction(const DevCamera& C, uint32_t x, uint32_t y) {
  const double x_offset = ((double)x + 0.5) * C.pixel_size;
  const double y_offset = ((double)y + 0.5) * C.pixel_size;
  const double world_x = C.half_width - x_offset;
  const double world_y = C.half_height - y_offset;
  V3 pixel = xform(C.inv, V3{world_x, world_y, -1.0}, 1.0);
  V3 dv = pixel - ld3(C.origin);
  double mag = sqrt_f64(dv.x * dv.x + dv.y * dv.y + dv.z * dv.z);
  return divide_by_length(dv, mag);
}

// Framebuffer store of one 32x8 tile (local tile k), transposed through LDS so that every
// wavefront writes whole 128-byte (RGBA8) / 384-byte (FP32 RGB) tile rows
// (src/graphics.rs:44-46, 59-74).  All 256 threads of the workgroup call it; s_rgb holds 768
// floats, s_px 256 words.  Three output layouts: the row-major frame; a rank's tile-major
// compact buffer (tile mode); a rank's own tile rows one after the other, row-major at the
// frame's width (band mode).
__device__ __forceinline__ void store_tile(const DevCamera& C, const RenderTarget& T, uint32_t k, V3 color, float* s_rgb,
                                           uint32_t* s_px) {
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  const uint32_t lx = (wave << 3) | (lane & 7u), ly = lane >> 3;
  const uint32_t slot = ly * 32u + lx;
  s_rgb[slot * 3 + 0] = (float)color.x;
  s_rgb[slot * 3 + 1] = (float)color.y;
  s_rgb[slot * 3 + 2] = (float)color.z;
  s_px[slot] = cast_u8(color.x) | (cast_u8(color.y) << 8) | (cast_u8(color.z) << 16) | 0xff000000u;
  __syncthreads();
  // the tile's place is worked out again here (from a copy of k the optimiser cannot see through)
  // rather than kept in scalar registers across the whole trace
  const TileRef tr = tile_ref(T, k);
  const bool tile_major = T.layout == 1u;
  // first output row of this tile: its own in the frame, or its place among the rank's tile rows
  const uint32_t out_row0 = (T.layout == 2u ? tr.lrow : tr.ty) * 8u;
  const uint32_t rx = threadIdx.x & 31u, ry = threadIdx.x >> 5;
  const uint32_t gx = tr.tx * 32u + rx, gy = tr.ty * 8u + ry;
  if (T.rgba8) {
    uint32_t* out = reinterpret_cast<uint32_t*>(T.rgba8);
    if (tile_major)
      out[(size_t)k * 256u + threadIdx.x] = s_px[threadIdx.x];
    else if (gx < C.width && gy < C.height)
      out[(size_t)gx + (size_t)(out_row0 + ry) * C.width] = s_px[threadIdx.x];
  }
  if (T.rgb) {
#pragma unroll
    for (uint32_t pass = 0; pass < 3; ++pass) {
      const uint32_t e = threadIdx.x + pass * 256u;  // element of the tile's [8][96] float image
      const uint32_t row = e / 96u, col = e % 96u;
      if (tile_major) {
        T.rgb[(size_t)k * 768u + e] = s_rgb[e];
      } else {
        const uint32_t fx = tr.tx * 32u + col / 3u, fy = tr.ty * 8u + row;
        if (fx < C.width && fy < C.height) T.rgb[((size_t)fx + (size_t)(out_row0 + row) * C.width) * 3u + col % 3u] = s_rgb[e];
      }
    }
  }
}

// Workgroups of ONE wavefront that own an 8x8 patch, a quarter of a 32x8 tile (kernels whose
// wavefronts cost very different amounts: as one workgroup per tile the fast wavefronts of a tile would
// wait for the slow one, at the store's barrier and for the release of the workgroup's resources).
// Workgroups go round the eight XCDs, each with an L2 of its own; the quarters of a tile are given to
// consecutive workgroups OF ONE XCD, so that its L2 sees the whole 128- and 384-byte tile rows before
// it writes them back.  Launch patch_groups(n_tiles) workgroups of 64 threads; those whose tile is
// >= n_tiles have nothing to do.
constexpr uint32_t kXcds = 8;
struct PatchOfGroup {
  uint32_t tile, quarter;
};
__device__ __forceinline__ PatchOfGroup patch_of_group() {
  // workgroup b runs on XCD b % 8; its XCD's workgroups 4t .. 4t + 3 are the quarters of one tile
  const uint32_t xcd = blockIdx.x % kXcds, slot = blockIdx.x / kXcds;
  return {(slot >> 2) * kXcds + xcd, slot & 3u};
}
inline uint32_t patch_groups(uint32_t n_tiles) { return (n_tiles + kXcds - 1u) / kXcds * kXcds * 4u; }

// The wavefront's own 64 pixels of local tile k, straight from the lanes (rows of 8: 32 bytes of RGBA8,
// 96 of FP32); layouts as in store_tile.
__device__ __forceinline__ void store_patch(const DevCamera& C, const RenderTarget& T, uint32_t k, uint32_t quarter, V3 color) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t lx = (quarter << 3) | (lane & 7u), ly = lane >> 3;
  const TileRef tr = tile_ref(T, k);
  const uint32_t x = tr.tx * 32u + lx, y = tr.ty * 8u + ly;
  size_t at;  // pixel index in the output
  if (T.layout == 1u) {
    at = (size_t)k * 256u + ly * 32u + lx;
  } else {
    if (x >= C.width || y >= C.height) return;
    at = (size_t)x + (size_t)((T.layout == 2u ? tr.lrow : tr.ty) * 8u + ly) * C.width;
  }
  if (T.rgba8) reinterpret_cast<uint32_t*>(T.rgba8)[at] = cast_u8(color.x) | (cast_u8(color.y) << 8) | (cast_u8(color.z) << 16) | 0xff000000u;
  if (T.rgb) {
    float* out = T.rgb + at * 3u;
    out[0] = (float)color.x, out[1] = (float)color.y, out[2] = (float)color.z;
  }
}

}  // namespace rtk
#endif
