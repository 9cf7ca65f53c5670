// rtk_trace.h -- one node of a ray tree (`Scene::trace`, src/scene.rs:136-296) and the depth-first
// walk of a whole tree by one lane.  Device code only (included by .hip files).
#ifndef RTK_TRACE_H
#define RTK_TRACE_H

#include "rtk_shading.h"
#include "rtk_sweeps.h"

#pragma clang fp contract(off)

namespace rtk {

// ------------------------------------------------------------------------------------
// `Scene::trace` (src/scene.rs:136-296) for one ray per thread, evaluated iteratively:
// child colours enter their parent linearly with scalar factors (reflectivity,
// transparency, Schlick reflectance), so each node adds weight * surface to the pixel and
// pushes its children with the product weight.  SECONDARY = false would remove the stack;
// every kernel built from this file traces secondary rays.
struct StackEntry {
  double ox, oy, oz, dx, dy, dz, weight;
  int depth;
};

// Where a node's results go.  trace_step reports, in the order the iterative evaluation adds them to
// the pixel: a miss (ambient * weight); for a hit, each child that may not be traced any more because
// the depth limit is reached (ambient * its weight: refraction, then reflection), each child that is
// to be traced (refraction, then reflection), children_done() once the node's children are all
// reported (whether or not it has any), and at the end the lit surface colour times the node's weight.  StackSink is the depth-first evaluation by one lane: terms are added to the pixel's colour
// as they come, children go on the lane's stack (so the reflected subtree is traced before the
// refracted one).
struct StackSink {
  StackEntry* stack;
  int& sp;
  V3& color;
  __device__ __forceinline__ void term(V3 v, double w) { color = color + v * w; }
  __device__ __forceinline__ void untraced_child(bool, V3 ambient, double w) { color = color + ambient * w; }
  __device__ __forceinline__ void children_done() {}
  __device__ __forceinline__ void child(bool, V3 o, V3 d, double w, int depth) {
    StackEntry& e = stack[sp++];
    e.ox = o.x, e.oy = o.y, e.oz = o.z;
    e.dx = d.x, e.dy = d.y, e.dz = d.z;
    e.weight = w;
    e.depth = depth;
  }
};

// Scenes of a handful of objects (the reference's bundled scene has three): every object is
// intersected exactly, in object order; no filter records, no clusters.
constexpr uint32_t kBruteObjects = 8;
template <bool TRACK>
__device__ __forceinline__ void closest_brute(const DevScene& S, bool active, V3 o, V3 d, Closest& best, bool& graze) {
  closest_init(best);
  constexpr bool track = TRACK;
  if (!active) return;
  for (uint32_t obj = 0; obj < S.n_objects; ++obj) {
    double t0, t1;
    const int n = intersect_object(S.objects[obj], o, d, 0.0, t0, t1);
    if (n >= 1) closest_consider(best, t0, obj, track);
    if (n == 2) closest_consider(best, t1, obj, track);
    if (n == 2 && track && fabs(t0 - t1) < 0.00001) graze = true;
  }
}

// n1 of `LightingData::compute_refractivity` (src/scene/lighting.rs:81-116) by the letter, for the
// rays whose hit list holds a sphere with both roots within 1e-5 of each other.  The reference
// walks the sorted list (stable: ties in object order, a sphere's smaller root first) up to the
// first entry that `Hit::eq` (same object, distances within 1e-5: src/scene.rs:354-364) calls the
// winner -- the winner itself, or the same sphere's other root just before it -- and keeps a list
// of "containers": every entry is pushed, except that an entry equal to one already there removes
// it and stays out, which only the two close roots of one sphere can do to each other.  n1 is the
// refractive index of the last container at that point, 1.0 without one.  (n2 is always the
// winner's own index: nothing before the stopping entry equals it.)  Every object is intersected
// again, exactly; this runs for a handful of rays per frame at most.
__device__ __forceinline__ double literal_n1(const DevScene& S, V3 o, V3 d, uint32_t own, double t_win) {
  double b_t = t_win;
  uint32_t b_root = 0;
  {
    double t0, t1;
    const int n = intersect_object(S.objects[own], o, d, 0.0, t0, t1);
    if (n == 2 && t_win != t0) {  // the winner is the larger root
      b_root = 1;
      if (fabs(t0 - t1) < 0.00001) b_t = t0, b_root = 0;  // ... and its sibling already counts as the winner
    }
  }
  double last_t = -DBL_MAX;
  uint32_t last_obj = 0xffffffffu, last_root = 0;
  for (uint32_t obj = 0; obj < S.n_objects; ++obj) {
    double t[2];
    const int n = intersect_object(S.objects[obj], o, d, 0.0, t[0], t[1]);
    if (n == 0) continue;
    bool before[2] = {false, false};
    for (int r = 0; r < n; ++r)
      before[r] = t[r] < b_t || (t[r] == b_t && (obj < own || (obj == own && (uint32_t)r < b_root)));
    if (n == 2 && before[0] && before[1] && fabs(t[0] - t[1]) < 0.00001) continue;  // the second root removes the first
    for (int r = 0; r < n; ++r)
      if (before[r] && (last_obj == 0xffffffffu || t[r] > last_t ||
                        (t[r] == last_t && (obj > last_obj || (obj == last_obj && (uint32_t)r > last_root)))))
        last_t = t[r], last_obj = obj, last_root = (uint32_t)r;
  }
  return last_obj == 0xffffffffu ? 1.0 : S.objects[last_obj].refractivity;
}
// `Scene::is_shadowed` (src/scene.rs:208-225): one shadow ray per light until one is blocked.
__device__ __forceinline__ bool shadow_brute(const DevScene& S, bool active, V3 p, WaveCount& wc) {
  bool need = active, in_shadow = false;
  for (uint32_t k = 0; k < S.n_lights; ++k) {
    const unsigned n_need = lanes_where(need);
    if (n_need == 0u) break;
    wc.shadow += n_need;
    if (need) {
      const RTK_RO DevLight& L = S.lights[k];
      const V3 lv = ld3(L.pos) - p;
      const double lw = L.pos[3] - 1.0;
      const double m = sqrt_f64(lv.x * lv.x + lv.y * lv.y + lv.z * lv.z + lw * lw);
      double dw;
      const V3 d = divide_by_length(lv, lw, m, dw);
      for (uint32_t obj = 0; obj < S.n_objects; ++obj) {
        double t0, t1;
        const int n = intersect_object(S.objects[obj], p, d, dw, t0, t1);
        if ((n >= 1 && t0 > 0.0 && t0 < m) || (n == 2 && t1 > 0.0 && t1 < m)) {
          in_shadow = true, need = false;
          break;
        }
      }
    }
  }
  return in_shadow;
}

// The children of a hit: `reflected_color` and `refracted_color` (src/scene.rs:228-272) with the
// Schlick reflectance where the material both reflects and transmits (src/scene.rs:191-204).  n1: the
// refractive index on the ray's side of the surface.  A child that may still be traced goes to
// sink.child (refraction first), one cut off by the depth limit to sink.untraced_child.
template <typename Sink>
__device__ __forceinline__ void emit_children(const DevScene& S, const RTK_RO DevObject& ob, const HitGeom& g, double n1, double weight,
                                              int depth, Sink& sink) {
  const double reflectivity = ob.reflectivity, transparency = ob.transparency;
  const bool has_refl = !is_approx_zero(reflectivity);   // reflected_color early-out
  const bool has_refr = !is_approx_zero(transparency);   // refracted_color early-out
  if (!has_refl && !has_refr) return;
  const V3 scene_ambient = ld3(S.ambient);
  const double n2 = ob.refractivity;
  const double cos_i = dot3(g.eye, g.normal);
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
      double cos_t = sqrt_f64(1.0 - sin_t2);
      V3 dir = g.normal * (n_ratio + cos_i - cos_t) - g.eye * n_ratio;  // the reference's `+`
      if (child_traces)
        sink.child(false, g.under, dir, w_refr, depth + 1);
      else
        sink.untraced_child(false, scene_ambient, w_refr);  // trace_inner(depth >= MAX_DEPTH)
    }
  }
  if (has_refl) {  // src/scene.rs:228-241
    if (child_traces)
      sink.child(true, g.over, g.reflect, w_refl, depth + 1);
    else
      sink.untraced_child(true, scene_ambient, w_refl);
  }
}

// One node of the ray tree, `trace_inner` (src/scene.rs:141-153) for the ray (ro, rd), in two halves.
// find_hit: the closest hit and, where the hit's material reflects or transmits, the refractive index
// n1 on the ray's side of the surface.  shade_hit: `apply_lighting` (src/scene.rs:172-205) with its
// shadow rays for a ray that enters the pixel's colour with factor `weight`, the node's terms and
// children handed to `sink`.  trace_step runs one after the other; the level-by-level order runs them
// as two kernels, each with the registers of its half only.  camera_sweep: the ray is a camera ray
// and the per-camera records may be used for its sweep.
struct NodeHit {
  double t, n1;
  uint32_t obj;  // 0xffffffff: the ray leaves the scene
};
template <bool PRIMARY_FILTER, bool WIDE, bool BRUTE, bool TRACK>
__device__ __forceinline__ NodeHit find_hit(const DevScene& S, const float4* __restrict__ prim, const float4* __restrict__ prim_cluster,
                                            bool have, bool camera_sweep, V3 ro, V3 rd, int depth, WaveCount& wc) {
  {
    const unsigned n_camera = lanes_where(have && depth == 0);
    wc.primary += n_camera;
    wc.secondary += lanes_where(have) - n_camera;
  }
  Closest best;
  bool graze = false;  // some sphere has both roots within 1e-5 of each other on this ray
  if (BRUTE)
    closest_brute<TRACK>(S, have, ro, rd, best, graze);
  else if (PRIMARY_FILTER && camera_sweep)
    closest_primary<TRACK>(S, prim, prim_cluster, have, ro, rd, best, wc, graze);
  else
    closest_secondary<WIDE, TRACK>(S, have, ro, rd, best, wc, graze);
  NodeHit h;
  h.t = best.t, h.n1 = 1.0;
  h.obj = have ? best.obj : 0xffffffffu;
  if constexpr (TRACK) {
    // n1: material of the entry just before the hit in the sorted list, else 1.0 (1.0 either way when
    // every refractivity in the scene is 1.0, track_n1 == 0).  Two roots within 1e-5 of each other
    // make the reference's container list drop entries: then by the letter.
    if (h.obj != 0xffffffffu) {
      const RTK_RO DevObject& ob = S.objects[h.obj];
      if (!is_approx_zero(ob.reflectivity) || !is_approx_zero(ob.transparency)) {
#if defined(RTK_WALK_STATS) && RTK_WALK_STATS == 4  // experiment build: exact_tests counts the rays that take the literal evaluation
        wc.exact += lanes_where(graze);
#endif
        if (graze)
          h.n1 = literal_n1(S, ro, rd, h.obj, best.t);
        else if (best.prev_obj != 0xffffffffu)
          h.n1 = S.objects[best.prev_obj].refractivity;
      }
    }
  }
  return h;
}

template <bool SECONDARY, int LB, bool WIDE, bool BRUTE, int SHADOWS = kShadowAny, typename Sink>
__device__ __forceinline__ void shade_hit(const DevScene& S, bool have, const NodeHit& h, V3 ro, V3 rd, double weight, int depth,
                                          Sink& sink, WaveCount& wc) {
  const V3 scene_ambient = ld3(S.ambient);
  const bool hit = have && h.obj != 0xffffffffu;
  if (have && !hit) sink.term(scene_ambient, weight);  // trace_inner: miss -> ambient_color

  // --- apply_lighting (src/scene.rs:172-205) ---
  V3 over = {0, 0, 0}, normal = {0, 0, 1}, eye = {0, 0, 1}, tex = {0, 0, 0};
  const uint32_t obj = hit ? h.obj : 0u;
  if (hit) {
    const RTK_RO DevObject& ob = S.objects[obj];
    HitGeom g = lighting_data(ob, ro, rd, h.t);
    over = g.over;
    normal = g.normal;
    eye = g.eye;
    tex = sample_texture_at(S, ob, obj, g.objp);
    // children first, so that nothing of them stays live across the shadow sweep
    if (SECONDARY) emit_children(S, ob, g, h.n1, weight, depth, sink);
  }
  sink.children_done();  // (every lane comes by here, with or without a hit)

#if defined(RTK_NODE_CUT) && RTK_NODE_CUT >= 1  // experiment builds (wrong images): a node without its shadow rays
  const bool in_shadow = false;
#else
  const bool in_shadow = BRUTE ? shadow_brute(S, hit, over, wc) : shadow_sweeps<LB, WIDE, SHADOWS>(S, hit, over, wc);
#endif

  if (hit) {
    const RTK_RO DevObject& ob = S.objects[obj];
    V3 surface = scene_ambient;
#if !defined(RTK_NODE_CUT) || RTK_NODE_CUT < 2  // ... and without its lights
    for (uint32_t k = 0; k < S.n_lights; ++k)
      surface = surface + phong_lighting(S.lights[k], ob, tex, over, eye, normal, in_shadow);
#endif
    sink.term(surface, weight);
  }
}

template <bool SECONDARY, bool PRIMARY_FILTER, int LB, bool WIDE, bool BRUTE, bool TRACK, typename Sink>
__device__ __forceinline__ void trace_step(const DevScene& S, const float4* __restrict__ prim,
                                           const float4* __restrict__ prim_cluster, bool have, bool camera_sweep, V3 ro, V3 rd,
                                           double weight, int depth, Sink& sink, WaveCount& wc) {
  const NodeHit h = find_hit<PRIMARY_FILTER, WIDE, BRUTE, TRACK>(S, prim, prim_cluster, have, camera_sweep, ro, rd, depth, wc);
  shade_hit<SECONDARY, LB, WIDE, BRUTE>(S, have, h, ro, rd, weight, depth, sink, wc);
}

// `Scene::trace` (src/scene.rs:136-138) for one ray per lane: the tree is walked with a stack per lane,
// every lane of the wavefront taking one node per round until all stacks are empty.
template <bool SECONDARY, bool PRIMARY_FILTER, int LB, bool WIDE, bool BRUTE, bool TRACK>
__device__ __forceinline__ V3 trace_ray(const DevScene& S, const float4* __restrict__ prim,
                                        const float4* __restrict__ prim_cluster, bool valid, V3 o,
                                        V3 d, WaveCount& wc) {
  V3 color = {0, 0, 0};
  StackEntry stack[SECONDARY ? kMaxDepth + 1 : 1];
  int sp = 0;
  // current ray
  V3 ro = o, rd = d;
  double weight = 1.0;
  int depth = 0;
  bool have = valid && S.max_depth > 0;
  if (valid && S.max_depth == 0) color = ld3(S.ambient);  // trace_inner(depth 0 >= MAX_DEPTH)

  StackSink sink{stack, sp, color};
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
      if (!__any(have)) break;  // the wavefront has run out of rays
    } else if (iter > 0) {
      break;
    }
    trace_step<SECONDARY, PRIMARY_FILTER, LB, WIDE, BRUTE, TRACK>(S, prim, prim_cluster, have, iter == 0, ro, rd, weight, depth, sink, wc);
  }
  return color;
}

}  // namespace rtk
#endif
