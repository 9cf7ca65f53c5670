// rtk_sweeps.h -- object sweeps with one ray (or one shading point) per lane, for rays that
// are not bundled per wavefront: secondary rays, shadow rays with per-lane light directions,
// caller-supplied rays, and the camera rays of the single-kernel path.  Device code only
// (included by .hip files after rtk_shading.h).
//
// The spheres are stored in spatial clusters of kBundle with one bounding ball each
// (rtk_scene_upload).  A sweep walks the clusters; each lane tests ITS ray against the
// cluster's ball, and the wavefront opens a cluster only when some lane's ray can reach it.
// Inside an opened cluster the per-ray filter runs over groups of kGroup sphere records,
// then f64 on the candidates.  Records are wavefront-uniform loads from global memory: no
// LDS staging and no workgroup barriers.
#ifndef RTK_SWEEPS_H
#define RTK_SWEEPS_H

#include "rtk_shading.h"

#pragma clang fp contract(off)

namespace rtk {

// An opened cluster is swept sphere-per-lane when at most this many of the 64 rays reach it
// (and it holds enough spheres for 64 lanes to be worth occupying).  Kernels are built with
// and without that path (WIDE): small scenes never need it and run faster without its code.
constexpr int kTransposeBelow = 24;

// ------------------------------------------------------------------------------------
// Camera rays: all rays of a frame leave from the camera origin, so the per-sphere part
// of the test is hoisted into the record {L = c - o, CC = |L|^2 (1 - slack) - R'^2}.
// With a unit direction d the line can touch the bounding ball only if (L.d)^2 >= CC:
// 3 multiply-adds, one fused square-and-subtract and a share of a max per test.
// Cluster record {L, Rs} (Rs carries its own rounding slack): the line misses the cluster
// when its distance from L exceeds Rs.
template <typename ExactFn>
__device__ __forceinline__ void sweep_primary(const DevScene& S, const float4* __restrict__ prim,
                                              const float4* __restrict__ prim_cluster, bool active, float dx, float dy, float dz,
                                              WaveCount& wc, ExactFn&& exact) {
  if (!__any(active)) return;
  for (uint32_t first = 0; first < S.n_spheres_grouped; first += kBundle) {
    const float4 cl = prim_cluster[first / kBundle];
    const float s = __builtin_fmaf(cl.z, dz, __builtin_fmaf(cl.y, dy, cl.x * dx));
    const float qx = __builtin_fmaf(-s, dx, cl.x), qy = __builtin_fmaf(-s, dy, cl.y), qz = __builtin_fmaf(-s, dz, cl.z);
    const float d2 = __builtin_fmaf(qz, qz, __builtin_fmaf(qy, qy, qx * qx));
    const bool reach = active && cl.w * cl.w * (1.0f + 8.0f * kEps32) >= d2;
    if (!__any(reach)) continue;
    const uint32_t end = min(first + (uint32_t)kBundle, S.n_spheres_grouped);
    for (uint32_t j = first; j < end; j += kGroup) {
      float4 q[kGroup];
#pragma unroll
      for (int g = 0; g < kGroup; ++g) q[g] = prim[j + g];
      float m = -1.0f;
#pragma unroll
      for (int g = 0; g < kGroup; ++g) {
        float t = q[g].x * dx;
        t = __builtin_fmaf(q[g].y, dy, t);
        t = __builtin_fmaf(q[g].z, dz, t);
        m = fmaxf(m, __builtin_fmaf(t, t, -q[g].w));
      }
      if (__any(reach && m >= 0.0f)) {
        // leave the fast loop: find which of the eight records kept a candidate
#pragma unroll 1
        for (int g = 0; g < kGroup; ++g) {
          float4 r = prim[j + g];
          float t = r.x * dx;
          t = __builtin_fmaf(r.y, dy, t);
          t = __builtin_fmaf(r.z, dz, t);
          bool pass = reach && (__builtin_fmaf(t, t, -r.w) >= 0.0f);
          const unsigned n_pass = lanes_where(pass);
          if (n_pass) {
            wc.exact += n_pass;
            uint32_t obj = S.sph_object[j + g];
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
//
// Cluster test.  The inequality says the line passes within
// R_e = R sqrt(1 + 2 |rho| max(-s, 0) + R^2 rho^2) of c, s = u.g (rho <= 0 for lights with
// w in [0, 1]).  For a cluster ball {C, Rc} around its members' balls, with s_C = (p - C).g:
//     dist(C, line) <= Rc + r_max |rho| max(Rc - s_C, 0) + r_max^3 rho^2 / 2
// is necessary for any member to be touched (r_max = largest sphere radius of the scene).  The same
// holds for any ball that contains the members' balls: the balls of whole groups of clusters.
// exact(slot, mask) runs the f64 tests of one lane's candidate and returns how many it ran (at most
// NR); the sweep adds them to wc.exact.
template <int NR, bool WIDE, typename ExactFn, typename DoneFn>
__device__ __forceinline__ void sweep_filter(const DevScene& S, bool& active, float px, float py, float pz,
                                             const float (&gx)[NR], const float (&gy)[NR], const float (&gz)[NR],
                                             const float (&rho)[NR], float thr, WaveCount& wc, ExactFn&& exact, DoneFn&& done,
                                             bool forward = false) {
  // forward: only hits at positive ray parameters matter to the caller (NR == 1, rho == 0): a
  // cluster that lies entirely behind the origin is not opened
  bool wave_live = __any(active);
  const bool cull = thr > -1.0e38f;  // rays of lights with w outside [0, 1] look at every sphere
  float lrho[NR];
  float grow = 0.0f;
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    lrho[r] = fabsf(rho[r]) * 1.0001f;
    grow = fmaxf(grow, 0.5f * S.r_max * S.r2_max * rho[r] * rho[r]);
  }
  // covers the FP32 rounding of the point, the centres, the directions and the arithmetic below
  const float pad = grow * 1.01f + 16.0f * kEps32 * (fabsf(px) + fabsf(py) + fabsf(pz) + S.c_max);
  // can any of this lane's rays touch a sphere inside the ball {C, Rc}?
  auto may_reach = [&](const float4 cl) {
    const float ux = px - cl.x, uy = py - cl.y, uz = pz - cl.z;
    const float room = cl.w + pad;
    bool pass = false;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const float s = __builtin_fmaf(uz, gz[r], __builtin_fmaf(uy, gy[r], ux * gx[r]));
      const float qx = __builtin_fmaf(-s, gx[r], ux), qy = __builtin_fmaf(-s, gy[r], uy), qz = __builtin_fmaf(-s, gz[r], uz);
      const float d2 = __builtin_fmaf(qz, qz, __builtin_fmaf(qy, qy, qx * qx));
      const float t = fmaxf(cl.w - s, 0.0f) * (1.0f + 8.0f * kEps32);
      const float rr = __builtin_fmaf(t, S.r_max * lrho[r], room) * (1.0f + 4.0f * kEps32);
      // s = (p - C).g: beyond room, every point of the ball has a negative ray parameter
      const bool behind = forward && s > room;
      pass |= !behind && rr >= 0.0f && __builtin_fmaf(rr, rr, -d2) >= 0.0f;
    }
    return pass;
  };
  // Two levels: kSuper consecutive clusters (a subtree of the k-d split that ordered the spheres)
  // share one ball; a wavefront none of whose rays can reach it skips all of them at once.
  constexpr uint32_t kSuperSpheres = kSuper * kBundle;
  for (uint32_t super_first = 0; wave_live && super_first < S.n_spheres_grouped; super_first += kSuperSpheres) {
    if (S.n_spheres_grouped > kSuperSpheres) {
      const bool reach_super = active && (!cull || may_reach(S.super_ball[super_first / kSuperSpheres]));
      if (!__any(reach_super)) continue;
    }
    const uint32_t super_end = min(super_first + kSuperSpheres, S.n_spheres_grouped);
    for (uint32_t first = super_first; wave_live && first < super_end; first += kBundle) {
      bool reach = active;
      if (cull) reach = active && may_reach(S.cluster_ball[first / kBundle]);
      const unsigned long long reach_mask = __builtin_amdgcn_ballot_w64(reach);
      if (reach_mask == 0ull) continue;
      if (WIDE && __builtin_popcountll(reach_mask) <= kTransposeBelow && S.n_spheres_grouped - first >= 32u) {
        // Few rays reach this cluster (incoherent rays): turn the loop around.  Every lane takes
        // ONE sphere of the cluster; the reaching rays are broadcast one after the other and
        // tested against all 64 spheres at once; each ray's owner keeps the set of spheres that
        // passed and re-evaluates those per light before the f64 test.
        const uint32_t lane = threadIdx.x & 63u;
        const float4 mine = S.sph_filter[first + lane];       // padded to kBundle: always in range
        const bool real = first + lane < S.n_spheres;         // padding repeats the last sphere
        unsigned long long cand = 0ull;
        for (unsigned long long rm = reach_mask; rm != 0ull; rm &= rm - 1ull) {
          const uint32_t src = (uint32_t)__builtin_amdgcn_readfirstlane((int)__builtin_ctzll(rm));
          const float qx = lane_value(px, src), qy = lane_value(py, src), qz = lane_value(pz, src);
          const float qthr = lane_value(thr, src);
          const float ux = qx - mine.x, uy = qy - mine.y, uz = qz - mine.z;
          const float uu = __builtin_fmaf(uz, uz, __builtin_fmaf(uy, uy, ux * ux));
          const float cf = __builtin_fmaf(uu, kKappa, -mine.w);
          float m = -3.0e38f;
#pragma unroll
          for (int r = 0; r < NR; ++r) {
            const float h = __builtin_fmaf(uz, lane_value(gz[r], src),
                                           __builtin_fmaf(uy, lane_value(gy[r], src),
                                                          __builtin_fmaf(ux, lane_value(gx[r], src), mine.w * lane_value(rho[r], src))));
            m = fmaxf(m, __builtin_fmaf(h, h, -cf));
          }
          const unsigned long long kept = __builtin_amdgcn_ballot_w64(real && m >= qthr);
          if (lane == src) cand = kept;
        }
        // every lane now works through its own candidates; lanes run different spheres side by side
        while (__any(cand != 0ull)) {
          unsigned ran = 0;
          if (cand != 0ull) {
            const uint32_t b = (uint32_t)__builtin_ctzll(cand);
            cand &= cand - 1ull;
            uint32_t mask = 1u;  // one ray per lane: the sphere passed that ray's test above
            if (NR > 1) {        // several: which of them
              const float4 c = S.sph_filter[first + b];
              const float ux = px - c.x, uy = py - c.y, uz = pz - c.z;
              const float uu = __builtin_fmaf(uz, uz, __builtin_fmaf(uy, uy, ux * ux));
              const float cf = __builtin_fmaf(uu, kKappa, -c.w);
              mask = 0;
#pragma unroll
              for (int r = 0; r < NR; ++r) {
                const float h = __builtin_fmaf(uz, gz[r], __builtin_fmaf(uy, gy[r], __builtin_fmaf(ux, gx[r], c.w * rho[r])));
                if (__builtin_fmaf(h, h, -cf) >= thr) mask |= 1u << r;
              }
            }
            if (mask) ran = exact(first + b, mask);
            if (active && done()) active = false, cand = 0ull;
          }
          wc.exact += lanes_total<bits_for(NR)>(ran);
        }
        wave_live = __any(active);
        continue;
      }
      const uint32_t end = min(first + (uint32_t)kBundle, S.n_spheres_grouped);
      for (uint32_t j = first; j < end; j += kGroup) {
        float4 q[kGroup];
#pragma unroll
        for (int g = 0; g < kGroup; ++g) q[g] = S.sph_filter[j + g];
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
        if (__any(reach && m >= thr)) {
          unsigned ran = 0;  // exact tests of this lane over the group
#pragma unroll 1
          for (int g = 0; g < kGroup; ++g) {
            float4 c = S.sph_filter[j + g];
            float ux = px - c.x, uy = py - c.y, uz = pz - c.z;
            float uu = __builtin_fmaf(uz, uz, __builtin_fmaf(uy, uy, ux * ux));
            float cf = __builtin_fmaf(uu, kKappa, -c.w);
            uint32_t mask = 0;
#pragma unroll
            for (int r = 0; r < NR; ++r) {
              float h = __builtin_fmaf(uz, gz[r], __builtin_fmaf(uy, gy[r], __builtin_fmaf(ux, gx[r], c.w * rho[r])));
              if (__builtin_fmaf(h, h, -cf) >= thr) mask |= 1u << r;
            }
            if (!reach) mask = 0;
            if (mask) ran += exact(j + g, mask);
          }
          wc.exact += lanes_total<bits_for(NR * kGroup)>(ran);
          // lanes that need no further test drop out; the wavefront stops once all have
          if (active && done()) active = false, reach = false;
          wave_live = __any(active);
          if (!wave_live) break;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------
// Shadow test for one shading point: `Scene::is_shadowed` (src/scene.rs:208-225).
// Lights are swept in batches of LB fused rays; the reference stops at the first blocked
// light, so `n_rays` = index of the first blocked light + 1 (or the light count).
template <int LB, bool WIDE>
__device__ __forceinline__ bool shadow_by_sweeps(const DevScene& S, bool active, V3 p, WaveCount& wc) {
  bool in_shadow = false;
  bool need = active;
  for (uint32_t kb = 0; kb < S.n_lights; kb += LB) {
    if (!__any(need)) break;  // nobody in the wavefront still needs a light
    const uint32_t nb = min((uint32_t)LB, S.n_lights - kb);
    uint32_t occl = 0;
    // planes: exact, per light (the shadow ray is rebuilt once per light)
    if (need) {
      for (uint32_t ip = 0; ip < S.n_planes; ++ip) {
        const RTK_RO DevObject& ob = S.objects[S.plane_object[ip]];
        const PlaneSide side = plane_side(ob, p);
        for (uint32_t r = 0; r < nb; ++r) {
          const RTK_RO DevLight& L = S.lights[kb + r];
          if (!plane_may_block(side, S.plane_light[ip * S.n_lights + kb + r])) continue;
          V3 lv = ld3(L.pos) - p;
          double lw = L.pos[3] - 1.0;
          double m = sqrt_f64(lv.x * lv.x + lv.y * lv.y + lv.z * lv.z + lw * lw);
          double dw;
          V3 d = divide_by_length(lv, lw, m, dw);
          double t0, t1;
          int n = intersect_object(ob, p, d, dw, t0, t1);
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
      const RTK_RO DevLight& L = S.lights[kb + min((uint32_t)r, nb - 1)];  // pad with the last real light
      double lw = L.pos[3];
      V3 G = ld3(L.pos) - p * lw;
      make_filter_ray(G, lw - 1.0, gx[r], gy[r], gz[r], rho[r]);
      rho_max = fmaxf(rho_max, fabsf(rho[r]) * 1.0001f);
      exotic |= !(lw >= 0.0 && lw <= 1.0) || !(dot3(G, G) > 0.0);  // the line form needs 1 + (w-1) t > 0 and G != 0
    }
    V3 pc = {p.x - S.center[0], p.y - S.center[1], p.z - S.center[2]};
    float thr = exotic ? -3.0e38f : filter_threshold(S, pc, rho_max);
    bool live = need && !(occl & 1u);
    auto exact = [&](uint32_t slot, uint32_t mask) -> unsigned {
      const RTK_RO SphRecord& rec = S.sph_record[slot];
      unsigned ran = 0;
      for (uint32_t r = 0; r < nb; ++r) {
        if (!((mask >> r) & 1u) || ((occl >> r) & 1u)) continue;
        const RTK_RO DevLight& L = S.lights[kb + r];
        V3 lv = ld3(L.pos) - p;
        double lw = L.pos[3] - 1.0;
        double m = sqrt_f64(lv.x * lv.x + lv.y * lv.y + lv.z * lv.z + lw * lw);
        double dw;
        V3 d = divide_by_length(lv, lw, m, dw);
        double t0, t1;
        int n = intersect_sphere_record(S, rec, p, d, dw, t0, t1);
        ++ran;
        if (n == 2 && ((t0 > 0.0 && t0 < m) || (t1 > 0.0 && t1 < m))) occl |= 1u << r;
      }
      return ran;
    };
    auto done = [&]() { return (occl & 1u) != 0; };
    sweep_filter<LB, WIDE>(S, live, (float)pc.x, (float)pc.y, (float)pc.z, gx, gy, gz, rho, thr, wc, exact, done);
    // the lights of the batch a point traces: up to and including the first blocked one
    unsigned traced = 0;
    if (need) {
      occl &= (1u << nb) - 1u;
      traced = nb;
      if (occl) {
        traced = (unsigned)__builtin_ctz(occl) + 1u;
        in_shadow = true;
        need = false;
      }
    }
    wc.shadow += lanes_total<bits_for(LB)>(traced);
  }
  return in_shadow;
}

// The same through the light grids (LightGrid in rtk_device.h; scenes whose lights all have w = 0):
// per light, the lane's point is projected along the light's direction, and the spheres listed in its
// cell go through the per-ray FP32 filter and, if kept, the exact test -- a few dozen instructions
// where the cluster sweep of 64 unrelated points costs thousands.  Lanes whose point lies below a
// grid's floor (far-away points of a plane) take the cluster sweeps.
template <int LB, bool WIDE>
__device__ __forceinline__ bool shadow_by_grids(const DevScene& S, bool active, V3 p, WaveCount& wc) {
  const V3 pc = {p.x - S.center[0], p.y - S.center[1], p.z - S.center[2]};
  const float px = (float)pc.x, py = (float)pc.y, pz = (float)pc.z;
  bool in_range = active;
  float rho_max = 0.0f;
  const float pp = __builtin_fmaf(pz, pz, __builtin_fmaf(py, py, px * px));
  for (uint32_t k = 0; k < S.n_lights; ++k) {
    const float4 lf = S.light_filter[k];
    const float h = __builtin_fmaf(pz, lf.z, __builtin_fmaf(py, lf.y, px * lf.x));
    in_range = in_range && h >= S.light_grid[k].h_min && pp <= S.light_grid[k].d_max2;
    rho_max = fmaxf(rho_max, fabsf(lf.w) * 1.0001f);
  }
#if defined(RTK_SHADOW_CUT) && RTK_SHADOW_CUT == 1  // experiment builds (wrong images): no point takes the cluster sweeps
  in_range = active;
#endif
#if defined(RTK_SHADOW_CUT) && RTK_SHADOW_CUT == 2  // ... no point looks at a grid's lists
  in_range = false;
  const bool swept = false;
  bool in_shadow = false;
#else
  bool in_shadow = false;
  const bool swept = active && !in_range;
#endif
  if (__any(swept)) {  // (one light at a time: the same rays, and fewer registers on a way few points take)
    const bool s = shadow_by_sweeps<1, WIDE>(S, swept, p, wc);
    if (swept) in_shadow = s;
  }
  const float thr = filter_threshold(S, pc, rho_max);
  bool need = in_range;
  for (uint32_t k = 0; k < S.n_lights; ++k) {
    if (!__any(need)) break;
    const RTK_RO DevLight& L = S.lights[k];
    bool blocked = false;
    // the reference's shadow ray towards this light (src/scene.rs:208-225)
    const V3 lv = ld3(L.pos) - p;
    const double lw = L.pos[3] - 1.0;
    const double m = sqrt_f64(lv.x * lv.x + lv.y * lv.y + lv.z * lv.z + lw * lw);
    double dw;
    const V3 d = divide_by_length(lv, lw, m, dw);
    wc.shadow += lanes_where(need);
    if (need) {
      for (uint32_t ip = 0; ip < S.n_planes; ++ip) {  // planes: exact
        const RTK_RO DevObject& ob = S.objects[S.plane_object[ip]];
        if (!plane_may_block(plane_side(ob, p), S.plane_light[ip * S.n_lights + k])) continue;
        double t0, t1;
        const int n = intersect_object(ob, p, d, dw, t0, t1);
        if (n >= 1 && t0 > 0.0 && t0 < m) blocked = true;
      }
    }
    const RTK_RO LightGrid& G = S.light_grid[k];
    const float4 lf = S.light_filter[k];
    uint32_t first = 0, count = 0;
    if (need && !blocked) {
      const float u = __builtin_fmaf(pz, G.e1[2], __builtin_fmaf(py, G.e1[1], px * G.e1[0]));
      const float v = __builtin_fmaf(pz, G.e2[2], __builtin_fmaf(py, G.e2[1], px * G.e2[0]));
      const float cu = (u - G.u0) * G.inv_cell, cv = (v - G.v0) * G.inv_cell;
      // outside the grid no sphere's shadow reaches a point at or above the floor
      if (cu >= 0.0f && cv >= 0.0f && cu < (float)G.nx && cv < (float)G.ny) {
        const uint2 c = S.grid_cells[G.cell_base + (uint32_t)cv * G.nx + (uint32_t)cu];
        first = c.x, count = c.y;
      }
    }
    // the listed spheres four at a time (their filter records lie side by side in the list, so the four
    // loads are in flight together): the per-ray filter, then f64 on those it keeps
    for (uint32_t base = 0; __any(base < count); base += 4u) {
      uint32_t kept = 0u, slot[4] = {0u, 0u, 0u, 0u};
      if (base < count) {
        const uint32_t here = min(count - base, 4u);
        float4 c[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const uint32_t e = first + base + min((uint32_t)i, here - 1u);  // a short batch repeats its last entry
          c[i] = S.grid_filter[e];
          slot[i] = S.grid_entries[e];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float ux = px - c[i].x, uy = py - c[i].y, uz = pz - c[i].z;
          const float uu = __builtin_fmaf(uz, uz, __builtin_fmaf(uy, uy, ux * ux));
          const float cf = __builtin_fmaf(uu, kKappa, -c[i].w);
          const float h = __builtin_fmaf(uz, lf.z, __builtin_fmaf(uy, lf.y, __builtin_fmaf(ux, lf.x, c[i].w * lf.w)));
          if ((uint32_t)i < here && __builtin_fmaf(h, h, -cf) >= thr) kept |= 1u << i;
        }
      }
      while (__any(kept != 0u)) {
        wc.exact += lanes_where(kept != 0u);
        if (kept != 0u) {
          const uint32_t i = (uint32_t)__builtin_ctz(kept);
          kept &= kept - 1u;
          const uint32_t at = i == 0u ? slot[0] : (i == 1u ? slot[1] : (i == 2u ? slot[2] : slot[3]));
          const RTK_RO SphRecord& rec = S.sph_record[at];
          double t0, t1;
          const int n = intersect_sphere_record(S, rec, p, d, dw, t0, t1);
          if (n == 2 && ((t0 > 0.0 && t0 < m) || (t1 > 0.0 && t1 < m))) blocked = true, kept = 0u, count = 0u;
        }
      }
    }
    if (need && blocked) in_shadow = true, need = false;
  }
  return in_shadow;
}

// ------------------------------------------------------------------------------------
// Closest hit of a general ray (origin o, direction d with w = 0, not necessarily unit):
// `Scene::intersect` + `HitList::closest_hit` (src/scene.rs:156-169, 325-340).
// TRACK: the scene holds different refractive indices, so the entry before the winner (n1) matters.
template <bool WIDE, bool TRACK>
__device__ __forceinline__ void closest_general(const DevScene& S, bool active, V3 o, V3 d, Closest& best,
                                                WaveCount& wc, bool& graze) {
  closest_init(best);
  constexpr bool track = TRACK;
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
  auto exact = [&](uint32_t slot, uint32_t) -> unsigned {
    const RTK_RO SphRecord& rec = S.sph_record[slot];
    const uint32_t obj = rec.object;
    double t0, t1;
    if (intersect_sphere_record(S, rec, o, d, 0.0, t0, t1)) {
      closest_consider(best, t0, obj, track);
      closest_consider(best, t1, obj, track);
      if (track && fabs(t0 - t1) < 0.00001) graze = true;  // `Hit::eq` would call the two roots one hit
    }
    return 1u;
  };
  auto done = [&]() { return false; };
  bool live = active;
  // hits behind the origin only matter for the refractive index n1 (the entry before the winner)
  sweep_filter<1, WIDE>(S, live, (float)pc.x, (float)pc.y, (float)pc.z, gx, gy, gz, rho, thr, wc, exact, done, !track);
}

// ------------------------------------------------------------------------------------
// The same for 64 rays that have nothing to do with each other (reflected and refracted rays queued
// from all over the frame), through the scene's cell grid: every lane walks ITS ray from cell to cell
// (a 3D DDA in cell units) and tests the spheres listed in each cell -- the per-ray FP32 filter, then
// f64 on what it keeps.  A wavefront-wide cluster sweep opens every cluster that any of its 64 rays
// comes near, which for unrelated rays is most of the scene; a walk looks at the few cells along one
// ray and stops at the first cell that begins behind the closest hit found.
//
// What makes stopping sound.  Lists hold every sphere whose bounding ball, grown by a margin of 3 % of a
// cell plus 1e-5 of the grid's size, reaches into the cell (rtk_scene_upload).  The walk starts from
// a point of the ray computed in f64 and keeps positions in cell units (at most 256 per axis), so the
// cells it visits differ from those of the exact ray only where the ray passes within about 1e-4 cells
// of a cell boundary; and the f64 roots of a sphere lie far less than the margin away from its ball.  So a
// sphere with a root in the stretch of the ray walked so far is listed in one of the cells visited,
// and a walk may end once the current cell's far side lies 0.05 cells behind the best root.  Rays
// that start further from the grid than 200,000 cells, or whose direction has no usable length, do not
// take the walk: the caller sweeps them.
//
// Lanes whose filter kept a sphere wait with it until 24 of them do, or until no lane has anything
// else to do: the f64 test then runs for all of them at once instead of once per lane and candidate.
// backward: the walk runs against the ray's direction.  t_stop: the ray parameter (in units of d, counted
// along the walk) behind which nothing matters to the caller -- the best root so far of a closest-hit
// search, the light's distance for a shadow ray; exact(sweep position) runs the caller's f64 test on a
// sphere the filter kept and returns the t_stop that holds from then on (negative: the lane's walk ends).
// Returns whether the lane's walk began in the cell of the ray's origin (the origin lies inside the grid's box).
// origin_cell_done: a walk of the same ray the other way has been through that cell already -- the filter does
// not depend on the direction's sign, so every sphere listed there that this walk would test has been tested
// -- and a walk that begins there passes it by.
template <typename ExactFn>
__device__ __forceinline__ bool grid_march(const DevScene& S, bool active, V3 o, V3 d, bool backward, double t_stop,
                                           WaveCount& wc, ExactFn&& exact, bool origin_cell_done = false) {
  const double dd = dot3(d, d);
  double rs = __builtin_amdgcn_rsq(dd);
  rs = rs * __builtin_fma(-0.5 * dd, rs * rs, 1.5);  // one Newton step: a ray that enters the grid from 200,000 cells away must still enter it at the right place
  const double sign = backward ? -1.0 : 1.0;
  const double len_h = dd * rs * S.tg_inv_h;   // ray parameter -> cell units
  const double oc[3] = {(o.x - S.tg_lo[0]) * S.tg_inv_h, (o.y - S.tg_lo[1]) * S.tg_inv_h, (o.z - S.tg_lo[2]) * S.tg_inv_h};
  const double g[3] = {d.x * rs * sign, d.y * rs * sign, d.z * rs * sign};
  // the stretch of the ray inside the grid's box, in cell units from the origin
  double s_in = 0.0, s_out = 1.0e300;
  bool alive = active;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const double n = (double)S.tg_n[a];
    if (fabs(g[a]) > 1.0e-9) {
      const double r = __builtin_amdgcn_rcp(g[a]);
      const double t0 = (0.0 - oc[a]) * r, t1 = (n - oc[a]) * r;
      s_in = fmax(s_in, fmin(t0, t1));
      s_out = fmin(s_out, fmax(t0, t1));
    } else if (!(oc[a] >= 0.0 && oc[a] <= n)) {
      alive = false;
    }
  }
  s_in = fmax(s_in - 0.01, 0.0);  // the reciprocals above are approximate: begin a little early
  alive = alive && s_in <= s_out + 0.01;
  // first cell, position inside it, and the walk's constants
  int ic[3], step[3];
  float e[3], inv[3], far[3], t_next[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const double ea = __builtin_fma(g[a], s_in, oc[a]);
    const int n = (int)S.tg_n[a];
    int i = (int)floor(ea);
    i = i < 0 ? 0 : (i >= n ? n - 1 : i);  // (a NaN lands in cell 0 of a lane that is not alive)
    ic[a] = i;
    e[a] = (float)ea;
    const float ga = (float)g[a];
    const bool moves = fabsf(ga) > 1.0e-9f;
    inv[a] = moves ? __builtin_amdgcn_rcpf(ga) : 0.0f;
    far[a] = moves ? 0.0f : 3.0e38f;  // an axis the ray does not move along is never the next to be crossed
    step[a] = ga > 0.0f ? 1 : -1;
    t_next[a] = __builtin_fmaf((float)(i + (step[a] > 0 ? 1 : 0)) - e[a], inv[a], far[a]);
  }
  // the FP32 filter of this ray (sweep_filter's, with rho = 0)
  float gx, gy, gz, rho_unused;
  make_filter_ray(d, 0.0, gx, gy, gz, rho_unused);
  const V3 pc = {o.x - S.center[0], o.y - S.center[1], o.z - S.center[2]};
  const float px = (float)pc.x, py = (float)pc.y, pz = (float)pc.z;
  const float thr = filter_threshold(S, pc, 0.0f);
  // t_stop in the walk's units (distance from the walk's first point, in cells); DBL_MAX stays beyond every cell
  auto in_cells = [&](double t) -> float { return (float)fmin(__builtin_fma(t, len_h, -s_in), 1.0e30); };
  float best_u = in_cells(t_stop);
  uint32_t first = 0, count = 0, base = 0;  // the current cell's list, and how much of it is done
  uint32_t kept = 0u, slot[4] = {0u, 0u, 0u, 0u}, seen0 = 0xffffffffu, seen1 = 0xffffffffu;
  int budget = (int)(S.tg_n[0] + S.tg_n[1] + S.tg_n[2]) + 4;  // cells a straight line can cross
  // the filter of one entry: does the ray come near the sphere?  (a ball lies in several cells: the last
  // two spheres tested are not tested again)
  auto keeps = [&](const float4 c, uint32_t at) {
    const float ux = px - c.x, uy = py - c.y, uz = pz - c.z;
    const float uu = __builtin_fmaf(uz, uz, __builtin_fmaf(uy, uy, ux * ux));
    const float cf = __builtin_fmaf(uu, kKappa, -c.w);
    const float h = __builtin_fmaf(uz, gz, __builtin_fmaf(uy, gy, ux * gx));
    return __builtin_fmaf(h, h, -cf) >= thr && at != seen0 && at != seen1;
  };
  // entering a cell: its record brings the first two entries of its list along
  auto enter_cell = [&]() {
    const RTK_RO TraceCell& cell = S.tg_cells[((uint32_t)ic[2] * S.tg_n[1] + (uint32_t)ic[1]) * S.tg_n[0] + (uint32_t)ic[0]];
    first = cell.first, count = cell.count, base = 2u;
    slot[0] = cell.slot[0], slot[1] = cell.slot[1];
    const float4 c0 = cell.filter[0], c1 = cell.filter[1];
    if (count >= 1u && keeps(c0, slot[0])) kept |= 1u;
    if (count >= 2u && keeps(c1, slot[1])) kept |= 2u;
  };
#if defined(RTK_HIT_CUT) && RTK_HIT_CUT == 1  // experiment builds (wrong images): no walk at all
  alive = false;
#endif
#if defined(RTK_HIT_CUT) && RTK_HIT_CUT == 2  // ... only the first cell
  budget = 0;
#endif
#if defined(RTK_HIT_BUDGET)  // ... at most that many cells
  budget = RTK_HIT_BUDGET;
#endif
  const bool at_origin = alive && s_in == 0.0;
  if (alive && !(origin_cell_done && at_origin)) enter_cell();  // (else: an empty list, and on to the cell's far side)
  [[maybe_unused]] int stat_rounds = 0;
  for (;;) {
    if (alive && kept == 0u) {
      if (base < count) {
        // the rest of a longer list, four entries at a time: their loads are in flight together
        const uint32_t here = min(count - base, 4u);
        float4 c[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const uint32_t en = first + base + min((uint32_t)i, here - 1u);  // a short batch repeats its last entry
          c[i] = S.tg_filter[en];
          slot[i] = S.tg_slot[en];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if ((uint32_t)i < here && keeps(c[i], slot[i])) kept |= 1u << i;
        base += 4u;
      } else {
        // done with this cell: stop behind the best root, else cross into the next cell
        const float u_exit = fminf(t_next[0], fminf(t_next[1], t_next[2]));
        if (best_u < u_exit - 0.05f || --budget < 0) {
          alive = false;
        } else {
          const int a = (t_next[0] <= t_next[1] && t_next[0] <= t_next[2]) ? 0 : (t_next[1] <= t_next[2] ? 1 : 2);
#pragma unroll
          for (int k = 0; k < 3; ++k)
            if (k == a) {
              ic[k] += step[k];
              t_next[k] = __builtin_fmaf((float)(ic[k] + (step[k] > 0 ? 1 : 0)) - e[k], inv[k], far[k]);
              if (ic[k] < 0 || ic[k] >= (int)S.tg_n[k]) alive = false;
            }
          if (alive) enter_cell();
        }
      }
    }
    const unsigned long long waiting = __builtin_amdgcn_ballot_w64(kept != 0u);
    const unsigned long long working = __builtin_amdgcn_ballot_w64(alive && kept == 0u);
#if defined(RTK_WALK_STATS)  // experiment builds: exact_tests counts something else
    if (RTK_WALK_STATS >= 16 && ++stat_rounds == RTK_WALK_STATS) wc.exact += 1u;         // walks of at least that many rounds
    if (RTK_WALK_STATS == 1) wc.exact += 1u;                                            // rounds of the walk's loop
    if (RTK_WALK_STATS == 2) wc.exact += (unsigned)__builtin_popcountll(working);       // lanes that worked in them
    if (RTK_WALK_STATS == 3 && waiting != 0ull && (__builtin_popcountll(waiting) >= 24 || working == 0ull)) wc.exact += 1u;  // rounds with an f64 phase
#endif
    if (waiting != 0ull && (__builtin_popcountll(waiting) >= 24 || working == 0ull)) {
#if !defined(RTK_WALK_STATS)
      wc.exact += (unsigned)__builtin_popcountll(waiting);
#endif
      if (kept != 0u) {
        const uint32_t i = (uint32_t)__builtin_ctz(kept);
        kept &= kept - 1u;
        const uint32_t at = i == 0u ? slot[0] : (i == 1u ? slot[1] : (i == 2u ? slot[2] : slot[3]));
        seen1 = seen0, seen0 = at;
        const double stop = exact(at);
        best_u = in_cells(stop);
        if (stop < 0.0) alive = false, kept = 0u;  // the caller has its answer
      }
    } else if (working == 0ull) {
      break;  // nobody waits, nobody works
    }
  }
  return at_origin;
}

// May this lane's ray take the walk?  (Comparisons are written so that a NaN says no.)
__device__ __forceinline__ bool grid_walk_applies(const DevScene& S, V3 o, V3 d) {
  const double dd = dot3(d, d);
  // 200,000 cells: the roots of a ray that grazes a sphere of ten cells' radius from that far are off by
  // 1.5e-8 x distance x radius / cell = 0.03 cells at most, inside what the walk allows for when it stops
  const double reach = 2.0e5;
  bool ok = dd > 1.0e-280 && dd < 1.0e280;
  ok = ok && fabs((o.x - S.tg_lo[0]) * S.tg_inv_h) <= reach && fabs((o.y - S.tg_lo[1]) * S.tg_inv_h) <= reach &&
       fabs((o.z - S.tg_lo[2]) * S.tg_inv_h) <= reach;
  return ok;
}

// Closest hit of a ray with an origin and a direction of its own: the walk where the scene has a cell
// grid and the ray may take it, the cluster sweep otherwise.
template <bool WIDE, bool TRACK>
__device__ __forceinline__ void closest_secondary(const DevScene& S, bool active, V3 o, V3 d, Closest& best, WaveCount& wc,
                                                  bool& graze) {
  if (!S.trace_grid) {
    closest_general<WIDE, TRACK>(S, active, o, d, best, wc, graze);
    return;
  }
  constexpr bool track = TRACK;
  closest_init(best);
#if defined(RTK_HIT_CUT) && RTK_HIT_CUT == 3  // ... every ray walks, whatever its origin
  const bool walks = active;
#else
  const bool walks = active && grid_walk_applies(S, o, d);
#endif
  if (walks) {
    for (uint32_t ip = 0; ip < S.n_planes; ++ip) {
      const uint32_t obj = S.plane_object[ip];
      double t0, t1;
      if (intersect_object(S.objects[obj], o, d, 0.0, t0, t1)) closest_consider(best, t0, obj, track);
    }
  }
  // forward for the winner; then, where the winner's material asks for n1, backward for the entry before it
  bool origin_cell_done = false;  // the forward walk began in the origin's cell and so tested what is listed there
#pragma unroll 1
#if defined(RTK_HIT_CUT) && RTK_HIT_CUT == 4  // experiment builds (wrong images): no walk against the direction
  for (int pass = 0; pass < 1; ++pass) {
#else
  for (int pass = 0; pass < (TRACK ? 2 : 1); ++pass) {
#endif
    bool go = walks;
    if (pass == 1) {
      go = walks && best.obj != 0xffffffffu;
      if (go) {
        const RTK_RO DevObject& ob = S.objects[best.obj];
        go = !is_approx_zero(ob.reflectivity) || !is_approx_zero(ob.transparency);
      }
    }
    if (__any(go)) {
      const bool backward = pass == 1;
      auto exact = [&](uint32_t at) -> double {
        const RTK_RO SphRecord& rec = S.sph_record[at];
        double t0, t1;
        if (intersect_sphere_record(S, rec, o, d, 0.0, t0, t1)) {
          closest_consider(best, t0, rec.object, track);
          closest_consider(best, t1, rec.object, track);
          if (track && fabs(t0 - t1) < 0.00001) graze = true;  // `Hit::eq` would call the two roots one hit
        }
        return backward ? -best.prev_t : best.t;
      };
      const bool began_there = grid_march(S, go, o, d, backward, backward ? -best.prev_t : best.t, wc, exact, backward && origin_cell_done);
      if (!backward) origin_cell_done = go && began_there;
    }
  }
  const bool swept = active && !walks;
  if (__any(swept)) {
    Closest other;
    bool other_graze = false;
    closest_general<WIDE, TRACK>(S, swept, o, d, other, wc, other_graze);
    if (swept) best = other, graze = other_graze;
  }
}

// Closest hit of a camera ray (unit direction, shared origin).
template <bool TRACK>
__device__ __forceinline__ void closest_primary(const DevScene& S, const float4* __restrict__ prim,
                                                const float4* __restrict__ prim_cluster, bool active, V3 o, V3 d,
                                                Closest& best, WaveCount& wc, bool& graze) {
  closest_init(best);
  constexpr bool track = TRACK;
  if (active) {
    for (uint32_t ip = 0; ip < S.n_planes; ++ip) {
      uint32_t obj = S.plane_object[ip];
      double t0, t1;
      if (intersect_object(S.objects[obj], o, d, 0.0, t0, t1)) closest_consider(best, t0, obj, track);
    }
  }
  auto exact = [&](uint32_t obj) {
    double t0, t1;
    if (intersect_object(S.objects[obj], o, d, 0.0, t0, t1)) {
      closest_consider(best, t0, obj, track);
      closest_consider(best, t1, obj, track);
      if (track && fabs(t0 - t1) < 0.00001) graze = true;
    }
  };
  sweep_primary(S, prim, prim_cluster, active, (float)d.x, (float)d.y, (float)d.z, wc, exact);
}

// ------------------------------------------------------------------------------------
// `Scene::is_shadowed` for shading points that have nothing to do with each other, towards lights built
// from points (w = 1): the shadow ray is a plain segment from the point to the light, and every lane
// walks ITS segment through the cell grid, light after light until one is blocked.  Points from which
// the walk cannot start (grid_walk_applies) take the cluster sweeps.
template <int LB, bool WIDE>
__device__ __forceinline__ bool shadow_by_walks(const DevScene& S, bool active, V3 p, WaveCount& wc) {
  // a segment that ends at the light: its direction has the length of the light's distance, never zero
  // for a point that is not the light itself (such a point sweeps)
  bool walks = active;
  for (uint32_t k = 0; k < S.n_lights; ++k) walks = walks && grid_walk_applies(S, p, ld3(S.lights[k].pos) - p);
  bool in_shadow = false;
  const bool swept = active && !walks;
  if (__any(swept)) {  // (one light at a time: the same rays, and fewer registers on a way few points take)
    const bool s = shadow_by_sweeps<1, WIDE>(S, swept, p, wc);
    if (swept) in_shadow = s;
  }
  bool need = walks;
  for (uint32_t k = 0; k < S.n_lights; ++k) {
    if (!__any(need)) break;
    const RTK_RO DevLight& L = S.lights[k];
    // the reference's shadow ray towards this light (src/scene.rs:208-225); w = 1: direction.w = 0
    const V3 lv = ld3(L.pos) - p;
    const double lw = L.pos[3] - 1.0;
    const double m = sqrt_f64(lv.x * lv.x + lv.y * lv.y + lv.z * lv.z + lw * lw);
    double dw;
    const V3 d = divide_by_length(lv, lw, m, dw);
    wc.shadow += lanes_where(need);
    bool blocked = false;
    if (need) {
      for (uint32_t ip = 0; ip < S.n_planes; ++ip) {  // planes: exact
        const RTK_RO DevObject& ob = S.objects[S.plane_object[ip]];
        if (!plane_may_block(plane_side(ob, p), S.plane_light[ip * S.n_lights + k])) continue;
        double t0, t1;
        const int n = intersect_object(ob, p, d, dw, t0, t1);
        if (n >= 1 && t0 > 0.0 && t0 < m) blocked = true;
      }
    }
    const bool go = need && !blocked;
    if (__any(go)) {
      auto exact = [&](uint32_t at) -> double {
        const RTK_RO SphRecord& rec = S.sph_record[at];
        double t0, t1;
        const int n = intersect_sphere_record(S, rec, p, d, dw, t0, t1);
        if (n == 2 && ((t0 > 0.0 && t0 < m) || (t1 > 0.0 && t1 < m))) {
          blocked = true;
          return -1.0;
        }
        return m;
      };
      grid_march(S, go, p, d, false, m, wc, exact);  // |d| = 1 up to rounding: the walk stops behind the light
    }
    if (need && blocked) in_shadow = true, need = false;
  }
  return in_shadow;
}

// Which of the three a scene's shading points take.  A kernel built for one of them (HOW >= 0, chosen by
// the host with shadow_mode) carries that one and the sweeps it falls back to, not all three.
constexpr int kShadowSweeps = 0, kShadowGrids = 1, kShadowWalks = 2, kShadowAny = -1;
template <int LB, bool WIDE, int HOW = kShadowAny>
__device__ __forceinline__ bool shadow_sweeps(const DevScene& S, bool active, V3 p, WaveCount& wc) {
  if (WIDE && (HOW == kShadowGrids || (HOW == kShadowAny && S.shadow_grid))) return shadow_by_grids<LB, WIDE>(S, active, p, wc);
  if (WIDE && (HOW == kShadowWalks || (HOW == kShadowAny && !S.shadow_grid && S.trace_grid && S.lights_point)))
    return shadow_by_walks<LB, WIDE>(S, active, p, wc);
  return shadow_by_sweeps<LB, WIDE>(S, active, p, wc);
}

}  // namespace rtk
#endif
