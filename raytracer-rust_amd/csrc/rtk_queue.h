// rtk_queue.h -- the node queue of the level-by-level order (rtk_wavefront.hip): control words,
// record flags, and the sink through which a kernel that traces one node per lane hands the node's
// terms and children on.  Device code only (included by .hip files after rtk_trace.h).
#ifndef RTK_QUEUE_H
#define RTK_QUEUE_H

#include "rtk_trace.h"

namespace rtk {

constexpr uint32_t kNoNode = 0xffffffffu;  // a child index: no such child
constexpr uint32_t kCutOff = 0xfffffffeu;  // ... a child cut off by the depth limit: it adds the scene's ambient colour times its weight

// words of the control block
constexpr int kCtrlOverflow = 1;    // some child found no room: the launch is to be rendered again
constexpr int kCtrlLevelCount = 3;  // [kCtrlLevelCount + l]: nodes queued for level l of this pass (1 <= l <= kMaxDepth)
static_assert(kCtrlLevelCount + kMaxDepth < (int)kWavefrontCtrlWords, "one count per depth fits the control block");

// Where the nodes of level l lie in the pass's numbering: the camera rays are nodes 0 .. n_roots - 1, the levels
// follow one another.  A level's count is final when the launch that shades the level before it has ended, so
// the launch that traces level l reads its own bounds, and nothing has to mark them in between.
struct LevelBounds {
  uint32_t begin, end;  // not clamped to the queue's capacity: children placed at `end` onwards notice themselves when they do not fit
};
__device__ __forceinline__ LevelBounds level_bounds(const WavefrontBuffers& B, uint32_t n_roots, uint32_t level) {
  uint32_t begin = n_roots;
  for (uint32_t l = 1; l < level; ++l) begin += B.ctrl[kCtrlLevelCount + l];
  return {begin, begin + B.ctrl[kCtrlLevelCount + level]};
}

// Queue entry j: its element of the block's first field (field f of the entry is [f * 64]).
__device__ __forceinline__ double* ray_fields(const WavefrontBuffers& B, uint32_t j) {
  return reinterpret_cast<double*>(B.rays + (size_t)(j >> 6) * kRayBlockBytes) + (j & 63u);
}

// Result sink for one node of the level-by-level order: keeps the node's terms and queues its
// children.  The workgroups of these kernels move in step (one node per lane, then all together): when
// every lane has formed the children of its hit (children_done, which every thread of the workgroup
// calls once per node, at the same point), the workgroup takes ONE run of queue positions for all of
// them -- the refracted children in thread order, then the reflected ones -- and every lane writes its
// children there straight from its registers, which are free again before the shadow rays start.
// Children of neighbouring parents so stay neighbours in the queue whatever other workgroups push
// meanwhile, and the next level's wavefronts (64 consecutive entries each) start close together.
template <int WAVES>  // wavefronts of the workgroup (at most four)
struct QueueSink {
  V3 t = {0, 0, 0};
  double w_refraction = 0.0, w_reflection = 0.0;
  uint32_t child_refraction = kNoNode, child_reflection = kNoNode;
  // the children, from sink.child() until children_done()
  V3 o_refraction = {0, 0, 0}, d_refraction = {0, 0, 0}, o_reflection = {0, 0, 0}, d_reflection = {0, 0, 0};
  double weight_refraction = 0.0, weight_reflection = 0.0;
  bool has_refraction = false, has_reflection = false;
  const WavefrontBuffers& B;
  uint32_t n_roots, child_base, child_depth;  // the children go to level child_depth, whose nodes start at child_base
  uint32_t* s_counts;  // 10 words of LDS
  bool ambient_black;  // the scene's ambient colour is zero: a cut-off child with a finite weight adds +-0, which changes no sum

  __device__ __forceinline__ QueueSink(const DevScene& S, const WavefrontBuffers& buffers, uint32_t roots, uint32_t first_child,
                                       uint32_t depth_of_children, uint32_t* lds_counts)
      : B(buffers), n_roots(roots), child_base(first_child), child_depth(depth_of_children), s_counts(lds_counts),
        ambient_black(S.ambient[0] == 0.0 && S.ambient[1] == 0.0 && S.ambient[2] == 0.0) {}
  __device__ __forceinline__ void term(V3 v, double w) { t = v * w; }
  __device__ __forceinline__ void untraced_child(bool reflection, V3, double w) {
    if (ambient_black && __builtin_isfinite(w)) return;  // (0 x inf and 0 x NaN are NaN: those are kept)
    if (reflection)
      w_reflection = w, child_reflection = kCutOff;
    else
      w_refraction = w, child_refraction = kCutOff;
  }
  __device__ __forceinline__ void child(bool reflection, V3 o, V3 d, double w, int) {
    if (reflection)
      o_reflection = o, d_reflection = d, weight_reflection = w, has_reflection = true;
    else
      o_refraction = o, d_refraction = d, weight_refraction = w, has_refraction = true;
  }
  __device__ __forceinline__ void children_done() {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const unsigned long long m0 = __builtin_amdgcn_ballot_w64(has_refraction), m1 = __builtin_amdgcn_ballot_w64(has_reflection);
    if (lane == 0) {
      s_counts[wave] = (uint32_t)__builtin_popcountll(m0);
      s_counts[4u + wave] = (uint32_t)__builtin_popcountll(m1);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      uint32_t total = 0;
#pragma unroll
      for (uint32_t w = 0; w < (uint32_t)WAVES; ++w) total += s_counts[w] + s_counts[4u + w];
      s_counts[8] = total ? atomicAdd(&B.ctrl[kCtrlLevelCount + child_depth], total) : 0u;
    }
    __syncthreads();
    uint32_t before0 = 0, before1 = 0;  // children of the workgroup queued ahead of this wavefront's, per kind
#pragma unroll
    for (uint32_t w = 0; w < (uint32_t)WAVES; ++w) {
      before1 += s_counts[w];
      if (w < wave) before0 += s_counts[w], before1 += s_counts[4u + w];
    }
    const uint32_t base = child_base + s_counts[8];
    auto place = [&](bool has, unsigned long long m, uint32_t before, V3 o, V3 d, double weight, uint32_t& child) {
      if (!has) return;
      const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
      const uint32_t at = base + before + rank;
      if (at < B.cap_nodes) {
        double* r = ray_fields(B, at - n_roots);
        r[(kRayOrigin + 0) * 64] = o.x, r[(kRayOrigin + 1) * 64] = o.y, r[(kRayOrigin + 2) * 64] = o.z;
        r[(kRayDirection + 0) * 64] = d.x, r[(kRayDirection + 1) * 64] = d.y, r[(kRayDirection + 2) * 64] = d.z;
        r[kRayWeight * 64] = weight;
        child = at;
      } else {
        B.ctrl[kCtrlOverflow] = 1u;
      }
    };
    place(has_refraction, m0, before0, o_refraction, d_refraction, weight_refraction, child_refraction);
    place(has_reflection, m1, before1, o_reflection, d_reflection, weight_reflection, child_reflection);
    __syncthreads();  // the counts are free again
  }
  __device__ __forceinline__ void store(uint32_t node) const {
    WfNodeOut& out = B.outs[node];
    out.t[0] = t.x, out.t[1] = t.y, out.t[2] = t.z;
    out.child_refraction = child_refraction;
    out.child_reflection = child_reflection;
    if (child_refraction == kCutOff || child_reflection == kCutOff) {
      WfNodeCut& cut = B.cuts[node];
      cut.w_refraction = w_refraction;
      cut.w_reflection = w_reflection;
    }
  }
};

}  // namespace rtk
#endif
