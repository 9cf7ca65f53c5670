// rtk_wavefront.hip -- scenes with reflective or transparent materials, traced level by level.
//
// The ray tree of a pixel (`Scene::trace`, src/scene.rs:136-296) has up to 2^max_depth - 1 nodes,
// and one lane walking it depth first needs one wavefront-wide sweep per node: a pixel of 60 nodes
// keeps its wavefront alive for 60 sweeps, long after the pixels around it are done, and the frame
// is not finished before its slowest pixel is.  Here the nodes of ALL pixels are traced one tree
// level at a time instead: the camera rays first; the children they produce are appended to a queue
// in device memory and traced by the next launch, one lane per node, every wavefront full; and so
// on down to the depth limit.  The longest dependency chain of a frame is max_depth launches.
//
// What a node adds to its pixel is not summed on the spot -- f64 sums depend on their order, and the
// order must be that of the depth-first evaluation whatever order the nodes were traced in.  Every
// node stores its terms and the queue positions of its children (NodeOut); when all levels are done
// one lane per pixel walks the pixel's tree depth first over these records, reflected subtree before
// refracted subtree, and adds the terms exactly as the one-lane evaluation (StackSink in rtk_trace.h)
// would have, then stores the pixel (src/graphics.rs:59-74).
//
// The queue has a fixed capacity.  A frame is cut into passes of as many tiles as are sure to have
// their camera rays and a generous number of children fit; should the children of a pass not fit
// after all, a flag is raised, and the caller re-renders the launch with the depth-first kernel
// (which needs no queue), guarded by that flag.
#include "rtk_queue.h"
#include "rtk_kernels.h"

#include <algorithm>

#pragma clang fp contract(off)

namespace rtk {

#ifndef RTK_WF_OCC
#define RTK_WF_OCC 4
#endif
#ifndef RTK_WF_GRID_MAX
#define RTK_WF_GRID_MAX 65536u
#endif
#ifndef RTK_WF_LEVEL_BLOCK
#define RTK_WF_LEVEL_BLOCK 128  // threads of a workgroup of the level kernel, which takes one run of queue positions per batch of its nodes
#endif
// Start of a pass after the first: no level has nodes yet.
__global__ void wf_mark_kernel(unsigned int* ctrl) {
  if (blockIdx.x == 0 && threadIdx.x <= (unsigned)kMaxDepth) ctrl[kCtrlLevelCount + threadIdx.x] = 0u;
}

// Level 0: the camera rays of the pass's tiles (ray generation: src/scene/cameras.rs:52-69 with the
// matrix inverse hoisted to the host).  One workgroup per 32x8 tile, one lane per pixel; node index
// = workgroup * 256 + thread.
template <int LB, bool WIDE, bool TRACK>
__global__ void __launch_bounds__(kBlock, 4) wf_primary_kernel(DevScene S, DevCamera C, RenderTarget T, const float4* __restrict__ prim,
                                                            const float4* __restrict__ prim_cluster, uint32_t tile_base,
                                                            WavefrontBuffers B) {
  const uint32_t k = tile_base + blockIdx.x;
  const TileRef tr = tile_ref(T, k);
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  const uint32_t lx = (wave << 3) | (lane & 7u), ly = lane >> 3;
  const uint32_t x = tr.tx * 32u + lx, y = tr.ty * 8u + ly;
  const bool valid = x < C.width && y < C.height;
  const V3 origin = ld3(C.origin);
  const V3 dir = camera_ray_direction(C, x, y);
  WaveCount wc;
  __shared__ uint32_t s_counts[10];
  QueueSink<4> sink(S, B, gridDim.x * kBlock, gridDim.x * kBlock, 1u, s_counts);
  trace_step<true, true, LB, WIDE, false, TRACK>(S, prim, prim_cluster, valid, true, origin, dir, 1.0, 0, sink, wc);
  sink.store(blockIdx.x * kBlock + threadIdx.x);
  flush_counts(counter_shard(B.counters), wc);
}

// Level l >= 1: the nodes queued by level l - 1 (all of depth l), one lane per node: closest hit, shading,
// children to the queue, the node's result record.  One launch per level: the ray is read once and the hit
// stays in registers.  (Two launches, one finding the hits and one shading them with the hit handed over
// through the queue, gave the first twice the wavefronts in flight at half the registers; the 116 bytes per
// node that cost weigh more: one launch is 1-7 % faster on every scene measured but one, 0.6 % slower.)
// Workgroups of two wavefronts take 128 nodes at a time, dealt round-robin, and queue the children of the 128
// together: a workgroup's resources fall free when its slowest wavefront is done, and its run of queue positions
// costs one atomic on one address -- with four wavefronts (fewer atomics, longer waits for the slowest) a 1080p
// glass frame takes 1.25 ms, with two 1.22, with one (as many atomics as wavefronts) 1.27.
// (Handing nodes out through a counter in memory costs more than the uneven shares it avoids: one address
// takes an atomic every 12 ns or so, and a level has tens of thousands of wavefronts' worth of nodes.)
template <int LB, bool WIDE, bool TRACK, int SHADOWS>
__global__ void __launch_bounds__(RTK_WF_LEVEL_BLOCK, RTK_WF_OCC) wf_level_kernel(DevScene S, uint32_t level, uint32_t n_roots, WavefrontBuffers B) {
  const LevelBounds lv = level_bounds(B, n_roots, level);
  const uint32_t end = min(lv.end, B.cap_nodes);
  WaveCount wc;
  __shared__ uint32_t s_counts[10];
  for (uint32_t base = lv.begin + blockIdx.x * RTK_WF_LEVEL_BLOCK; base < end; base += gridDim.x * RTK_WF_LEVEL_BLOCK) {
    const uint32_t i = base + threadIdx.x;
    const bool have = i < end;
    V3 ro = {0, 0, 0}, rd = {0, 0, 1};
    double weight = 0.0;
    if (have) {
      const double* r = ray_fields(B, i - n_roots);
      ro = {r[(kRayOrigin + 0) * 64], r[(kRayOrigin + 1) * 64], r[(kRayOrigin + 2) * 64]};
      rd = {r[(kRayDirection + 0) * 64], r[(kRayDirection + 1) * 64], r[(kRayDirection + 2) * 64]};
      weight = r[kRayWeight * 64];
    }
    const NodeHit h = find_hit<false, WIDE, false, TRACK>(S, nullptr, nullptr, have, false, ro, rd, (int)level, wc);
    QueueSink<RTK_WF_LEVEL_BLOCK / 64> sink(S, B, n_roots, lv.end, level + 1u, s_counts);
    shade_hit<true, LB, WIDE, false, SHADOWS>(S, have, h, ro, rd, weight, (int)level, sink, wc);  // (barriers inside: every wavefront goes in)
    if (have) sink.store(i);
  }
  flush_counts(counter_shard(B.counters), wc);
}

// The pixels of the pass's tiles from the node records: depth-first walk, terms added in the order
// of the one-lane evaluation (a node's untraced refraction and reflection, its own term, then the
// reflected subtree, then the refracted one), and the framebuffer store (src/graphics.rs:44-46, 59-74).
__global__ void __launch_bounds__(kBlock) wf_gather_kernel(DevScene S, DevCamera C, RenderTarget T, uint32_t tile_base,
                                                           const WfNodeOut* __restrict__ outs, const WfNodeCut* __restrict__ cuts) {
  __shared__ float s_rgb[768];
  __shared__ uint32_t s_px[256];
  const V3 ambient = ld3(S.ambient);
  V3 color = {0, 0, 0};
  uint32_t pending[2 * kMaxDepth + 2];
  int sp = 0;
  pending[sp++] = blockIdx.x * kBlock + threadIdx.x;
  while (sp > 0) {
    const uint32_t at = pending[--sp];
    const WfNodeOut& n = outs[at];
    const uint32_t refraction = n.child_refraction, reflection = n.child_reflection;
    if (refraction == kCutOff) color = color + ambient * cuts[at].w_refraction;
    if (reflection == kCutOff) color = color + ambient * cuts[at].w_reflection;
    color = color + V3{n.t[0], n.t[1], n.t[2]};
    if (refraction < kCutOff) pending[sp++] = refraction;
    if (reflection < kCutOff) pending[sp++] = reflection;  // on top: walked first
  }
  store_tile(C, T, tile_base + blockIdx.x, color, s_rgb, s_px);
}

// End of a launch: the rays counted by its passes join the frame's counters, unless the launch is
// about to be rendered again by the depth-first kernel (which counts them itself).
__global__ void __launch_bounds__(kCounterShards) wf_commit_kernel(const unsigned int* __restrict__ ctrl,
                                                                   const DevCounters* __restrict__ mine, DevCounters* frame,
                                                                   unsigned int* host_overflow) {
  if (ctrl[kCtrlOverflow]) {
    if (threadIdx.x == 0 && host_overflow) *host_overflow = 1u;  // the host reserves more room per pixel from now on
    return;
  }
  const DevCounters c = mine[threadIdx.x];
  DevCounters& f = frame[threadIdx.x];
  f.rays_primary += c.rays_primary;
  f.rays_shadow += c.rays_shadow;
  f.rays_secondary += c.rays_secondary;
  f.exact_tests += c.exact_tests;
}

// Scenes that are swept through clusters at all (a handful of objects are intersected one by one by a
// kernel of their own) and whose rays recurse.
bool wavefront_applies(const DevScene& S) { return S.any_secondary && S.n_objects > kBruteObjects && S.max_depth >= 2; }

// ... and launches small enough for the depth-first kernel's last pixels to matter.  That kernel
// spends a fixed time on its slowest pixels (a wavefront stays for as long as its deepest ray tree
// takes) and then less memory traffic per pixel than the queue, which moves about 250 bytes per node.
// Measured on glass scenes, by level / depth first in ms: 10,000 spheres 1080p 1.28 / 2.17, 4K 3.78 / 5.40,
// 5K 6.66 / 8.54, 8K (in passes) 13.5 / 17.3; 1,000 spheres 1080p 1.06 / 1.78, 4K 3.25 / 4.23, 5K 5.75 / 6.86,
// 8K 11.8 / 14.0; 300 spheres 4K 2.46 / 3.14, 8K 9.24 / 10.8; 64 spheres 1080p 0.81 / 1.15, 4K 2.41 / 2.55,
// 8K 9.19 / 8.57; 12 spheres 4K 1.53 / 1.80, 8K 5.79 / 6.45.
bool wavefront_pays(const DevScene& S, uint64_t launch_pixels) {
  return launch_pixels <= (S.n_spheres >= 256u ? 64000000ull : 16000000ull);
}

namespace {

template <bool WIDE, bool TRACK>
hipError_t launch_primary_lb(int lb, dim3 grid, hipStream_t st, const DevScene& S, const DevCamera& C, const RenderTarget& T,
                             const float4* prim, const float4* prim_cluster, uint32_t tile_base, const WavefrontBuffers& B) {
  switch (lb) {
    case 1: hipLaunchKernelGGL((wf_primary_kernel<1, WIDE, TRACK>), grid, dim3(kBlock), 0, st, S, C, T, prim, prim_cluster, tile_base, B); break;
    case 2: hipLaunchKernelGGL((wf_primary_kernel<2, WIDE, TRACK>), grid, dim3(kBlock), 0, st, S, C, T, prim, prim_cluster, tile_base, B); break;
    case 4: hipLaunchKernelGGL((wf_primary_kernel<4, WIDE, TRACK>), grid, dim3(kBlock), 0, st, S, C, T, prim, prim_cluster, tile_base, B); break;
    default: hipLaunchKernelGGL((wf_primary_kernel<8, WIDE, TRACK>), grid, dim3(kBlock), 0, st, S, C, T, prim, prim_cluster, tile_base, B); break;
  }
  return hipGetLastError();
}

template <bool TRACK, int SHADOWS>
hipError_t launch_level_lb(int lb, dim3 grid, hipStream_t st, const DevScene& S, uint32_t level, uint32_t n_roots,
                           const WavefrontBuffers& B) {
  switch (lb) {
    case 1: hipLaunchKernelGGL((wf_level_kernel<1, true, TRACK, SHADOWS>), grid, dim3(RTK_WF_LEVEL_BLOCK), 0, st, S, level, n_roots, B); break;
    case 2: hipLaunchKernelGGL((wf_level_kernel<2, true, TRACK, SHADOWS>), grid, dim3(RTK_WF_LEVEL_BLOCK), 0, st, S, level, n_roots, B); break;
    case 4: hipLaunchKernelGGL((wf_level_kernel<4, true, TRACK, SHADOWS>), grid, dim3(RTK_WF_LEVEL_BLOCK), 0, st, S, level, n_roots, B); break;
    default: hipLaunchKernelGGL((wf_level_kernel<8, true, TRACK, SHADOWS>), grid, dim3(RTK_WF_LEVEL_BLOCK), 0, st, S, level, n_roots, B); break;
  }
  return hipGetLastError();
}

// How the shading points of a scene's secondary hits look for their blockers (shadow_sweeps).
int shadow_mode(const DevScene& S) {
  if (S.shadow_grid) return kShadowGrids;
  if (S.trace_grid && S.lights_point) return kShadowWalks;
  return kShadowSweeps;
}

}  // namespace

hipError_t launch_render_wavefront(hipStream_t st, const DevScene& S, const DevCamera& C, const float4* prim,
                                   const float4* prim_ball, const float4* prim_cluster, const RenderTarget& T,
                                   const WavefrontBuffers& B, DevCounters* counters) {
  drop_stale_error();
  if (T.n_local_tiles == 0) return hipSuccess;
  const bool track = S.track_n1 != 0;
  const int lb = light_batch_for(S.n_lights);
  hipError_t e = hipMemsetAsync(B.ctrl, 0, sizeof(unsigned int) * kWavefrontCtrlWords, st);
  if (e == hipSuccess) e = hipMemsetAsync(B.counters, 0, sizeof(DevCounters) * kCounterShards, st);
  if (e != hipSuccess) return e;
  // tiles per pass: their camera rays and nodes_per_pixel - 1 children per pixel must fit
  const uint32_t pass_tiles = std::max<uint32_t>(1u, B.cap_nodes / (kBlock * std::max(1u, B.nodes_per_pixel)));
  for (uint32_t tile_base = 0; tile_base < T.n_local_tiles; tile_base += pass_tiles) {
    const uint32_t n_tiles = std::min(pass_tiles, T.n_local_tiles - tile_base);
    const uint32_t n_roots = n_tiles * kBlock;
    if (tile_base != 0u) {  // (the first pass finds the control block cleared)
      hipLaunchKernelGGL(wf_mark_kernel, dim3(1), dim3(64), 0, st, B.ctrl);
      if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    const dim3 grid(n_tiles);
    // the camera level: on the opaque path's bundle tests where they apply, else through trace_step
    if (queue_primary_bundled_applies(S, C))
      e = launch_queue_primary_bundled(st, S, C, prim, prim_ball, prim_cluster, T, tile_base, n_tiles, B);
    else
      e = track ? launch_primary_lb<true, true>(lb, grid, st, S, C, T, prim, prim_cluster, tile_base, B)
                : launch_primary_lb<true, false>(lb, grid, st, S, C, T, prim, prim_cluster, tile_base, B);
    if (e != hipSuccess) return e;
    // levels 1 .. max_depth - 1 (nodes at depth max_depth are not traced: their parents account for them);
    // enough resident workgroups to fill the GPU, no more than the pass could ever need
    // As many threads as the pass has pixels: a level usually has fewer nodes than that, so most workgroups take
    // one batch of nodes (or none) and the hardware's dispatcher evens out what the nodes cost; a level with
    // more nodes gives every workgroup several batches.
    const dim3 level_grid(std::min<uint32_t>(RTK_WF_GRID_MAX, n_tiles * (kBlock / RTK_WF_LEVEL_BLOCK)));
    const int mode = shadow_mode(S);
    for (uint32_t level = 1; level < S.max_depth; ++level) {
      if (track)
        e = mode == kShadowGrids   ? launch_level_lb<true, kShadowGrids>(lb, level_grid, st, S, level, n_roots, B)
            : mode == kShadowWalks ? launch_level_lb<true, kShadowWalks>(lb, level_grid, st, S, level, n_roots, B)
                                   : launch_level_lb<true, kShadowSweeps>(lb, level_grid, st, S, level, n_roots, B);
      else
        e = mode == kShadowGrids   ? launch_level_lb<false, kShadowGrids>(lb, level_grid, st, S, level, n_roots, B)
            : mode == kShadowWalks ? launch_level_lb<false, kShadowWalks>(lb, level_grid, st, S, level, n_roots, B)
                                   : launch_level_lb<false, kShadowSweeps>(lb, level_grid, st, S, level, n_roots, B);
      if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(wf_gather_kernel, grid, dim3(kBlock), 0, st, S, C, T, tile_base, B.outs, B.cuts);
    if ((e = hipGetLastError()) != hipSuccess) return e;
  }
  hipLaunchKernelGGL(wf_commit_kernel, dim3(1), dim3(kCounterShards), 0, st, B.ctrl, B.counters, counters, B.host_overflow);
  return hipGetLastError();
}

const unsigned int* wavefront_overflow_flag(const WavefrontBuffers& B) { return B.ctrl + kCtrlOverflow; }

}  // namespace rtk
