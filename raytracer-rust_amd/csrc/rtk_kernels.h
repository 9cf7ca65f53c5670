// rtk_kernels.h -- launch wrappers implemented in rtk_kernels.hip.
#ifndef RTK_KERNELS_H
#define RTK_KERNELS_H

#include "rtk_device.h"

namespace rtk {

// A kernel launch reports failure only through hipGetLastError, which also returns whatever error an
// earlier, unrelated call on this thread left behind (a probe of a pageable pointer, say).  Every
// launch wrapper starts by dropping that, so that what it reads after its launch is its own.
inline void drop_stale_error() { (void)hipGetLastError(); }

// Lights fused per shadow sweep for a scene with n_lights lights (1, 2, 4 or 8).
int light_batch_for(uint32_t n_lights);

// Start of a frame: rebuilds the camera-ray records for camera C -- prim[n_spheres_padded] (per-ray
// filter), prim_ball[n_spheres_padded] and prim_cluster[n_clusters_padded] (wavefront-level cone
// tests) -- and clears the frame's counters (kCounterSlots elements).
hipError_t launch_prep_primary(hipStream_t st, const DevScene& S, const DevCamera& C, float4* prim, float4* prim_ball,
                               float4* prim_cluster, DevCounters* counters);

// Renders the tiles selected by T.
// `counters` holds kCounterShards shards followed by one more element whose first word is the
// launch's work counter; all of it zeroed before the launch.  wf (may be null): queue buffers; a scene
// with secondary rays and a depth limit of 2 or more is then traced level by level (the caller decides
// whether that pays, see wavefront_applies / wavefront_pays); prim / prim_cluster are the per-camera records.
hipError_t launch_render(hipStream_t st, const DevScene& S, const DevCamera& C, const RenderTarget& T, DevCounters* counters,
                         const WavefrontBuffers* wf = nullptr, const float4* prim = nullptr, const float4* prim_ball = nullptr,
                         const float4* prim_cluster = nullptr);

// Level-by-level tracing (rtk_wavefront.hip).  wavefront_applies: the scene has secondary rays to
// queue and enough objects for the cluster sweeps.  launch_render_wavefront renders the tiles of T
// in passes that fit B and adds its ray counts to `counters`, unless some pass ran out of queue
// room: then the word at wavefront_overflow_flag(B) is non-zero after the launch, nothing has been
// added to `counters`, and the caller renders T again with a kernel that needs no queue.
bool wavefront_applies(const DevScene& S);
// ... and a launch of that many pixels is expected to finish sooner that way than depth first.
bool wavefront_pays(const DevScene& S, uint64_t launch_pixels);
hipError_t launch_render_wavefront(hipStream_t st, const DevScene& S, const DevCamera& C, const float4* prim,
                                   const float4* prim_ball, const float4* prim_cluster, const RenderTarget& T,
                                   const WavefrontBuffers& B, DevCounters* counters);
const unsigned int* wavefront_overflow_flag(const WavefrontBuffers& B);
// Level 0 of that order on the bundle tests of the opaque path (rtk_opaque.hip), for the scenes it
// applies to: the camera rays of tiles [tile_base, tile_base + n_tiles) of T, records and children into B.
bool queue_primary_bundled_applies(const DevScene& S, const DevCamera& C);
hipError_t launch_queue_primary_bundled(hipStream_t st, const DevScene& S, const DevCamera& C, const float4* prim,
                                        const float4* prim_ball, const float4* prim_cluster, const RenderTarget& T,
                                        uint32_t tile_base, uint32_t n_tiles, const WavefrontBuffers& B);

// Same for scenes without reflective or transparent materials: ray kernel(s) (camera +
// shadow rays) and a shading kernel, with intermediate records in B.  `after_rays`, when not
// null, is recorded between the two.
hipError_t launch_render_opaque(hipStream_t st, const DevScene& S, const DevCamera& C, const float4* prim,
                                const float4* prim_ball, const float4* prim_cluster, const RenderTarget& T,
                                const OpaqueBuffers& B, DevCounters* counters, hipEvent_t after_rays);

// True when launch_render_opaque runs its three-kernel path for S and reads / writes the per-pixel records B.
bool opaque_needs_records(const DevScene& S);

// Traces n caller-supplied rays (device pointers, n x 4 doubles in, n x 3 doubles out).
hipError_t launch_trace_rays(hipStream_t st, const DevScene& S, const double* origins, const double* directions, uint64_t n,
                             double* colors, DevCounters* counters);

// Compact per-rank tile buffers -> row-major frame.  Rank r's FP32 tiles start
// r * rgb_stride_floats floats into rgb_c, its RGBA8 tiles r * px_stride_pixels pixels into px_c.
hipError_t launch_untile(hipStream_t st, uint32_t width, uint32_t height, uint32_t world, uint64_t rgb_stride_floats,
                         uint64_t px_stride_pixels, const float* rgb_c, float* rgb_f, const uint8_t* px_c, uint8_t* px_f);

// s x s supersampled FP32 frame (s*width x s*height samples, row-major) -> width x height
// frame: per-pixel mean, FP32 and RGBA8 (either output may be null).
hipError_t launch_resolve(hipStream_t st, uint32_t width, uint32_t height, uint32_t s, const float* samples, float* rgb,
                          uint8_t* rgba8);

// The same on a rank's compact buffer of n_tiles 32x8 sample tiles (s divides 8): tile k becomes
// 256 / s^2 pixels, a (32/s) x (8/s) tile in row-major order.
hipError_t launch_resolve_tiles(hipStream_t st, uint64_t n_tiles, uint32_t s, const float* samples, float* rgb, uint8_t* rgba8);

// launch_untile for buffers made of such (32/s) x (8/s) tiles.
hipError_t launch_untile_sampled(hipStream_t st, uint32_t width, uint32_t height, uint32_t s, uint32_t world,
                                 uint64_t rgb_stride_floats, uint64_t px_stride_pixels, const float* rgb_c, float* rgb_f,
                                 const uint8_t* px_c, uint8_t* px_f);

}  // namespace rtk
#endif
