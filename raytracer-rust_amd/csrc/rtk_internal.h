// rtk_internal.h -- what the device-side translation units of the host code share (rtk_api.hip,
// rtk_group.hip): the context, HIP error reporting, placing a flattened scene on a device, and the
// single-device render call the multi-device code is built on.  Host code only.
#ifndef RTK_INTERNAL_H
#define RTK_INTERNAL_H

#include "rtk_host.h"
#include "rtk_kernels.h"

namespace rtk {

int fail_hip(hipError_t e, const char* what);

}  // namespace rtk

#define RTK_HIP(call)                                      \
  do {                                                     \
    hipError_t e_ = (call);                                \
    if (e_ != hipSuccess) return rtk::fail_hip(e_, #call); \
  } while (0)

struct rtk_context {
  int device = -1;
  hipStream_t stream = nullptr;
  hipStream_t copy_stream = nullptr;  // device-to-host copies of rtk_render, overlapping its kernels
  // One (start, end of the ray kernels, stop) event triple per recent call, used round-robin, so
  // that a caller can render many frames back to back and read their kernel times afterwards.
  hipEvent_t ev_start[RTK_TIMING_HISTORY] = {}, ev_rays[RTK_TIMING_HISTORY] = {}, ev_stop[RTK_TIMING_HISTORY] = {};
  bool ev_split[RTK_TIMING_HISTORY] = {};  // ev_rays was recorded by the call that used the slot
  uint64_t n_timed = 0;                    // calls that recorded a triple; the newest sits in slot (n_timed - 1) % N
  rtk::DevCounters* d_counters = nullptr;
  rtk::DevCounters* last_counters = nullptr;  // where the last call counted its rays
  unsigned char* static_scratch = nullptr;    // this device's copy of the library's static scratch array
  hipStream_t last_stream = nullptr;
  // staging for rtk_render (host-buffer variant)
  float* d_rgb = nullptr;
  uint8_t* d_rgba8 = nullptr;
  size_t cap_pixels = 0;
  static constexpr int kHostBlocks = 16;  // blocks a frame is cut into by rtk_render
  hipEvent_t ev_block[kHostBlocks] = {};   // a block's kernels have finished
  hipEvent_t ev_landed[kHostBlocks] = {};  // ... and its rows have arrived in host memory
  unsigned char* h_bounce = nullptr;       // page-locked landing area for pageable destinations (grown on demand)
  size_t cap_bounce = 0;
  // intermediate records of the three-kernel path (grown on demand, kept for the next frame)
  rtk::OpaqueBuffers opaque{};
  size_t cap_opaque = 0;
  // node queue of the level-by-level path for scenes with secondary rays (grown on demand up to
  // wavefront_max_nodes; 0 there keeps such scenes on the depth-first kernel)
  rtk::WavefrontBuffers wavefront{};
  uint32_t wavefront_max_nodes = 0;
  uint32_t trace_order = RTK_TRACE_AUTO;
  // page-locked staging for scene uploads (grown on demand)
  unsigned char* h_stage = nullptr;
  size_t cap_stage = 0;
  // the scene of the last rtk_render call, kept on the device: a caller that renders the same scene
  // again (another camera, another frame size) skips flattening and upload
  std::vector<rtk_object> cached_objects;
  std::vector<rtk_light> cached_lights;
  rtk_scene cached_scene{};
  rtk_scene_handle* cached_handle = nullptr;
  bool cached_has_pointers = false;  // some material of it points to inner pattern transforms
  // sample frame of rtk_render_device_sampled (grown on demand)
  float* d_samples = nullptr;
  size_t cap_samples = 0;  // in samples
};

namespace rtk {

// Allocates on ctx's device, copies, and fills a new handle.  With `reuse` (a handle of ctx that no
// work in flight uses any more): its allocation takes the scene when it is large enough, and it is
// released otherwise; either way `reuse` must not be used again.
int place_scene(rtk_context* ctx, const FlatScene& flat, rtk_scene_handle** out, rtk_scene_handle* reuse = nullptr);


// rtk_render_device for an arbitrary partition of the frame (one sample per pixel).
int render_device_partition(rtk_context* ctx, rtk_scene_handle* scene, const rtk_camera* camera, const Partition& p,
                            float* d_rgb_f32, uint8_t* d_rgba8, void* hip_stream);

}  // namespace rtk

#endif
