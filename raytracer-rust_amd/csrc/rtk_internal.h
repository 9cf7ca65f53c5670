// rtk_internal.h -- what the host-side translation units of the library share: the opaque
// handles of include/rtk.h, error reporting, and the single-device render call the multi-device
// code is built on.  Host code only.
#ifndef RTK_INTERNAL_H
#define RTK_INTERNAL_H

#include "../../include/rtk.h"
#include "rtk_device.h"
#include "rtk_kernels.h"

#include <memory>
#include <string>
#include <vector>

namespace rtk {

// Sets the thread's error text (rtk_last_error) and returns `code`.
int fail(int code, const std::string& msg);
int fail_hip(hipError_t e, const char* what);

}  // namespace rtk

#define RTK_HIP(call)                                      \
  do {                                                     \
    hipError_t e_ = (call);                                \
    if (e_ != hipSuccess) return rtk::fail_hip(e_, #call); \
  } while (0)

// A scene resident in HBM: one device allocation, the pointers into it, and the argument block
// the kernels take.
struct rtk_scene_handle {
  rtk::DevScene dev{};
  void* d_block = nullptr;  // the one device allocation every pointer below points into
  size_t block_bytes = 0;   // its size
  void* d_objects = nullptr;
  void* d_lights = nullptr;
  void* d_filter = nullptr;
  void* d_exact = nullptr;
  void* d_sph_object = nullptr;
  void* d_sph_record = nullptr;
  void* d_plane_object = nullptr;
  void* d_plane_light = nullptr;
  void* d_light_kind = nullptr;
  void* d_pattern_inv = nullptr;
  void* d_prim = nullptr;          // camera-ray filter records (rebuilt per camera)
  void* d_prim_ball = nullptr;     // camera-ray records of the wavefront-level cone test (rebuilt per camera)
  void* d_sph_ball = nullptr;
  void* d_cluster_ball = nullptr;
  void* d_super_ball = nullptr;
  void* d_cluster_plain = nullptr;
  void* d_light_grid = nullptr;
  void* d_grid_cells = nullptr;
  void* d_grid_entries = nullptr;
  void* d_prim_cluster = nullptr;  // camera-ray records of the cluster balls (rebuilt per camera)
  void* d_light_filter = nullptr;
  void* d_k_table = nullptr;
  void* d_sph_tab = nullptr;
};

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
  // sample frame of rtk_render_device_sampled (grown on demand)
  float* d_samples = nullptr;
  size_t cap_samples = 0;  // in samples
};

namespace rtk {

// A scene flattened for the device, not yet on one: the bytes of the record arrays laid out as
// they will lie in the device allocation, and the kernels' argument block with its pointers unset.
struct FlatScene {
  struct Piece {
    void* rtk_scene_handle::*dst;
    size_t bytes, offset;
  };
  std::vector<Piece> pieces;
  unsigned char* staged = nullptr;  // the first `staged_bytes` bytes of the allocation
  size_t staged_bytes = 0;
  std::vector<unsigned char> extra;  // bytes that follow the staged ones in the allocation, built apart from them
  size_t total = 0;                 // allocation size (the per-camera records follow the staged part and the extra bytes)
  DevScene dev{};
  // Where the staged bytes live: memory from `grow` (e.g. a context's page-locked staging buffer,
  // which must hold at least the asked-for bytes until the scene has been placed), else `own`.
  unsigned char* (*grow)(void* user, size_t bytes) = nullptr;
  void* user = nullptr;
  std::unique_ptr<unsigned char[]> own;
};
// Validation and flattening (host work only); RTK_OK or an rtk_status with the error text set.
int flatten_scene(const rtk_scene* scene, FlatScene& out);
// Allocates on ctx's device, copies, and fills a new handle.  With `reuse` (a handle of ctx that no
// work in flight uses any more): its allocation takes the scene when it is large enough, and it is
// released otherwise; either way `reuse` must not be used again.
int place_scene(rtk_context* ctx, const FlatScene& flat, rtk_scene_handle** out, rtk_scene_handle* reuse = nullptr);

// Which tiles of the frame a launch renders and where they go: the host-side form of RenderTarget.
struct Partition {
  uint32_t layout = 0;  // 0 frame, 1 tile-major compact, 2 the rank's tile rows one after the other
  uint32_t step = 1, offset = 0;
  uint32_t band_rows = 1, band_period = 1, band_offset = 0, row0 = 0;
  uint64_t n_local_tiles = 0;
};
int partition_from_tiling(const rtk_camera* camera, const rtk_tiling* tiling, Partition& p, const char* who);
int partition_from_banding(const rtk_camera* camera, const rtk_banding* banding, Partition& p, const char* who);
// The tile rows first + k * period + [offset, offset + width), k = 0, 1, ..., clipped to the frame
// (width >= 1, offset + width <= period); compact: the outputs hold only these rows, one after the
// other, else whole frames.  What a group member with a share of its own renders.
int partition_from_rows(const rtk_camera* camera, uint32_t first, uint32_t period, uint32_t offset, uint32_t width, bool compact,
                        Partition& p, const char* who);
// Tile rows of such a partition (all of them, or those that lie above tile row `before`).
uint64_t rows_of_share(uint32_t rows, uint32_t period, uint32_t offset, uint32_t width);

// rtk_render_device for an arbitrary partition of the frame (one sample per pixel).
int render_device_partition(rtk_context* ctx, rtk_scene_handle* scene, const rtk_camera* camera, const Partition& p,
                            float* d_rgb_f32, uint8_t* d_rgba8, void* hip_stream);

}  // namespace rtk

#endif
