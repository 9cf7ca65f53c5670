// rtk_host.h -- the host half of the library that needs no GPU (rtk_host.cpp): error reporting, the
// scene handle's layout, flattening, the kernels' camera, and the arithmetic of tiles, bands and shares.
// Includes no HIP runtime header: a plain C++ compiler builds everything declared here.
#ifndef RTK_HOST_H
#define RTK_HOST_H

#include "../../include/rtk.h"
#include "rtk_device.h"

#include <memory>
#include <string>
#include <vector>

namespace rtk {

// Sets the thread's error text (rtk_last_error) and returns `code`.
int fail(int code, const std::string& msg);

}  // namespace rtk

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
  void* d_pattern_chain = nullptr;
  void* d_prim = nullptr;          // camera-ray filter records (rebuilt per camera)
  void* d_prim_ball = nullptr;     // camera-ray records of the wavefront-level cone test (rebuilt per camera)
  void* d_sph_ball = nullptr;
  void* d_cluster_ball = nullptr;
  void* d_super_ball = nullptr;
  void* d_cluster_plain = nullptr;
  void* d_light_grid = nullptr;
  void* d_grid_cells = nullptr;
  void* d_grid_entries = nullptr;
  void* d_grid_filter = nullptr;
  void* d_tg_cells = nullptr;
  void* d_tg_filter = nullptr;
  void* d_tg_slot = nullptr;
  void* d_prim_cluster = nullptr;  // camera-ray records of the cluster balls (rebuilt per camera)
  void* d_light_filter = nullptr;
  void* d_k_table = nullptr;
  void* d_sph_tab = nullptr;
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


// The camera as the kernels take it: the inverse transform by the reference's cofactor method, hoisted
// out of the pixel loop (src/scene/cameras.rs:52-69 inverts per pixel), and the origin.
// RTK_ERR_SINGULAR_CAMERA where the reference panics.
int make_dev_camera(const rtk_camera* camera, DevCamera& c);

inline uint32_t tiles_across(uint32_t width) { return (width + RTK_TILE_W - 1) / RTK_TILE_W; }
inline uint32_t tiles_down(uint32_t height) { return (height + RTK_TILE_H - 1) / RTK_TILE_H; }

// Deterministic 64-bit digest (FNV-1a) of everything flatten_scene produced for a scene: the bytes of
// every record array, the extra arrays, the allocation's layout and the kernels' scalar arguments.
// Two flattenings of one scene agree on it, whatever thread or build produced them.
uint64_t flat_scene_digest(const FlatScene& flat);

}  // namespace rtk

#endif
