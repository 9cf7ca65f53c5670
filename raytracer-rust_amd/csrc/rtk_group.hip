// rtk_group.hip -- one frame over several GPUs of a node from ONE process (rtk_group_* of
// include/rtk.h).  Host code only.
//
// Pixels are independent (src/scene/cameras.rs:75-82 has no cross-iteration state), so the frame
// shards with no data-path collective: member r renders whole tile rows -- bands dealt in rounds, of
// which the root may hold several per round, or one run of rows per member with boundaries laid by
// measurement (rtk_group_balance).  The one exchange step per frame brings the other members' rows to
// the root: for every run of rows one ncclSend on the member's communicator and one ncclRecv on the
// root's whose target is the run's final rows in the caller's frame, all inside one ncclGroupStart /
// ncclGroupEnd.  Rows of a run are contiguous in the row-major frame, so nothing is staged or
// de-interleaved on the root.  xGMI is point to point: every peer's bytes arrive over that peer's own
// link to the root.  Every entry point leaves the calling thread's current device as it found it.
//
// librccl is loaded at run time, when the first group of more than one member is created: a
// single-GPU user of the library never pays for loading it.  Where it cannot be loaded or its
// communicators do not come up, the same bands travel as peer-to-peer copies
// (hipMemcpyPeerAsync on the sending member's communication stream) instead; RTK_GROUP_EXCHANGE
// = "rccl" (fail rather than fall back) or "peer" (copies from the start) overrides the choice.

#include "rtk_internal.h"

#include <rccl/rccl.h>

#include <dlfcn.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <type_traits>
#include <vector>

using rtk::fail;
using rtk::fail_hip;

namespace {

// ---------------------------------------------------------------- RCCL, bound at run time
struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;
std::mutex g_rccl_mutex;

int load_rccl() {
  std::lock_guard<std::mutex> lock(g_rccl_mutex);
  if (g_rccl.lib) return RTK_OK;
  void* lib = nullptr;
  // a process that already holds an RCCL (e.g. the one PyTorch ships) keeps using that one
  for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
    lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    if (lib) break;
  }
  if (!lib) return fail(RTK_ERR_RCCL, std::string("rtk_group_create: cannot load librccl.so.1: ") + dlerror());
  Rccl r;
  r.lib = lib;
  auto bind = [&](const char* name, auto& fn) {
    fn = reinterpret_cast<std::remove_reference_t<decltype(fn)>>(dlsym(lib, name));
    return fn != nullptr;
  };
  if (!bind("ncclCommInitAll", r.CommInitAll) || !bind("ncclCommDestroy", r.CommDestroy) || !bind("ncclGroupStart", r.GroupStart) ||
      !bind("ncclGroupEnd", r.GroupEnd) || !bind("ncclSend", r.Send) || !bind("ncclRecv", r.Recv) ||
      !bind("ncclGetErrorString", r.GetErrorString))
    return fail(RTK_ERR_RCCL, "rtk_group_create: librccl.so.1 lacks a symbol the exchange needs");
  g_rccl = r;
  return RTK_OK;
}

int fail_rccl(ncclResult_t e, const char* what) {
  return fail(RTK_ERR_RCCL, std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(e) : "RCCL error"));
}
#define RTK_RCCL(call)                                  \
  do {                                                  \
    ncclResult_t r_ = (call);                           \
    if (r_ != ncclSuccess) return fail_rccl(r_, #call); \
  } while (0)

// One per distinct device of the group: its communicator and the stream the exchange runs on.
struct DeviceSlot {
  int device = -1;
  ncclComm_t comm = nullptr;
  int comm_rank = -1;
  hipStream_t comm_stream = nullptr;
};

struct Member {
  rtk_context* ctx = nullptr;
  rtk_scene_handle* scene = nullptr;
  int slot = -1;
  // their own tile rows of a frame, two buffers used alternately (the root only while the shares are
  // being measured: its rows of a frame go straight into the caller's buffers)
  float* band_rgb[2] = {nullptr, nullptr};
  uint8_t* band_px[2] = {nullptr, nullptr};
  size_t cap_rgb = 0, cap_px = 0;  // pixels
  hipEvent_t ev_rendered = nullptr;
  hipEvent_t ev_sent[2] = {nullptr, nullptr};
  bool sent_valid[2] = {false, false};
};

// A member's part of the frame: the tile rows k * period + [offset, offset + width), k = 0, 1, ...
struct Share {
  uint32_t offset = 0, width = 0;
};

// The calling thread's current device is put back when an entry point returns, on every path.
struct DeviceGuard {
  int device = -1;
  DeviceGuard() {
    if (hipGetDevice(&device) != hipSuccess) {
      (void)hipGetLastError();
      device = -1;
    }
  }
  ~DeviceGuard() {
    if (device >= 0) (void)hipSetDevice(device);
  }
  DeviceGuard(const DeviceGuard&) = delete;
  DeviceGuard& operator=(const DeviceGuard&) = delete;
};

}  // namespace

struct rtk_group {
  std::vector<DeviceSlot> slots;
  std::vector<Member> members;
  uint64_t frame = 0;
  // the partition: interleaved bands (bands_per_member, root_share), or one run of rows per member
  uint32_t bands_per_member = 0;   // 0: the library's choice
  uint32_t root_share = 1;         // bands the root holds in every round of the deal
  std::vector<uint32_t> rows;      // members + 1 boundaries in tile rows, for frames of rows_tiles tile rows; empty: bands
  uint32_t rows_tiles = 0;
  bool use_rccl = false;          // else peer-to-peer copies
  hipEvent_t ev_enter = nullptr;                // the caller's stream has reached the call
  hipEvent_t ev_x0 = nullptr, ev_x1 = nullptr;  // the exchange of the last frame on the root's communication stream
  bool exchanged = false;                       // ... was there one
  hipStream_t last_stream = nullptr;
  // the last frame, for rtk_group_get_info
  uint32_t last_messages = 0;
  uint64_t last_bytes = 0;
  std::vector<uint32_t> last_rows;  // tile rows per member
};

namespace {

// Bands per member unless the caller chooses.  More bands mix cheap sky rows and expensive ground rows
// better (10,000-sphere scene at 3840x2160 on 8 members: the slowest member's kernels take 0.145 ms
// with 4 bands each, 0.129 ms with 32); every band of every peer is one message per output, and a
// send/receive pair costs RCCL about 6 us beyond its bytes.  Two per member from 8 members on: 14
// messages per output and frame.
uint32_t default_bands_per_member(size_t members) { return members >= 8 ? 2u : members >= 4 ? 3u : 4u; }

uint32_t tiles_down(uint32_t height) { return (height + RTK_TILE_H - 1) / RTK_TILE_H; }

// Shares of a frame of `height` pixel rows: the period of the deal and each member's rows in it.
// Rows mode (rtk_group_set_rows / rtk_group_balance, for frames of the height they were made for):
// one run of rows per member, the period is the whole frame.  Otherwise bands of m tile rows dealt
// in rounds of members - 1 + root_share, of which the root holds the first root_share.
uint32_t layout_shares(const rtk_group* g, uint32_t height, std::vector<Share>& shares) {
  const size_t n = g->members.size();
  const uint32_t tiles_y = tiles_down(height);
  shares.assign(n, Share{});
  if (!g->rows.empty() && g->rows_tiles == tiles_y) {
    for (size_t r = 0; r < n; ++r) shares[r] = {g->rows[r], g->rows[r + 1] - g->rows[r]};
    return std::max(1u, tiles_y);
  }
  const uint32_t per_member = g->bands_per_member ? g->bands_per_member : default_bands_per_member(n);
  const uint64_t virtual_members = (uint64_t)n - 1u + g->root_share;
  const uint32_t m = std::max<uint32_t>(1u, (uint32_t)(tiles_y / (virtual_members * per_member)));
  shares[0] = {0u, g->root_share * m};
  for (size_t r = 1; r < n; ++r) shares[r] = {(uint32_t)((g->root_share + r - 1u) * m), m};
  return (uint32_t)(virtual_members * m);
}

uint32_t band_tile_rows_for(const rtk_group* g, uint32_t height) {
  std::vector<Share> shares;
  layout_shares(g, height, shares);
  return g->members.size() > 1 ? shares[1].width : shares[0].width;
}

void release_scenes(rtk_group* g) {
  for (Member& m : g->members) {
    if (m.scene && m.ctx) {
      (void)hipSetDevice(m.ctx->device);
      (void)hipDeviceSynchronize();
      rtk_scene_release(m.ctx, m.scene);
    }
    m.scene = nullptr;
  }
}

// Band buffers of a member for `pixels` pixels of the outputs that are asked for: a call that wants
// the 8-bit frame only neither allocates nor moves the 12 bytes per pixel of the FP32 one.
int ensure_band_buffers(Member& m, size_t pixels, bool rgb, bool px) {
  const bool grow_rgb = rgb && pixels > m.cap_rgb, grow_px = px && pixels > m.cap_px;
  if (!grow_rgb && !grow_px) return RTK_OK;
  RTK_HIP(hipSetDevice(m.ctx->device));
  RTK_HIP(hipDeviceSynchronize());  // a send of an earlier frame may still read the old buffers
  for (int b = 0; b < 2; ++b) {
    if (grow_rgb && m.band_rgb[b]) (void)hipFree(m.band_rgb[b]), m.band_rgb[b] = nullptr;
    if (grow_px && m.band_px[b]) (void)hipFree(m.band_px[b]), m.band_px[b] = nullptr;
    m.sent_valid[b] = false;
  }
  if (grow_rgb) m.cap_rgb = 0;
  if (grow_px) m.cap_px = 0;
  for (int b = 0; b < 2; ++b) {
    if (grow_rgb) RTK_HIP(hipMalloc((void**)&m.band_rgb[b], pixels * 3 * sizeof(float)));
    if (grow_px) RTK_HIP(hipMalloc((void**)&m.band_px[b], pixels * 4));
  }
  if (grow_rgb) m.cap_rgb = pixels;
  if (grow_px) m.cap_px = pixels;
  return RTK_OK;
}

}  // namespace

extern "C" {

int rtk_group_create(const int* device_ids, int n_members, rtk_group** out) {
  if (!out) return fail(RTK_ERR_INVALID, "rtk_group_create: null out pointer");
  *out = nullptr;
  if (!device_ids || n_members <= 0 || n_members > 1024) return fail(RTK_ERR_INVALID, "rtk_group_create: bad device list");
  DeviceGuard keep_device;
  rtk_group* g = new (std::nothrow) rtk_group();
  if (!g) return fail(RTK_ERR_NOMEM, "rtk_group_create: out of host memory");
  int rc = RTK_OK;
  try {
    g->members.resize((size_t)n_members);
    for (int r = 0; r < n_members && rc == RTK_OK; ++r) {
      Member& m = g->members[(size_t)r];
      rc = rtk_create(device_ids[r], &m.ctx);
      if (rc != RTK_OK) break;
      int slot = -1;
      for (size_t s = 0; s < g->slots.size(); ++s)
        if (g->slots[s].device == device_ids[r]) slot = (int)s;
      if (slot < 0) {
        DeviceSlot ds;
        ds.device = device_ids[r];
        g->slots.push_back(ds);
        slot = (int)g->slots.size() - 1;
      }
      m.slot = slot;
    }
    if (rc == RTK_OK && n_members > 1) {
      const char* want = std::getenv("RTK_GROUP_EXCHANGE");
      const bool force_rccl = want && std::strcmp(want, "rccl") == 0, force_peer = want && std::strcmp(want, "peer") == 0;
      if (want && *want && !force_rccl && !force_peer)
        rc = fail(RTK_ERR_INVALID, "rtk_group_create: RTK_GROUP_EXCHANGE must be \"rccl\" or \"peer\"");
      if (rc == RTK_OK && !force_peer) {
        rc = load_rccl();
        std::vector<ncclComm_t> comms(g->slots.size(), nullptr);
        if (rc == RTK_OK) {
          // one communicator per distinct device, ranks in order of first appearance
          std::vector<int> devices;
          for (const DeviceSlot& s : g->slots) devices.push_back(s.device);
          const ncclResult_t e = g_rccl.CommInitAll(comms.data(), (int)devices.size(), devices.data());
          if (e != ncclSuccess) rc = fail_rccl(e, "ncclCommInitAll");
        }
        if (rc == RTK_OK) {
          g->use_rccl = true;
          for (size_t s = 0; s < g->slots.size(); ++s) g->slots[s].comm = comms[s], g->slots[s].comm_rank = (int)s;
        } else if (!force_rccl) {
          rc = RTK_OK;  // peer-to-peer copies need nothing but HIP
        }
      }
      for (size_t s = 0; s < g->slots.size() && rc == RTK_OK; ++s) {
        hipError_t he = hipSetDevice(g->slots[s].device);
        if (he == hipSuccess) he = hipStreamCreateWithFlags(&g->slots[s].comm_stream, hipStreamNonBlocking);
        // copies between two devices: each must be allowed to address the other's memory
        for (size_t o = 0; o < g->slots.size() && he == hipSuccess && !g->use_rccl; ++o) {
          if (o == s) continue;
          int can = 0;
          he = hipDeviceCanAccessPeer(&can, g->slots[s].device, g->slots[o].device);
          if (he == hipSuccess && can) {
            he = hipDeviceEnablePeerAccess(g->slots[o].device, 0);
            if (he == hipErrorPeerAccessAlreadyEnabled) {
              (void)hipGetLastError();
              he = hipSuccess;
            }
          }
        }
        if (he != hipSuccess) rc = fail_hip(he, "rtk_group_create: communication stream");
      }
      for (int r = 0; r < n_members && rc == RTK_OK; ++r) {
        Member& m = g->members[(size_t)r];
        hipError_t he = hipSetDevice(m.ctx->device);
        if (he == hipSuccess) he = hipEventCreateWithFlags(&m.ev_rendered, hipEventDisableTiming);
        for (int b = 0; b < 2 && he == hipSuccess; ++b) he = hipEventCreateWithFlags(&m.ev_sent[b], hipEventDisableTiming);
        if (he != hipSuccess) rc = fail_hip(he, "rtk_group_create: events");
      }
    }
    if (rc == RTK_OK) {
      hipError_t he = hipSetDevice(g->members[0].ctx->device);
      if (he == hipSuccess) he = hipEventCreateWithFlags(&g->ev_enter, hipEventDisableTiming);
      if (he == hipSuccess) he = hipEventCreate(&g->ev_x0);
      if (he == hipSuccess) he = hipEventCreate(&g->ev_x1);
      if (he != hipSuccess) rc = fail_hip(he, "rtk_group_create: events");
    }
  } catch (const std::bad_alloc&) {
    rc = fail(RTK_ERR_NOMEM, "rtk_group_create: out of host memory");
  }
  if (rc != RTK_OK) {
    const std::string keep = rtk_last_error();
    rtk_group_destroy(g);
    return fail(rc, keep);
  }
  *out = g;
  return RTK_OK;
}

void rtk_group_destroy(rtk_group* g) {
  if (!g) return;
  DeviceGuard keep_device;
  for (DeviceSlot& s : g->slots) {
    (void)hipSetDevice(s.device);
    if (s.comm_stream) (void)hipStreamSynchronize(s.comm_stream);
  }
  release_scenes(g);
  for (Member& m : g->members) {
    if (!m.ctx) continue;
    (void)hipSetDevice(m.ctx->device);
    (void)hipDeviceSynchronize();
    for (int b = 0; b < 2; ++b) {
      if (m.band_rgb[b]) (void)hipFree(m.band_rgb[b]);
      if (m.band_px[b]) (void)hipFree(m.band_px[b]);
      if (m.ev_sent[b]) (void)hipEventDestroy(m.ev_sent[b]);
    }
    if (m.ev_rendered) (void)hipEventDestroy(m.ev_rendered);
  }
  if (!g->members.empty() && g->members[0].ctx) {
    (void)hipSetDevice(g->members[0].ctx->device);
    if (g->ev_enter) (void)hipEventDestroy(g->ev_enter);
    if (g->ev_x0) (void)hipEventDestroy(g->ev_x0);
    if (g->ev_x1) (void)hipEventDestroy(g->ev_x1);
  }
  for (DeviceSlot& s : g->slots) {
    (void)hipSetDevice(s.device);
    if (s.comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(s.comm);
    if (s.comm_stream) (void)hipStreamDestroy(s.comm_stream);
  }
  for (Member& m : g->members)
    if (m.ctx) rtk_destroy(m.ctx);
  delete g;
}

int rtk_group_size(const rtk_group* g) { return g ? (int)g->members.size() : 0; }

const char* rtk_group_exchange(const rtk_group* g) {
  if (!g || g->members.size() < 2) return "none";
  return g->use_rccl ? "rccl" : "peer copies";
}

uint32_t rtk_group_band_tile_rows(const rtk_group* g, uint32_t height) {
  if (!g || height == 0) return 0;
  return band_tile_rows_for(g, height);
}

int rtk_group_set_bands_per_member(rtk_group* g, uint32_t bands_per_member) {
  if (!g) return fail(RTK_ERR_INVALID, "rtk_group_set_bands_per_member: null group");
  if (bands_per_member > 65536u) return fail(RTK_ERR_INVALID, "rtk_group_set_bands_per_member: more than 65,536 bands per member");
  g->bands_per_member = bands_per_member;
  return RTK_OK;
}

int rtk_group_set_root_share(rtk_group* g, uint32_t root_share) {
  if (!g) return fail(RTK_ERR_INVALID, "rtk_group_set_root_share: null group");
  if (root_share > 1024u) return fail(RTK_ERR_INVALID, "rtk_group_set_root_share: more than 1,024 bands per round for the root");
  g->root_share = root_share ? root_share : 1u;
  return RTK_OK;
}

int rtk_group_set_rows(rtk_group* g, uint32_t height, const uint32_t* first_tile_row) {
  if (!g) return fail(RTK_ERR_INVALID, "rtk_group_set_rows: null group");
  if (!first_tile_row) {  // back to bands
    g->rows.clear();
    g->rows_tiles = 0;
    return RTK_OK;
  }
  if (height == 0) return fail(RTK_ERR_INVALID, "rtk_group_set_rows: zero height");
  const size_t n = g->members.size();
  const uint32_t tiles_y = tiles_down(height);
  if (first_tile_row[0] != 0u || first_tile_row[n] != tiles_y)
    return fail(RTK_ERR_INVALID, "rtk_group_set_rows: the boundaries must start at 0 and end at the frame's tile rows");
  for (size_t r = 0; r < n; ++r)
    if (first_tile_row[r] > first_tile_row[r + 1]) return fail(RTK_ERR_INVALID, "rtk_group_set_rows: boundaries must not decrease");
  try {
    g->rows.assign(first_tile_row, first_tile_row + n + 1);
  } catch (const std::bad_alloc&) {
    return fail(RTK_ERR_NOMEM, "rtk_group_set_rows: out of host memory");
  }
  g->rows_tiles = tiles_y;
  return RTK_OK;
}

int rtk_group_upload(rtk_group* g, const rtk_scene* scene) {
  if (!g || !scene) return fail(RTK_ERR_INVALID, "rtk_group_upload: null argument");
  DeviceGuard keep_device;
  rtk::FlatScene flat;  // flattened once, placed on every member's device
  int rc = rtk::flatten_scene(scene, flat);
  if (rc != RTK_OK) return rc;
  release_scenes(g);
  for (Member& m : g->members)
    if ((rc = rtk::place_scene(m.ctx, flat, &m.scene)) != RTK_OK) {
      const std::string keep = rtk_last_error();
      release_scenes(g);
      return fail(rc, keep);
    }
  return RTK_OK;
}

int rtk_group_render_device(rtk_group* g, const rtk_camera* camera, float* d_rgb_f32, uint8_t* d_rgba8, void* hip_stream) {
  if (!g || !camera) return fail(RTK_ERR_INVALID, "rtk_group_render_device: null argument");
  if (camera->width == 0 || camera->height == 0) return fail(RTK_ERR_INVALID, "camera has zero size");
  Member& root = g->members[0];
  if (!root.scene) return fail(RTK_ERR_INVALID, "rtk_group_render_device: no scene (call rtk_group_upload first)");
  DeviceGuard keep_device;
  const size_t n = g->members.size();
  RTK_HIP(hipSetDevice(root.ctx->device));
  hipStream_t st0 = hip_stream ? (hipStream_t)hip_stream : root.ctx->stream;
  g->last_stream = st0;
  const uint32_t width = camera->width, height = camera->height;
  const uint32_t tiles_y = tiles_down(height);
  g->last_messages = 0, g->last_bytes = 0;
  if (n == 1) {
    g->exchanged = false;
    ++g->frame;
    try {
      g->last_rows.assign(1, tiles_y);
    } catch (const std::bad_alloc&) {
      return fail(RTK_ERR_NOMEM, "rtk_group_render_device: out of host memory");
    }
    return rtk_render_device(root.ctx, root.scene, camera, nullptr, d_rgb_f32, d_rgba8, st0);
  }
  std::vector<Share> shares;
  uint32_t period = 1;
  try {
    period = layout_shares(g, height, shares);
    g->last_rows.assign(n, 0u);
  } catch (const std::bad_alloc&) {
    return fail(RTK_ERR_NOMEM, "rtk_group_render_device: out of host memory");
  }
  for (size_t r = 0; r < n; ++r) g->last_rows[r] = (uint32_t)rtk::rows_of_share(tiles_y, period, shares[r].offset, shares[r].width);
  const int b = (int)(g->frame & 1u);
  const bool exchange = d_rgb_f32 != nullptr || d_rgba8 != nullptr;
  int rc;

  // every other member: its rows into its own buffer, on its own stream
  for (size_t r = 1; r < n; ++r) {
    Member& m = g->members[r];
    if (g->last_rows[r] == 0u) continue;  // more members than the frame has bands
    rtk::Partition p;
    if ((rc = rtk::partition_from_rows(camera, 0u, period, shares[r].offset, shares[r].width, true, p, "rtk_group_render_device")) != RTK_OK)
      return rc;
    const size_t pixels = (size_t)g->last_rows[r] * RTK_TILE_H * width;
    if (exchange && (rc = ensure_band_buffers(m, pixels, d_rgb_f32 != nullptr, d_rgba8 != nullptr)) != RTK_OK) return rc;
    RTK_HIP(hipSetDevice(m.ctx->device));
    // this buffer's rows of two frames ago must have left before they are overwritten
    if (m.sent_valid[b]) RTK_HIP(hipStreamWaitEvent(m.ctx->stream, m.ev_sent[b], 0));
    if ((rc = rtk::render_device_partition(m.ctx, m.scene, camera, p, d_rgb_f32 ? m.band_rgb[b] : nullptr,
                                           d_rgba8 ? m.band_px[b] : nullptr, nullptr)) != RTK_OK)
      return rc;
    if (exchange) {
      RTK_HIP(hipEventRecord(m.ev_rendered, m.ctx->stream));
      RTK_HIP(hipStreamWaitEvent(g->slots[(size_t)m.slot].comm_stream, m.ev_rendered, 0));
    }
  }

  // the root: what the caller's stream did with the frame buffers before this call comes first,
  // then its own rows (on the caller's stream) and the incoming ones (on the communication stream)
  // fill disjoint rows of the frame side by side
  DeviceSlot& slot0 = g->slots[(size_t)root.slot];
  RTK_HIP(hipSetDevice(root.ctx->device));
  if (exchange) {
    RTK_HIP(hipEventRecord(g->ev_enter, st0));
    RTK_HIP(hipStreamWaitEvent(slot0.comm_stream, g->ev_enter, 0));
  }
  if (g->last_rows[0] != 0u) {
    rtk::Partition p;
    if ((rc = rtk::partition_from_rows(camera, 0u, period, shares[0].offset, shares[0].width, false, p, "rtk_group_render_device")) != RTK_OK)
      return rc;
    if ((rc = rtk::render_device_partition(root.ctx, root.scene, camera, p, d_rgb_f32, d_rgba8, st0)) != RTK_OK) return rc;
  }
  g->exchanged = exchange;
  if (exchange) {
    RTK_HIP(hipEventRecord(g->ev_x0, slot0.comm_stream));
    // Visits every run of rows of every other member: where it lies in the member's buffer (`from`),
    // where its rows go in the frame (`to`), both in pixels.  Rows of one run are contiguous in the
    // row-major frame, so a run is ONE message per output, landed in place.
    auto for_each_run = [&](auto&& visit) {
      for (size_t r = 1; r < n; ++r) {
        if (shares[r].width == 0u) continue;
        size_t local_row = 0;  // pixel rows of this member's buffer consumed so far
        for (uint64_t t0 = shares[r].offset; t0 < tiles_y; t0 += period) {
          const uint64_t t1 = std::min<uint64_t>(tiles_y, t0 + shares[r].width);
          const size_t y0 = (size_t)t0 * RTK_TILE_H, y1 = std::min<size_t>(height, (size_t)t1 * RTK_TILE_H);
          if (!visit(g->members[r], (y1 - y0) * width, local_row * width, y0 * width)) return false;
          local_row += (size_t)(t1 - t0) * RTK_TILE_H;
        }
      }
      return true;
    };
    const uint32_t outputs = (d_rgb_f32 ? 1u : 0u) + (d_rgba8 ? 1u : 0u);
    const uint64_t bytes_per_pixel = (d_rgb_f32 ? 12u : 0u) + (d_rgba8 ? 4u : 0u);
    if (g->use_rccl) {
      RTK_RCCL(g_rccl.GroupStart());
      ncclResult_t first_bad = ncclSuccess;
      for_each_run([&](Member& m, size_t count, size_t from, size_t to) {
        DeviceSlot& slot = g->slots[(size_t)m.slot];
        if (d_rgb_f32) {
          first_bad = g_rccl.Send(m.band_rgb[b] + from * 3, count * 3, ncclFloat, slot0.comm_rank, slot.comm, slot.comm_stream);
          if (first_bad == ncclSuccess)
            first_bad = g_rccl.Recv(d_rgb_f32 + to * 3, count * 3, ncclFloat, slot.comm_rank, slot0.comm, slot0.comm_stream);
        }
        if (d_rgba8 && first_bad == ncclSuccess) {
          first_bad = g_rccl.Send(m.band_px[b] + from * 4, count * 4, ncclUint8, slot0.comm_rank, slot.comm, slot.comm_stream);
          if (first_bad == ncclSuccess)
            first_bad = g_rccl.Recv(d_rgba8 + to * 4, count * 4, ncclUint8, slot.comm_rank, slot0.comm, slot0.comm_stream);
        }
        g->last_messages += outputs, g->last_bytes += count * bytes_per_pixel;
        return first_bad == ncclSuccess;
      });
      const ncclResult_t end = g_rccl.GroupEnd();  // always closes the group, even after a failed call inside it
      if (first_bad != ncclSuccess) return fail_rccl(first_bad, "ncclSend / ncclRecv");
      if (end != ncclSuccess) return fail_rccl(end, "ncclGroupEnd");
    } else {
      // each member pushes its rows over its own link, on its own communication stream, once the
      // caller's stream has let go of the frame buffers
      for (size_t r = 1; r < n; ++r) {
        Member& m = g->members[r];
        RTK_HIP(hipSetDevice(m.ctx->device));
        RTK_HIP(hipStreamWaitEvent(g->slots[(size_t)m.slot].comm_stream, g->ev_enter, 0));
      }
      hipError_t first_bad = hipSuccess;
      const int root_device = root.ctx->device;
      for_each_run([&](Member& m, size_t count, size_t from, size_t to) {
        hipStream_t cs = g->slots[(size_t)m.slot].comm_stream;
        first_bad = hipSetDevice(m.ctx->device);
        if (d_rgb_f32 && first_bad == hipSuccess)
          first_bad = hipMemcpyPeerAsync(d_rgb_f32 + to * 3, root_device, m.band_rgb[b] + from * 3, m.ctx->device,
                                         count * 3 * sizeof(float), cs);
        if (d_rgba8 && first_bad == hipSuccess)
          first_bad = hipMemcpyPeerAsync(d_rgba8 + to * 4, root_device, m.band_px[b] + from * 4, m.ctx->device, count * 4, cs);
        g->last_messages += outputs, g->last_bytes += count * bytes_per_pixel;
        return first_bad == hipSuccess;
      });
      if (first_bad != hipSuccess) return fail_hip(first_bad, "hipMemcpyPeerAsync");
    }
    for (size_t r = 1; r < n; ++r) {
      Member& m = g->members[r];
      RTK_HIP(hipSetDevice(m.ctx->device));
      RTK_HIP(hipEventRecord(m.ev_sent[b], g->slots[(size_t)m.slot].comm_stream));
      m.sent_valid[b] = true;
    }
    RTK_HIP(hipSetDevice(root.ctx->device));
    if (!g->use_rccl)  // the copies ran on the members' streams: the root's waits for each of them
      for (size_t r = 1; r < n; ++r) RTK_HIP(hipStreamWaitEvent(slot0.comm_stream, g->members[r].ev_sent[b], 0));
    RTK_HIP(hipEventRecord(g->ev_x1, slot0.comm_stream));
    RTK_HIP(hipStreamWaitEvent(st0, g->ev_x1, 0));  // the caller's stream sees the whole frame
  }
  ++g->frame;
  return RTK_OK;
}

// Shares of equal cost, from measurement.  Every member renders a run of rows of its own (the root the
// top of the frame) into a buffer of its own, one member after the other, and its kernel time is read
// back; the time per tile row of each run is taken as the cost of its rows, and new boundaries are
// laid so that the largest of
//     a member's kernel time   and, for every member but the root,   its bytes / the link rate
// is as small as possible (the root's rows do not travel): a bisection on that bound, the rows handed
// out from the top in member order, each member taking rows while it stays within the bound.  A few
// rounds settle it; the best boundaries measured are kept (rtk_group_set_rows).
int rtk_group_balance(rtk_group* g, const rtk_camera* camera, int with_rgb_f32, int with_rgba8, uint32_t rounds, double link_gbps) {
  if (!g || !camera) return fail(RTK_ERR_INVALID, "rtk_group_balance: null argument");
  if (camera->width == 0 || camera->height == 0) return fail(RTK_ERR_INVALID, "camera has zero size");
  if (!with_rgb_f32 && !with_rgba8) return fail(RTK_ERR_INVALID, "rtk_group_balance: no output to balance for");
  if (!(link_gbps >= 0.0)) return fail(RTK_ERR_INVALID, "rtk_group_balance: negative link rate");
  if (!g->members[0].scene) return fail(RTK_ERR_INVALID, "rtk_group_balance: no scene (call rtk_group_upload first)");
  const size_t n = g->members.size();
  if (n == 1) return RTK_OK;
  DeviceGuard keep_device;
  for (Member& m : g->members) {  // frames in flight use the buffers the measurement renders into
    RTK_HIP(hipSetDevice(m.ctx->device));
    RTK_HIP(hipDeviceSynchronize());
  }
  if (rounds == 0) rounds = 4;
  if (link_gbps == 0.0) {
    link_gbps = 100.0;  // what RCCL's send / receive is assumed to reach on one xGMI link unless told otherwise
    if (const char* v = std::getenv("RTK_GROUP_LINK_GBPS")) {
      char* end = nullptr;
      const double x = std::strtod(v, &end);
      if (end != v && x > 0.0) link_gbps = x;
    }
  }
  const uint32_t width = camera->width, height = camera->height, tiles_y = tiles_down(height);
  const double link_ms_per_row = (double)width * RTK_TILE_H * ((with_rgb_f32 ? 12.0 : 0.0) + (with_rgba8 ? 4.0 : 0.0)) / (link_gbps * 1.0e6);
  try {
    std::vector<double> cost((size_t)tiles_y, 1.0);  // ms per tile row, as far as known
    std::vector<uint32_t> bounds(n + 1), best_bounds;
    for (size_t r = 0; r <= n; ++r) bounds[r] = (uint32_t)((uint64_t)tiles_y * r / n);
    double best = 0.0;
    // rows from the top, member after member, each within `bound`; true when the frame is used up
    auto lay = [&](double bound, std::vector<uint32_t>& out) {
      uint32_t row = 0;
      out[0] = 0;
      for (size_t r = 0; r < n; ++r) {
        double acc = 0.0;
        uint32_t h = 0;
        while (row < tiles_y && acc + cost[row] <= bound && (r == 0 || (h + 1u) * link_ms_per_row <= bound)) acc += cost[row], ++h, ++row;
        out[r + 1] = row;
      }
      return row == tiles_y;
    };
    for (uint32_t round = 0; round <= rounds; ++round) {
      // measure the current boundaries: every member's run alone, compact, into its own buffers
      double worst = 0.0;
      for (size_t r = 0; r < n; ++r) {
        Member& m = g->members[r];
        const uint32_t h = bounds[r + 1] - bounds[r];
        if (h == 0) continue;
        rtk::Partition p;
        int rc = rtk::partition_from_rows(camera, 0u, std::max(1u, tiles_y), bounds[r], h, true, p, "rtk_group_balance");
        if (rc != RTK_OK) return rc;
        if ((rc = ensure_band_buffers(m, (size_t)h * RTK_TILE_H * width, with_rgb_f32 != 0, with_rgba8 != 0)) != RTK_OK) return rc;
        RTK_HIP(hipSetDevice(m.ctx->device));
        double ms = 0.0;
        for (int rep = 0; rep < 3; ++rep) {  // the fastest of three: the first launch of a shape pays for its set-up
          if ((rc = rtk::render_device_partition(m.ctx, m.scene, camera, p, with_rgb_f32 ? m.band_rgb[0] : nullptr,
                                                 with_rgba8 ? m.band_px[0] : nullptr, nullptr)) != RTK_OK)
            return rc;
          rtk_stats st;
          if ((rc = rtk_get_stats(m.ctx, &st)) != RTK_OK) return rc;  // waits for the launch
          ms = rep == 0 ? st.kernel_ms : std::min(ms, st.kernel_ms);
        }
        for (uint32_t row = bounds[r]; row < bounds[r + 1]; ++row) cost[row] = ms / h;
        worst = std::max(worst, std::max(ms, r == 0 ? 0.0 : h * link_ms_per_row));
      }
      if (best_bounds.empty() || worst < best) best = worst, best_bounds = bounds;
      if (round == rounds) break;
      // new boundaries for the costs known now
      double lo = 0.0, hi = 0.0;
      for (double c : cost) hi += c;
      hi += tiles_y * link_ms_per_row;
      std::vector<uint32_t> trial(n + 1);
      for (int it = 0; it < 50; ++it) {
        const double mid = 0.5 * (lo + hi);
        if (lay(mid, trial)) hi = mid; else lo = mid;
      }
      if (!lay(hi, trial)) break;  // cannot happen: hi admits the whole frame for the root alone
      trial[n] = tiles_y;
      if (trial == bounds) break;  // settled
      bounds = trial;
    }
    // the root's band buffers were only needed here
    Member& root = g->members[0];
    RTK_HIP(hipSetDevice(root.ctx->device));
    RTK_HIP(hipDeviceSynchronize());
    for (int b = 0; b < 2; ++b) {
      if (root.band_rgb[b]) (void)hipFree(root.band_rgb[b]), root.band_rgb[b] = nullptr;
      if (root.band_px[b]) (void)hipFree(root.band_px[b]), root.band_px[b] = nullptr;
    }
    root.cap_rgb = root.cap_px = 0;
    return rtk_group_set_rows(g, height, best_bounds.data());
  } catch (const std::bad_alloc&) {
    return fail(RTK_ERR_NOMEM, "rtk_group_balance: out of host memory");
  }
}

int rtk_group_get_stats(rtk_group* g, rtk_stats* stats) {
  if (!g || !stats) return fail(RTK_ERR_INVALID, "rtk_group_get_stats: null argument");
  std::memset(stats, 0, sizeof *stats);
  if (g->frame == 0) return RTK_OK;
  DeviceGuard keep_device;
  for (size_t r = 0; r < g->members.size(); ++r) {
    if (r < g->last_rows.size() && g->last_rows[r] == 0u) continue;  // rendered nothing in the last frame
    rtk_stats s;
    const int rc = rtk_get_stats(g->members[r].ctx, &s);  // waits for the member's last launch
    if (rc != RTK_OK) return rc;
    stats->rays_primary += s.rays_primary;
    stats->rays_shadow += s.rays_shadow;
    stats->rays_secondary += s.rays_secondary;
    stats->exact_tests += s.exact_tests;
    stats->kernel_ms = std::max(stats->kernel_ms, s.kernel_ms);
    stats->rays_ms = std::max(stats->rays_ms, s.rays_ms);
  }
  Member& root = g->members[0];
  RTK_HIP(hipSetDevice(root.ctx->device));
  if (g->last_stream) RTK_HIP(hipStreamSynchronize(g->last_stream));
  if (g->exchanged) {
    RTK_HIP(hipEventSynchronize(g->ev_x1));
    float ms = 0.f;
    RTK_HIP(hipEventElapsedTime(&ms, g->ev_x0, g->ev_x1));
    stats->gather_ms = ms;
  }
  return RTK_OK;
}

int rtk_group_get_info(rtk_group* g, rtk_group_info* info) {
  if (!g || !info) return fail(RTK_ERR_INVALID, "rtk_group_get_info: null argument");
  std::memset(info, 0, sizeof *info);
  info->n_members = (uint32_t)g->members.size();
  info->exchange = g->members.size() < 2 ? 0u : g->use_rccl ? 1u : 2u;
  info->root_share = g->root_share;
  info->rows_mode = g->rows.empty() ? 0u : 1u;
  if (g->frame == 0) return RTK_OK;
  DeviceGuard keep_device;
  info->messages = g->last_messages;
  info->bytes_into_root = g->last_bytes;
  for (size_t r = 0; r < g->members.size() && r < RTK_GROUP_INFO_MEMBERS; ++r) {
    info->share_tile_rows[r] = r < g->last_rows.size() ? g->last_rows[r] : 0u;
    if (info->share_tile_rows[r] == 0u) continue;
    rtk_stats s;
    const int rc = rtk_get_stats(g->members[r].ctx, &s);  // waits for the member's last launch
    if (rc != RTK_OK) return rc;
    info->share_ms[r] = s.kernel_ms;
    info->share_rays[r] = s.rays_primary + s.rays_shadow + s.rays_secondary;
  }
  if (g->exchanged) {
    RTK_HIP(hipSetDevice(g->members[0].ctx->device));
    RTK_HIP(hipEventSynchronize(g->ev_x1));
    float ms = 0.f;
    RTK_HIP(hipEventElapsedTime(&ms, g->ev_x0, g->ev_x1));
    info->gather_ms = ms;
  }
  return RTK_OK;
}

}  // extern "C"
