// rtk_group.hip -- one frame over several GPUs of a node from ONE process (rtk_group_* of
// include/rtk.h).  Host code only.
//
// Pixels are independent (src/scene/cameras.rs:75-82 has no cross-iteration state), so the frame
// shards with no data-path collective: member r renders the bands of whole tile rows dealt to r
// (rtk_banding).  The one exchange step per frame brings the other members' rows to the root:
// for every band one ncclSend on the member's communicator and one ncclRecv on the root's whose
// target is the band's final rows in the caller's frame, all inside one ncclGroupStart /
// ncclGroupEnd.  Rows of whole bands are contiguous in the row-major frame, so nothing is staged
// or de-interleaved on the root.  xGMI is point to point: every peer's bytes arrive over that
// peer's own link to the root.
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
  // members other than the root: their own tile rows of a frame, two buffers used alternately
  float* band_rgb[2] = {nullptr, nullptr};
  uint8_t* band_px[2] = {nullptr, nullptr};
  size_t cap_pixels = 0;
  hipEvent_t ev_rendered = nullptr;
  hipEvent_t ev_sent[2] = {nullptr, nullptr};
  bool sent_valid[2] = {false, false};
};

}  // namespace

struct rtk_group {
  std::vector<DeviceSlot> slots;
  std::vector<Member> members;
  uint64_t frame = 0;
  uint32_t bands_per_member = 0;  // 0: the library's choice
  bool use_rccl = false;          // else peer-to-peer copies
  hipEvent_t ev_enter = nullptr;                // the caller's stream has reached the call
  hipEvent_t ev_x0 = nullptr, ev_x1 = nullptr;  // the exchange of the last frame on the root's communication stream
  bool exchanged = false;                       // ... was there one
  hipStream_t last_stream = nullptr;
};

namespace {

// Bands per member unless the caller chooses: enough to mix cheap sky rows and expensive ground
// rows (10,000-sphere scene at 3840x2160 on 8 members: the slowest member's kernels take 0.145 ms
// with 4 bands each, 0.129 ms with 32), few enough messages per frame (a send/receive pair costs
// RCCL about 6 us on top of its bytes: 0.5 ms per frame with 4 bands each, 3 ms with 32).
constexpr uint32_t kBandsPerMember = 4;

uint32_t band_tile_rows_for(const rtk_group* g, uint32_t height) {
  const uint32_t per_member = g->bands_per_member ? g->bands_per_member : kBandsPerMember;
  const uint32_t tiles_y = (height + RTK_TILE_H - 1) / RTK_TILE_H;
  return std::max<uint32_t>(1u, (uint32_t)(tiles_y / (g->members.size() * (size_t)per_member)));
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

int ensure_band_buffers(Member& m, size_t pixels) {
  if (pixels <= m.cap_pixels) return RTK_OK;
  RTK_HIP(hipSetDevice(m.ctx->device));
  RTK_HIP(hipDeviceSynchronize());  // a send of an earlier frame may still read the old buffers
  for (int b = 0; b < 2; ++b) {
    if (m.band_rgb[b]) (void)hipFree(m.band_rgb[b]);
    if (m.band_px[b]) (void)hipFree(m.band_px[b]);
    m.band_rgb[b] = nullptr;
    m.band_px[b] = nullptr;
    m.sent_valid[b] = false;
  }
  m.cap_pixels = 0;
  for (int b = 0; b < 2; ++b) {
    RTK_HIP(hipMalloc((void**)&m.band_rgb[b], pixels * 3 * sizeof(float)));
    RTK_HIP(hipMalloc((void**)&m.band_px[b], pixels * 4));
  }
  m.cap_pixels = pixels;
  return RTK_OK;
}

}  // namespace

extern "C" {

int rtk_group_create(const int* device_ids, int n_members, rtk_group** out) {
  if (!out) return fail(RTK_ERR_INVALID, "rtk_group_create: null out pointer");
  *out = nullptr;
  if (!device_ids || n_members <= 0 || n_members > 1024) return fail(RTK_ERR_INVALID, "rtk_group_create: bad device list");
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

int rtk_group_upload(rtk_group* g, const rtk_scene* scene) {
  if (!g || !scene) return fail(RTK_ERR_INVALID, "rtk_group_upload: null argument");
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
  const size_t n = g->members.size();
  RTK_HIP(hipSetDevice(root.ctx->device));
  hipStream_t st0 = hip_stream ? (hipStream_t)hip_stream : root.ctx->stream;
  g->last_stream = st0;
  if (n == 1) {
    g->exchanged = false;
    ++g->frame;
    return rtk_render_device(root.ctx, root.scene, camera, nullptr, d_rgb_f32, d_rgba8, st0);
  }
  const uint32_t width = camera->width, height = camera->height;
  const uint32_t m_rows = band_tile_rows_for(g, height);
  const uint32_t tiles_y = (height + RTK_TILE_H - 1) / RTK_TILE_H;
  const uint32_t n_bands = (tiles_y + m_rows - 1) / m_rows;
  const int b = (int)(g->frame & 1u);
  const bool exchange = d_rgb_f32 != nullptr || d_rgba8 != nullptr;
  int rc;

  // every other member: its bands into its own buffer, on its own stream
  for (size_t r = 1; r < n; ++r) {
    Member& m = g->members[r];
    rtk_banding bd = {(uint32_t)r, (uint32_t)n, m_rows, 0u, 0u, 1u};
    const size_t pixels = (size_t)rtk_band_pixel_rows(height, &bd) * width;
    if (exchange && (rc = ensure_band_buffers(m, pixels)) != RTK_OK) return rc;
    RTK_HIP(hipSetDevice(m.ctx->device));
    // this buffer's rows of two frames ago must have left before they are overwritten
    if (m.sent_valid[b]) RTK_HIP(hipStreamWaitEvent(m.ctx->stream, m.ev_sent[b], 0));
    if ((rc = rtk_render_device_bands(m.ctx, m.scene, camera, &bd, d_rgb_f32 ? m.band_rgb[b] : nullptr,
                                      d_rgba8 ? m.band_px[b] : nullptr, nullptr)) != RTK_OK)
      return rc;
    if (exchange) {
      RTK_HIP(hipEventRecord(m.ev_rendered, m.ctx->stream));
      RTK_HIP(hipStreamWaitEvent(g->slots[(size_t)m.slot].comm_stream, m.ev_rendered, 0));
    }
  }

  // the root: what the caller's stream did with the frame buffers before this call comes first,
  // then its own bands (on the caller's stream) and the incoming bands (on the communication stream)
  // fill disjoint rows of the frame side by side
  DeviceSlot& slot0 = g->slots[(size_t)root.slot];
  RTK_HIP(hipSetDevice(root.ctx->device));
  if (exchange) {
    RTK_HIP(hipEventRecord(g->ev_enter, st0));
    RTK_HIP(hipStreamWaitEvent(slot0.comm_stream, g->ev_enter, 0));
  }
  {
    rtk_banding bd = {0u, (uint32_t)n, m_rows, 0u, 0u, 0u};
    if ((rc = rtk_render_device_bands(root.ctx, root.scene, camera, &bd, d_rgb_f32, d_rgba8, st0)) != RTK_OK) return rc;
  }
  g->exchanged = exchange;
  if (exchange) {
    RTK_HIP(hipEventRecord(g->ev_x0, slot0.comm_stream));
    // Visits every band of every other member: where it lies in the member's buffer (`from`), where
    // its rows go in the frame (`to`), both in pixels.
    auto for_each_band = [&](auto&& visit) {
      for (size_t r = 1; r < n; ++r) {
        size_t local_row = 0;  // pixel rows of this member's buffer consumed so far
        for (uint32_t band = (uint32_t)r; band < n_bands; band += (uint32_t)n) {
          const uint32_t t0 = band * m_rows, t1 = std::min(tiles_y, t0 + m_rows);
          const size_t y0 = (size_t)t0 * RTK_TILE_H, y1 = std::min<size_t>(height, (size_t)t1 * RTK_TILE_H);
          if (!visit(g->members[r], (y1 - y0) * width, local_row * width, y0 * width)) return false;
          local_row += (size_t)(t1 - t0) * RTK_TILE_H;
        }
      }
      return true;
    };
    if (g->use_rccl) {
      RTK_RCCL(g_rccl.GroupStart());
      ncclResult_t first_bad = ncclSuccess;
      for_each_band([&](Member& m, size_t count, size_t from, size_t to) {
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
        return first_bad == ncclSuccess;
      });
      const ncclResult_t end = g_rccl.GroupEnd();  // always closes the group, even after a failed call inside it
      if (first_bad != ncclSuccess) return fail_rccl(first_bad, "ncclSend / ncclRecv");
      if (end != ncclSuccess) return fail_rccl(end, "ncclGroupEnd");
    } else {
      // each member pushes its bands over its own link, on its own communication stream, once the
      // caller's stream has let go of the frame buffers
      for (size_t r = 1; r < n; ++r) {
        Member& m = g->members[r];
        RTK_HIP(hipSetDevice(m.ctx->device));
        RTK_HIP(hipStreamWaitEvent(g->slots[(size_t)m.slot].comm_stream, g->ev_enter, 0));
      }
      hipError_t first_bad = hipSuccess;
      const int root_device = root.ctx->device;
      for_each_band([&](Member& m, size_t count, size_t from, size_t to) {
        hipStream_t cs = g->slots[(size_t)m.slot].comm_stream;
        first_bad = hipSetDevice(m.ctx->device);
        if (d_rgb_f32 && first_bad == hipSuccess)
          first_bad = hipMemcpyPeerAsync(d_rgb_f32 + to * 3, root_device, m.band_rgb[b] + from * 3, m.ctx->device,
                                         count * 3 * sizeof(float), cs);
        if (d_rgba8 && first_bad == hipSuccess)
          first_bad = hipMemcpyPeerAsync(d_rgba8 + to * 4, root_device, m.band_px[b] + from * 4, m.ctx->device, count * 4, cs);
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

int rtk_group_get_stats(rtk_group* g, rtk_stats* stats) {
  if (!g || !stats) return fail(RTK_ERR_INVALID, "rtk_group_get_stats: null argument");
  std::memset(stats, 0, sizeof *stats);
  if (g->frame == 0) return RTK_OK;
  for (Member& m : g->members) {
    rtk_stats s;
    const int rc = rtk_get_stats(m.ctx, &s);  // waits for the member's last launch
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

}  // extern "C"
