// rtk_api.hip -- the device-facing half of the C ABI of include/rtk.h: contexts, HBM residency of
// flattened scenes, kernel launches, timing, copies to the host.  Host code only; the kernels live in
// rtk_kernels.hip, rtk_opaque.hip and rtk_wavefront.hip, and everything that needs no device
// (validation, flattening, camera and partition arithmetic) in rtk_host.cpp.
//
// There is no CPU rendering path in this library: every entry point that produces pixels
// launches the gfx950 kernels and fails with an error code when no device is usable.

#include "rtk_internal.h"

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <functional>
#include <utility>
#include <vector>

#include <pthread.h>

namespace rtk {
int fail_hip(hipError_t e, const char* what) { return fail(RTK_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e)); }
}  // namespace rtk

using rtk::fail;
using rtk::fail_hip;
using rtk::tiles_across;
using rtk::tiles_down;

// ---------------------------------------------------------------- static scratch (rtk_render_device_static)
// Device memory that exists without any allocation call: one array per GPU inside the library's
// code object.  Layout (offsets in bytes), sized by the limits in include/rtk.h:
//   counters      kCounterSlots x DevCounters (the shards and the work counter behind them)
//   prim          RTK_STATIC_MAX_SPHERES x float4     per-camera records of the per-ray filter
//   prim_ball     RTK_STATIC_MAX_SPHERES x float4     ... of the wavefront-level cone test
//   prim_cluster  RTK_STATIC_MAX_SPHERES / 64 x float4  ... of the cluster balls
//   hit_t         RTK_STATIC_MAX_RECORD_PIXELS x f64  hit records of the three-kernel path
//   hit_obj       RTK_STATIC_MAX_RECORD_PIXELS x u32
//   shadow        RTK_STATIC_MAX_RECORD_PIXELS x u16
namespace {
constexpr size_t kStaticCounters = 0;
constexpr size_t kStaticPrim = (kStaticCounters + sizeof(rtk::DevCounters) * rtk::kCounterSlots + 255u) & ~(size_t)255u;
constexpr size_t kStaticPrimBall = kStaticPrim + (size_t)RTK_STATIC_MAX_SPHERES * sizeof(float4);
constexpr size_t kStaticPrimCluster = kStaticPrimBall + (size_t)RTK_STATIC_MAX_SPHERES * sizeof(float4);
constexpr size_t kStaticHitT = kStaticPrimCluster + (size_t)RTK_STATIC_MAX_SPHERES / rtk::kBundle * sizeof(float4);
constexpr size_t kStaticHitObj = kStaticHitT + (size_t)RTK_STATIC_MAX_RECORD_PIXELS * sizeof(double);
constexpr size_t kStaticShadow = kStaticHitObj + (size_t)RTK_STATIC_MAX_RECORD_PIXELS * sizeof(uint32_t);
constexpr size_t kStaticEnd = kStaticShadow + (size_t)RTK_STATIC_MAX_RECORD_PIXELS * sizeof(uint16_t);
static_assert(kStaticEnd <= RTK_STATIC_SCRATCH_BYTES, "static scratch layout exceeds RTK_STATIC_SCRATCH_BYTES");
static_assert(kStaticPrim % 256 == 0 && kStaticHitT % 256 == 0 && kStaticHitObj % 256 == 0 && kStaticShadow % 256 == 0,
              "pieces of the static scratch are 256-byte aligned");
static_assert(RTK_STATIC_MAX_SPHERES % rtk::kBundle == 0, "the sphere limit is a whole number of clusters");
}  // namespace

static __device__ __align__(256) unsigned char scratch[RTK_STATIC_SCRATCH_BYTES];

namespace {
// In host code the array's name is only a handle for the runtime: each GPU holds its own copy at
// its own address, looked up once per device with that device current (rtk_create).
constexpr int kMaxDevices = 64;
unsigned char* g_static_scratch[kMaxDevices] = {};
std::mutex g_static_mutex;

hipError_t static_scratch_for_current_device(int device, unsigned char** out) {
  std::lock_guard<std::mutex> lock(g_static_mutex);
  if (!g_static_scratch[device]) {
    void* p = nullptr;
    const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess) return e;
    if (!p) return hipErrorInvalidSymbol;
    g_static_scratch[device] = (unsigned char*)p;
  }
  *out = g_static_scratch[device];
  return hipSuccess;
}
}  // namespace


extern "C" {


static hipError_t create_timing_events(rtk_context* ctx) {
  hipError_t e;
  for (int i = 0; i < RTK_TIMING_HISTORY; ++i) {
    if ((e = hipEventCreate(&ctx->ev_start[i])) != hipSuccess || (e = hipEventCreate(&ctx->ev_rays[i])) != hipSuccess ||
        (e = hipEventCreate(&ctx->ev_stop[i])) != hipSuccess)
      return e;
  }
  for (int i = 0; i < rtk_context::kHostBlocks; ++i)
    if ((e = hipEventCreateWithFlags(&ctx->ev_block[i], hipEventDisableTiming)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&ctx->ev_landed[i], hipEventDisableTiming)) != hipSuccess)
      return e;
  return hipSuccess;
}

int rtk_create(int device_id, rtk_context** out) {
  if (!out) return fail(RTK_ERR_INVALID, "rtk_create: null out pointer");
  *out = nullptr;
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0) return fail(RTK_ERR_NO_DEVICE, "rtk_create: no HIP device available");
  if (device_id < 0 || device_id >= count || device_id >= kMaxDevices)
    return fail(RTK_ERR_NO_DEVICE, "rtk_create: device id out of range");
  hipDeviceProp_t prop;
  RTK_HIP(hipGetDeviceProperties(&prop, device_id));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(RTK_ERR_NO_DEVICE, std::string("rtk_create: kernels are built for gfx950, device is ") + prop.gcnArchName);
  RTK_HIP(hipSetDevice(device_id));
  rtk_context* ctx = new (std::nothrow) rtk_context();
  if (!ctx) return fail(RTK_ERR_NOMEM, "rtk_create: out of host memory");
  ctx->device = device_id;
  // Queue room of the level-by-level path: 128 bytes per node, up to 8 GB of the device's 288 unless
  // RTK_WAVEFRONT_NODES says otherwise (0: scenes with secondary rays stay on the depth-first kernel).
  ctx->wavefront_max_nodes = 64u << 20;
  if (const char* v = std::getenv("RTK_WAVEFRONT_NODES")) {
    char* end = nullptr;
    const unsigned long long n = std::strtoull(v, &end, 10);
    if (end != v && *end == '\0') ctx->wavefront_max_nodes = (uint32_t)std::min<unsigned long long>(n, 1ull << 31);
  }
  if ((e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess ||
      (e = hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking)) != hipSuccess ||
      (e = create_timing_events(ctx)) != hipSuccess ||
      (e = static_scratch_for_current_device(device_id, &ctx->static_scratch)) != hipSuccess ||
      (e = hipMalloc((void**)&ctx->d_counters, sizeof(rtk::DevCounters) * rtk::kCounterSlots)) != hipSuccess ||
      (e = hipMemset(ctx->d_counters, 0, sizeof(rtk::DevCounters) * rtk::kCounterSlots)) != hipSuccess) {
    rtk_destroy(ctx);
    return fail_hip(e, "rtk_create");
  }
  *out = ctx;
  return RTK_OK;
}

int rtk_set_trace_order(rtk_context* ctx, uint32_t order, uint64_t max_queue_nodes) {
  if (!ctx) return fail(RTK_ERR_INVALID, "rtk_set_trace_order: null context");
  if (order > RTK_TRACE_BY_LEVEL) return fail(RTK_ERR_INVALID, "rtk_set_trace_order: unknown order");
  if (max_queue_nodes > (1ull << 31)) return fail(RTK_ERR_INVALID, "rtk_set_trace_order: more than 2^31 queue nodes");
  ctx->trace_order = order;
  if (max_queue_nodes) {
    RTK_HIP(hipSetDevice(ctx->device));
    RTK_HIP(hipDeviceSynchronize());  // frames in flight may use the queue
    rtk::WavefrontBuffers& w = ctx->wavefront;
    if (w.rays) (void)hipFree(w.rays);
    if (w.outs) (void)hipFree(w.outs);
    w.rays = nullptr, w.outs = nullptr, w.cuts = nullptr, w.cap_nodes = 0;
    w.nodes_per_pixel = rtk::kWavefrontNodesPerPixel;
    ctx->wavefront_max_nodes = (uint32_t)max_queue_nodes;
  }
  return RTK_OK;
}

void rtk_destroy(rtk_context* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  if (ctx->copy_stream) (void)hipStreamSynchronize(ctx->copy_stream);
  if (ctx->d_rgb) (void)hipFree(ctx->d_rgb);
  if (ctx->d_rgba8) (void)hipFree(ctx->d_rgba8);
  if (ctx->d_counters) (void)hipFree(ctx->d_counters);
  if (ctx->opaque.hit_obj) (void)hipFree(ctx->opaque.hit_obj);
  if (ctx->opaque.hit_t) (void)hipFree(ctx->opaque.hit_t);
  if (ctx->opaque.shadow) (void)hipFree(ctx->opaque.shadow);
  if (ctx->d_samples) (void)hipFree(ctx->d_samples);
  if (ctx->wavefront.rays) (void)hipFree(ctx->wavefront.rays);
  if (ctx->wavefront.outs) (void)hipFree(ctx->wavefront.outs);
  if (ctx->wavefront.ctrl) (void)hipFree(ctx->wavefront.ctrl);
  if (ctx->wavefront.counters) (void)hipFree(ctx->wavefront.counters);
  if (ctx->wavefront.host_overflow) (void)hipHostFree(ctx->wavefront.host_overflow);
  if (ctx->cached_handle) rtk_scene_release(ctx, ctx->cached_handle);
  if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
  for (int i = 0; i < RTK_TIMING_HISTORY; ++i) {
    if (ctx->ev_start[i]) (void)hipEventDestroy(ctx->ev_start[i]);
    if (ctx->ev_rays[i]) (void)hipEventDestroy(ctx->ev_rays[i]);
    if (ctx->ev_stop[i]) (void)hipEventDestroy(ctx->ev_stop[i]);
  }
  for (hipEvent_t ev : ctx->ev_block)
    if (ev) (void)hipEventDestroy(ev);
  for (hipEvent_t ev : ctx->ev_landed)
    if (ev) (void)hipEventDestroy(ev);
  if (ctx->h_bounce) (void)hipHostFree(ctx->h_bounce);
  if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

void rtk_scene_release(rtk_context* ctx, rtk_scene_handle* h) {
  if (!h) return;
  if (ctx) (void)hipSetDevice(ctx->device);
  if (h->d_block) (void)hipFree(h->d_block);
  delete h;
}


// Page-locked staging of a context, grown on demand: uploads copy from it at full PCIe rate and no
// fresh pages are faulted in per scene.  Null when the allocation fails (the error text is set).
static unsigned char* context_staging(void* user, size_t bytes) {
  rtk_context* ctx = static_cast<rtk_context*>(user);
  if (bytes > ctx->cap_stage) {
    if (hipSetDevice(ctx->device) != hipSuccess) return nullptr;
    if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
    ctx->h_stage = nullptr;
    ctx->cap_stage = 0;
    const size_t want = bytes + bytes / 4;
    if (hipHostMalloc((void**)&ctx->h_stage, want, hipHostMallocDefault) != hipSuccess) {
      ctx->h_stage = nullptr;
      return nullptr;
    }
    ctx->cap_stage = want;
  }
  return ctx->h_stage;
}

int rtk_scene_upload(rtk_context* ctx, const rtk_scene* scene, rtk_scene_handle** out) {
  if (!ctx || !scene || !out) return fail(RTK_ERR_INVALID, "rtk_scene_upload: null argument");
  *out = nullptr;
  rtk::FlatScene flat;
  flat.grow = context_staging;
  flat.user = ctx;
  const int rc = rtk::flatten_scene(scene, flat);
  if (rc != RTK_OK) return rc;
  return rtk::place_scene(ctx, flat, out);
}

}  // extern "C"


int rtk::place_scene(rtk_context* ctx, const rtk::FlatScene& flat, rtk_scene_handle** out, rtk_scene_handle* reuse) {
  *out = nullptr;
  rtk_scene_handle* h = nullptr;
  if (reuse && reuse->d_block && reuse->block_bytes >= flat.total) {
    // the old scene's allocation is large enough: no hipFree / hipMalloc pair (a quarter of a millisecond)
    void* const block = reuse->d_block;
    const size_t bytes = reuse->block_bytes;
    *reuse = rtk_scene_handle{};
    reuse->d_block = block, reuse->block_bytes = bytes;
    h = reuse;
  } else {
    if (reuse) rtk_scene_release(ctx, reuse);
    h = new (std::nothrow) rtk_scene_handle();
    if (!h) return fail(RTK_ERR_NOMEM, "rtk_scene_upload: out of host memory");
  }
  hipError_t e = hipSetDevice(ctx->device);
  if (e == hipSuccess && flat.total) {
    if (!h->d_block) {
      e = hipMalloc(&h->d_block, flat.total);
      if (e == hipSuccess) h->block_bytes = flat.total;
    }
    if (e == hipSuccess && flat.staged_bytes)
      e = hipMemcpy(h->d_block, flat.staged, flat.staged_bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess && !flat.extra.empty())
      e = hipMemcpy((unsigned char*)h->d_block + flat.staged_bytes, flat.extra.data(), flat.extra.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess)
      for (const rtk::FlatScene::Piece& p : flat.pieces) h->*(p.dst) = (unsigned char*)h->d_block + p.offset;
  }
  if (e != hipSuccess) {
    rtk_scene_release(ctx, h);
    return fail_hip(e, "rtk_scene_upload");
  }
  rtk::DevScene& d = h->dev;
  d = flat.dev;
  d.objects = (decltype(d.objects))h->d_objects;
  d.lights = (decltype(d.lights))h->d_lights;
  d.sph_filter = (decltype(d.sph_filter))h->d_filter;
  d.sph_exact = (decltype(d.sph_exact))h->d_exact;
  d.sph_object = (decltype(d.sph_object))h->d_sph_object;
  d.sph_record = (decltype(d.sph_record))h->d_sph_record;
  d.plane_object = (decltype(d.plane_object))h->d_plane_object;
  d.plane_light = (decltype(d.plane_light))h->d_plane_light;
  d.pattern_inv = (decltype(d.pattern_inv))h->d_pattern_inv;
  d.pattern_chain = (decltype(d.pattern_chain))h->d_pattern_chain;
  d.light_filter = (decltype(d.light_filter))h->d_light_filter;
  d.light_kind = (decltype(d.light_kind))h->d_light_kind;
  d.k_table = (decltype(d.k_table))h->d_k_table;
  d.sph_tab = (decltype(d.sph_tab))h->d_sph_tab;
  d.sph_ball = (decltype(d.sph_ball))h->d_sph_ball;
  d.cluster_ball = (decltype(d.cluster_ball))h->d_cluster_ball;
  d.super_ball = (decltype(d.super_ball))h->d_super_ball;
  d.cluster_plain = (decltype(d.cluster_plain))h->d_cluster_plain;
  d.light_grid = (decltype(d.light_grid))h->d_light_grid;
  d.grid_cells = (decltype(d.grid_cells))h->d_grid_cells;
  d.grid_entries = (decltype(d.grid_entries))h->d_grid_entries;
  d.grid_filter = (decltype(d.grid_filter))h->d_grid_filter;
  d.tg_cells = (decltype(d.tg_cells))h->d_tg_cells;
  d.tg_filter = (decltype(d.tg_filter))h->d_tg_filter;
  d.tg_slot = (decltype(d.tg_slot))h->d_tg_slot;
  *out = h;
  return RTK_OK;
}


// ---------------------------------------------------------------- one frame on one device
// What follows the render kernels inside the timed region of a sampled call: the sample frame
// d_rgb_f32 of the (supersampled) camera is averaged into the caller's frame.
struct ResolveStep {
  uint32_t s = 1;  // samples per axis; 1 = nothing to do
  uint32_t width = 0, height = 0;
  float* d_rgb = nullptr;
  uint8_t* d_rgba8 = nullptr;
  bool tiles = false;  // the sample frame is a rank's compact tile buffer
  uint64_t n_tiles = 0;
};

// Device memory a frame's kernels use beyond the scene, the camera and the outputs.
struct Scratch {
  float4 *prim = nullptr, *prim_ball = nullptr, *prim_cluster = nullptr;  // per-camera records
  rtk::DevCounters* counters = nullptr;
  rtk::OpaqueBuffers opaque{};  // hit records of the three-kernel path (null when not needed)
  const rtk::WavefrontBuffers* wavefront = nullptr;  // node queue of the level-by-level path (null: depth-first kernel)
};

// A frame in the making: frame_begin, one or more frame_launch, frame_end.
struct Frame {
  rtk_context* ctx = nullptr;
  rtk_scene_handle* scene = nullptr;
  const rtk_camera* camera = nullptr;
  rtk::DevCamera cam{};
  hipStream_t st = nullptr;
  int slot = 0;
  bool opaque = false;
  bool capturing = false;  // the stream is recording a HIP graph: no timing events, no bookkeeping of "the last call"
  Scratch scr;
};

// Records of the three-kernel path for launches of up to `pixels` pixels, in the context's own
// buffers (grown on demand).
static int context_records(rtk_context* ctx, size_t pixels, rtk::OpaqueBuffers& out) {
  if (pixels > ctx->cap_opaque) {
    if (ctx->opaque.hit_obj) (void)hipFree(ctx->opaque.hit_obj);
    if (ctx->opaque.hit_t) (void)hipFree(ctx->opaque.hit_t);
    if (ctx->opaque.shadow) (void)hipFree(ctx->opaque.shadow);
    ctx->opaque = rtk::OpaqueBuffers{};
    ctx->cap_opaque = 0;
    RTK_HIP(hipMalloc((void**)&ctx->opaque.hit_obj, pixels * sizeof(uint32_t)));
    RTK_HIP(hipMalloc((void**)&ctx->opaque.hit_t, pixels * sizeof(double)));
    RTK_HIP(hipMalloc((void**)&ctx->opaque.shadow, pixels * sizeof(uint16_t)));
    ctx->cap_opaque = pixels;
  }
  out = ctx->opaque;
  return RTK_OK;
}

// Node queue of the level-by-level path for launches of up to `pixels` pixels, in the context's own
// buffers: room for kWavefrontNodesPerPixel nodes per pixel, capped (larger launches run in passes).
// Null when device memory cannot be had: the launch then runs on the depth-first kernel.
static const rtk::WavefrontBuffers* context_wavefront(rtk_context* ctx, size_t pixels) {
  rtk::WavefrontBuffers& w = ctx->wavefront;
  if (!w.host_overflow) {
    if (hipHostMalloc((void**)&w.host_overflow, sizeof(unsigned int), hipHostMallocDefault) != hipSuccess) {
      (void)hipGetLastError();
      w.host_overflow = nullptr;
      return nullptr;
    }
    *w.host_overflow = 0u;
  }
  // an earlier launch ran out of room (and was traced again depth first): this scene's trees are
  // larger than allowed for, so reserve twice the room per pixel from now on
  if (*static_cast<volatile unsigned int*>(w.host_overflow)) {
    *w.host_overflow = 0u;
    w.nodes_per_pixel = std::min(w.nodes_per_pixel * 2u, (1u << rtk::kMaxDepth) - 1u);
  }
  const size_t floor_nodes = (size_t)rtk::kBlock * w.nodes_per_pixel;
  size_t want = std::min<size_t>(std::max<size_t>(pixels * w.nodes_per_pixel, floor_nodes), ctx->wavefront_max_nodes);
  if (want < floor_nodes) return nullptr;
  // the queue is kept while it fits the launch and is not more than eight times too large for it (a
  // context that went from 4K frames to thumbnails gives the memory back)
  if (w.cap_nodes >= want && w.cap_nodes / 8u <= want) return &w;
  // never more than an eighth of the device memory that is free now: larger launches run in passes
  // (104 bytes per node: a 1080p launch asks for 1.3 GB, a 4K launch for 5.2 GB, the cap is 6.5 GB)
  size_t free_bytes = 0, total_bytes = 0;
  if (hipMemGetInfo(&free_bytes, &total_bytes) == hipSuccess) {
    const size_t held = (size_t)w.cap_nodes * rtk::kWavefrontNodeBytes;
    want = std::min(want, (free_bytes + held) / 8u / rtk::kWavefrontNodeBytes);
    if (want < floor_nodes) return nullptr;
    if (w.cap_nodes >= want && w.cap_nodes / 8u <= want) return &w;
  } else {
    (void)hipGetLastError();
  }
  // an earlier frame's kernels may still use the old buffers
  if (hipDeviceSynchronize() != hipSuccess) return nullptr;
  if (w.rays) (void)hipFree(w.rays);
  if (w.outs) (void)hipFree(w.outs);
  w.rays = nullptr, w.outs = nullptr, w.cuts = nullptr, w.cap_nodes = 0;
  if (!w.ctrl && hipMalloc((void**)&w.ctrl, sizeof(unsigned int) * rtk::kWavefrontCtrlWords) != hipSuccess) w.ctrl = nullptr;
  if (!w.counters && hipMalloc((void**)&w.counters, sizeof(rtk::DevCounters) * rtk::kCounterShards) != hipSuccess) w.counters = nullptr;
  if (!w.ctrl || !w.counters || hipMalloc((void**)&w.rays, (want + 63u) / 64u * rtk::kRayBlockBytes) != hipSuccess ||
      hipMalloc((void**)&w.outs, want * (sizeof(rtk::WfNodeOut) + sizeof(rtk::WfNodeCut))) != hipSuccess) {
    (void)hipGetLastError();
    if (w.rays) (void)hipFree(w.rays);
    w.rays = nullptr, w.outs = nullptr, w.cuts = nullptr;
    return nullptr;
  }
  w.cuts = reinterpret_cast<rtk::WfNodeCut*>(w.outs + want);
  w.cap_nodes = (uint32_t)want;
  return &w;
}

// Orders this call behind the context's previous one (they share per-camera records, counters
// and hit records), resets the counters, rebuilds the per-camera records and starts the timing.
static int frame_begin(Frame& f, rtk_context* ctx, rtk_scene_handle* scene, const rtk_camera* camera, void* hip_stream,
                       const Scratch& scr) {
  f.ctx = ctx, f.scene = scene, f.camera = camera, f.scr = scr;
  const int rc = rtk::make_dev_camera(camera, f.cam);
  if (rc != RTK_OK) return rc;
  RTK_HIP(hipSetDevice(ctx->device));
  f.st = hip_stream ? (hipStream_t)hip_stream : ctx->stream;
  f.opaque = !scene->dev.any_secondary;
  // A stream that is recording a graph takes the kernels as graph nodes.  An event recorded there
  // cannot be waited on or read afterwards, and an event of work outside the graph cannot be waited on
  // from inside it: such a call records no timing events and does not become "the context's last call"
  // (rtk_get_stats keeps describing the last call made outside a capture); ordering the graph's launches
  // against the context's other frames, which share the per-camera records and counters, is the caller's.
  hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
  RTK_HIP(hipStreamIsCapturing(f.st, &capture));
  f.capturing = capture != hipStreamCaptureStatusNone;
  if (!f.capturing && ctx->n_timed && ctx->last_stream != f.st)
    RTK_HIP(hipStreamWaitEvent(f.st, ctx->ev_stop[(ctx->n_timed - 1) % RTK_TIMING_HISTORY], 0));
  RTK_HIP(rtk::launch_prep_primary(f.st, scene->dev, f.cam, f.scr.prim, f.scr.prim_ball, f.scr.prim_cluster, f.scr.counters));
  f.slot = (int)(ctx->n_timed % RTK_TIMING_HISTORY);
  if (f.capturing) return RTK_OK;
  RTK_HIP(hipEventRecord(ctx->ev_start[f.slot], f.st));  // the per-camera record kernel (microseconds) stays outside
  ctx->ev_split[f.slot] = false;
  return RTK_OK;
}

// Renders the tiles of one partition.  `split`: record the end-of-ray-kernels event (a frame
// made of one launch only).
static int frame_launch(Frame& f, const rtk::Partition& p, float* d_rgb_f32, uint8_t* d_rgba8, bool split) {
  // per-launch pixel indices (local tile * 256 + slot) are 32-bit
  if (p.n_local_tiles > 0x00ffffffull)
    return fail(RTK_ERR_UNSUPPORTED, "rtk_render_device: more than 16,777,215 tiles (4.29e9 pixels) per launch");
  rtk::RenderTarget t{};
  t.rgb = d_rgb_f32;
  t.rgba8 = d_rgba8;
  t.tiles_x = tiles_across(f.camera->width);
  t.tiles_y = tiles_down(f.camera->height);
  t.n_local_tiles = (uint32_t)p.n_local_tiles;
  t.layout = p.layout;
  t.step = p.step, t.offset = p.offset;
  t.band_rows = p.band_rows, t.band_period = p.band_period, t.band_offset = p.band_offset, t.row0 = p.row0;
  rtk_context* ctx = f.ctx;
  if (f.capturing) split = false;
  // rays and shading in one kernel: its end is the frame's end, and a third event per frame is a packet the
  // stream can do without
  if (f.opaque && f.scene->dev.max_depth > 0 && (f.scene->dev.lights_uniform || f.scene->dev.lights_bundled)) split = false;
  if (f.opaque) {
    if (split) ctx->ev_split[f.slot] = true;
    RTK_HIP(rtk::launch_render_opaque(f.st, f.scene->dev, f.cam, f.scr.prim, f.scr.prim_ball, f.scr.prim_cluster, t,
                                      f.scr.opaque, f.scr.counters, split ? ctx->ev_rays[f.slot] : nullptr));
  } else {
    RTK_HIP(rtk::launch_render(f.st, f.scene->dev, f.cam, t, f.scr.counters, f.scr.wavefront, f.scr.prim, f.scr.prim_ball, f.scr.prim_cluster));
  }
  return RTK_OK;
}

static int frame_end(Frame& f, const ResolveStep& resolve, const float* d_samples) {
  rtk_context* ctx = f.ctx;
  if (resolve.s > 1 && resolve.tiles)
    RTK_HIP(rtk::launch_resolve_tiles(f.st, resolve.n_tiles, resolve.s, d_samples, resolve.d_rgb, resolve.d_rgba8));
  else if (resolve.s > 1)
    RTK_HIP(rtk::launch_resolve(f.st, resolve.width, resolve.height, resolve.s, d_samples, resolve.d_rgb, resolve.d_rgba8));
  if (f.capturing) return RTK_OK;
  RTK_HIP(hipEventRecord(ctx->ev_stop[f.slot], f.st));
  ctx->last_stream = f.st;
  ctx->last_counters = f.scr.counters;
  ++ctx->n_timed;
  return RTK_OK;
}

// Scratch in memory of the scene handle (per-camera records) and the context (counters, hit records).
static int context_scratch(rtk_context* ctx, rtk_scene_handle* scene, size_t launch_pixels, Scratch& scr) {
  scr.prim = (float4*)scene->d_prim;
  scr.prim_ball = (float4*)scene->d_prim_ball;
  scr.prim_cluster = (float4*)scene->d_prim_cluster;
  scr.counters = ctx->d_counters;
  scr.opaque = rtk::OpaqueBuffers{};
  if (!scene->dev.any_secondary && rtk::opaque_needs_records(scene->dev)) {
    RTK_HIP(hipSetDevice(ctx->device));
    return context_records(ctx, launch_pixels, scr.opaque);
  }
  const rtk::DevScene& ds = scene->dev;
  // (scenes of a handful of objects go to a kernel of their own before the queue is ever looked at)
  const bool by_level = rtk::wavefront_applies(ds) &&
                        (ctx->trace_order == RTK_TRACE_BY_LEVEL ||
                         (ctx->trace_order == RTK_TRACE_AUTO && rtk::wavefront_pays(ds, launch_pixels)));
  if (by_level && ds.any_secondary && ds.max_depth >= 2 && ctx->wavefront_max_nodes) {
    RTK_HIP(hipSetDevice(ctx->device));
    scr.wavefront = context_wavefront(ctx, launch_pixels);
  }
  return RTK_OK;
}

static int render_device(rtk_context* ctx, rtk_scene_handle* scene, const rtk_camera* camera, const rtk::Partition& p,
                         float* d_rgb_f32, uint8_t* d_rgba8, void* hip_stream, const ResolveStep& resolve) {
  if (camera->width == 0 || camera->height == 0) return fail(RTK_ERR_INVALID, "camera has zero size");
  Scratch scr;
  int rc = context_scratch(ctx, scene, (size_t)p.n_local_tiles * 256u, scr);
  if (rc != RTK_OK) return rc;
  Frame f;
  if ((rc = frame_begin(f, ctx, scene, camera, hip_stream, scr)) != RTK_OK) return rc;
  if ((rc = frame_launch(f, p, d_rgb_f32, d_rgba8, true)) != RTK_OK) return rc;
  ResolveStep step = resolve;
  step.n_tiles = p.n_local_tiles;
  return frame_end(f, step, d_rgb_f32);
}

int rtk::render_device_partition(rtk_context* ctx, rtk_scene_handle* scene, const rtk_camera* camera, const rtk::Partition& p,
                                 float* d_rgb_f32, uint8_t* d_rgba8, void* hip_stream) {
  return render_device(ctx, scene, camera, p, d_rgb_f32, d_rgba8, hip_stream, ResolveStep{});
}

extern "C" {

int rtk_render_device(rtk_context* ctx, rtk_scene_handle* scene, const rtk_camera* camera, const rtk_tiling* tiling,
                      float* d_rgb_f32, uint8_t* d_rgba8, void* hip_stream) {
  if (!ctx || !scene || !camera) return fail(RTK_ERR_INVALID, "rtk_render_device: null argument");
  rtk::Partition p;
  const int rc = rtk::partition_from_tiling(camera, tiling, p, "rtk_render_device");
  if (rc != RTK_OK) return rc;
  return render_device(ctx, scene, camera, p, d_rgb_f32, d_rgba8, hip_stream, ResolveStep{});
}

int rtk_render_device_bands(rtk_context* ctx, rtk_scene_handle* scene, const rtk_camera* camera, const rtk_banding* banding,
                            float* d_rgb_f32, uint8_t* d_rgba8, void* hip_stream) {
  if (!ctx || !scene || !camera || !banding) return fail(RTK_ERR_INVALID, "rtk_render_device_bands: null argument");
  rtk::Partition p;
  const int rc = rtk::partition_from_banding(camera, banding, p, "rtk_render_device_bands");
  if (rc != RTK_OK) return rc;
  return render_device(ctx, scene, camera, p, d_rgb_f32, d_rgba8, hip_stream, ResolveStep{});
}

int rtk_render_device_static(rtk_context* ctx, rtk_scene_handle* scene, const rtk_camera* camera, const rtk_tiling* tiling,
                             float* d_rgb_f32, uint8_t* d_rgba8, void* hip_stream) {
  const char* who = "rtk_render_device_static";
  if (!ctx || !scene || !camera) return fail(RTK_ERR_INVALID, std::string(who) + ": null argument");
  if (camera->width == 0 || camera->height == 0) return fail(RTK_ERR_INVALID, "camera has zero size");
  rtk::Partition p;
  int rc = rtk::partition_from_tiling(camera, tiling, p, who);
  if (rc != RTK_OK) return rc;
  const rtk::DevScene& S = scene->dev;
  if (S.n_spheres_padded > RTK_STATIC_MAX_SPHERES)
    return fail(RTK_ERR_UNSUPPORTED, std::string(who) + ": the scene has more than " + std::to_string(RTK_STATIC_MAX_SPHERES) +
                                         " spheres (the limit of the static scratch); use rtk_render_device");
  const bool records = !S.any_secondary && rtk::opaque_needs_records(S);
  if (records && p.n_local_tiles * 256u > RTK_STATIC_MAX_RECORD_PIXELS)
    return fail(RTK_ERR_UNSUPPORTED,
                std::string(who) + ": this scene keeps a hit record per pixel and the call covers more than " +
                    std::to_string(RTK_STATIC_MAX_RECORD_PIXELS) + " pixels; split the frame with `tiling` or use rtk_render_device");
  unsigned char* base = ctx->static_scratch;  // this device's copy of the array, looked up in rtk_create
  Scratch scr;
  scr.counters = (rtk::DevCounters*)(base + kStaticCounters);
  scr.prim = (float4*)(base + kStaticPrim);
  scr.prim_ball = (float4*)(base + kStaticPrimBall);
  scr.prim_cluster = (float4*)(base + kStaticPrimCluster);
  if (records) {
    scr.opaque.hit_t = (double*)(base + kStaticHitT);
    scr.opaque.hit_obj = (uint32_t*)(base + kStaticHitObj);
    scr.opaque.shadow = (uint16_t*)(base + kStaticShadow);
  }
  // from here on: stream waits, a memset of the counters, kernel launches and event records -- no allocation
  Frame f;
  if ((rc = frame_begin(f, ctx, scene, camera, hip_stream, scr)) != RTK_OK) return rc;
  if ((rc = frame_launch(f, p, d_rgb_f32, d_rgba8, true)) != RTK_OK) return rc;
  return frame_end(f, ResolveStep{}, nullptr);
}


int rtk_render_device_sampled(rtk_context* ctx, rtk_scene_handle* scene, const rtk_camera* camera, uint32_t s,
                              const rtk_tiling* tiling, float* d_rgb_f32, uint8_t* d_rgba8, void* hip_stream) {
  if (!ctx || !scene || !camera) return fail(RTK_ERR_INVALID, "rtk_render_device_sampled: null argument");
  rtk_camera fine;
  int rc = rtk_camera_supersample(camera, s, &fine);
  if (rc != RTK_OK) return rc;
  rtk_tiling tl = {0u, 1u};
  if (tiling) tl = *tiling;
  rtk::Partition p;
  if ((rc = rtk::partition_from_tiling(&fine, &tl, p, "rtk_render_device_sampled")) != RTK_OK) return rc;
  if (s == 1) return render_device(ctx, scene, camera, p, d_rgb_f32, d_rgba8, hip_stream, ResolveStep{});
  if (tl.world > 1 && 8u % s != 0)
    return fail(RTK_ERR_UNSUPPORTED, "rtk_render_device_sampled: with world > 1 samples_per_axis must be 1, 2, 4 or 8");
  RTK_HIP(hipSetDevice(ctx->device));
  const size_t samples = tl.world > 1 ? (size_t)rtk_compact_pixels(fine.width, fine.height, tl.rank, tl.world)
                                      : (size_t)fine.width * fine.height;
  if (samples > ctx->cap_samples) {
    // the buffer may still be read by an earlier call on another stream
    RTK_HIP(hipDeviceSynchronize());
    if (ctx->d_samples) (void)hipFree(ctx->d_samples);
    ctx->d_samples = nullptr;
    ctx->cap_samples = 0;
    RTK_HIP(hipMalloc((void**)&ctx->d_samples, samples * 3 * sizeof(float)));
    ctx->cap_samples = samples;
  }
  ResolveStep step;
  step.s = s, step.width = camera->width, step.height = camera->height;
  step.d_rgb = d_rgb_f32, step.d_rgba8 = d_rgba8;
  step.tiles = tl.world > 1;
  return render_device(ctx, scene, &fine, p, ctx->d_samples, nullptr, hip_stream, step);
}

int rtk_resolve(rtk_context* ctx, uint32_t width, uint32_t height, uint32_t s, const float* d_samples_rgb_f32,
                float* d_rgb_f32, uint8_t* d_rgba8, void* hip_stream) {
  if (!ctx || !d_samples_rgb_f32 || width == 0 || height == 0) return fail(RTK_ERR_INVALID, "rtk_resolve: bad argument");
  if (s == 0 || s > RTK_MAX_SAMPLES_PER_AXIS)
    return fail(RTK_ERR_INVALID, "rtk_resolve: samples_per_axis must be 1.." + std::to_string(RTK_MAX_SAMPLES_PER_AXIS));
  RTK_HIP(hipSetDevice(ctx->device));
  hipStream_t st = hip_stream ? (hipStream_t)hip_stream : ctx->stream;
  RTK_HIP(rtk::launch_resolve(st, width, height, s, d_samples_rgb_f32, d_rgb_f32, d_rgba8));
  return RTK_OK;
}

// Kernel times of the call that used timing slot `slot`; the caller has synchronised on it.
static int slot_times(rtk_context* ctx, int slot, double* kernel_ms, double* rays_ms) {
  float ms = 0.f;
  RTK_HIP(hipEventElapsedTime(&ms, ctx->ev_start[slot], ctx->ev_stop[slot]));
  float rms = ms;
  if (ctx->ev_split[slot]) RTK_HIP(hipEventElapsedTime(&rms, ctx->ev_start[slot], ctx->ev_rays[slot]));
  *kernel_ms = ms;
  *rays_ms = rms;
  return RTK_OK;
}

int rtk_get_stats(rtk_context* ctx, rtk_stats* stats) {
  if (!ctx || !stats) return fail(RTK_ERR_INVALID, "rtk_get_stats: null argument");
  std::memset(stats, 0, sizeof *stats);
  if (ctx->n_timed == 0) return RTK_OK;
  RTK_HIP(hipSetDevice(ctx->device));
  const int slot = (int)((ctx->n_timed - 1) % RTK_TIMING_HISTORY);
  RTK_HIP(hipEventSynchronize(ctx->ev_stop[slot]));
  const int rc = slot_times(ctx, slot, &stats->kernel_ms, &stats->rays_ms);
  if (rc != RTK_OK) return rc;
  rtk::DevCounters shards[rtk::kCounterShards];
  RTK_HIP(hipMemcpyAsync(shards, ctx->last_counters, sizeof shards, hipMemcpyDeviceToHost, ctx->last_stream));
  RTK_HIP(hipStreamSynchronize(ctx->last_stream));
  for (const rtk::DevCounters& c : shards) {
    stats->rays_primary += c.rays_primary;
    stats->rays_shadow += c.rays_shadow;
    stats->rays_secondary += c.rays_secondary;
    stats->exact_tests += c.exact_tests;
  }
  return RTK_OK;
}

int rtk_get_timing_history(rtk_context* ctx, uint32_t capacity, double* kernel_ms, double* rays_ms) {
  if (!ctx || !kernel_ms || !rays_ms) return fail(RTK_ERR_INVALID, "rtk_get_timing_history: null argument");
  uint64_t n = ctx->n_timed < RTK_TIMING_HISTORY ? ctx->n_timed : RTK_TIMING_HISTORY;
  if (n > capacity) n = capacity;
  if (n == 0) return 0;
  RTK_HIP(hipSetDevice(ctx->device));
  RTK_HIP(hipEventSynchronize(ctx->ev_stop[(ctx->n_timed - 1) % RTK_TIMING_HISTORY]));
  for (uint64_t i = 0; i < n; ++i) {
    const uint64_t call = ctx->n_timed - n + i;  // oldest of the n first
    const int slot = (int)(call % RTK_TIMING_HISTORY);
    RTK_HIP(hipEventSynchronize(ctx->ev_stop[slot]));  // calls may have gone to different streams
    const int rc = slot_times(ctx, slot, &kernel_ms[i], &rays_ms[i]);
    if (rc != RTK_OK) return rc;
  }
  return (int)n;
}

int rtk_render(rtk_context* ctx, const rtk_scene* scene, const rtk_camera* camera, float* rgb_f32_out, uint8_t* rgba8_out,
               rtk_stats* stats) {
  return rtk_render_sampled(ctx, scene, camera, 1, rgb_f32_out, rgba8_out, stats);
}

// Whether the device can write `p` directly (page-locked or registered host memory).  Ordinary
// pageable memory makes hipMemcpyAsync stage and block; such buffers are filled through the
// context's page-locked bounce buffer instead.
static bool host_pointer_is_pinned(const void* p) {
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
    (void)hipGetLastError();  // "invalid value" is the answer for pageable memory, not a failure
    return false;
  }
  return attr.type == hipMemoryTypeHost || attr.type == hipMemoryTypeManaged || attr.type == hipMemoryTypeUnified;
}

// The frame of `camera` into host buffers, one sample per pixel: the frame is cut into blocks of
// whole tile rows, rendered bottom-up on the context's stream; each block's rows start for the
// host on the copy stream as soon as its kernel has finished, while the next block renders.
// Pageable destinations receive their blocks from a page-locked bounce buffer, copied by a few
// host threads as the blocks land.
static int render_to_host(rtk_context* ctx, rtk_scene_handle* h, const rtk_camera* camera, float* rgb_f32_out,
                          uint8_t* rgba8_out) {
  const uint32_t width = camera->width, height = camera->height;
  const size_t pixels = (size_t)width * height;
  const uint32_t rows = tiles_down(height);
  // about a million pixels per block: enough tiles per launch to fill the GPU several times over
  const uint32_t n_blocks =
      std::max<uint32_t>(1u, std::min<uint32_t>({(uint32_t)std::min<size_t>((pixels + (1u << 19)) >> 20, 1u << 20),
                                                 (uint32_t)rtk_context::kHostBlocks, rows}));
  const uint32_t per_block = (rows + n_blocks - 1) / n_blocks;
  Scratch scr;
  int rc = context_scratch(ctx, h, (size_t)per_block * tiles_across(width) * 256u, scr);
  if (rc != RTK_OK) return rc;
  // bounce buffer: FP32 part first, RGBA8 part behind it
  const bool bounce_rgb = rgb_f32_out && !host_pointer_is_pinned(rgb_f32_out);
  const bool bounce_px = rgba8_out && !host_pointer_is_pinned(rgba8_out);
  if (bounce_rgb || bounce_px) {
    RTK_HIP(hipSetDevice(ctx->device));
    const size_t need = pixels * 16u;
    if (need > ctx->cap_bounce) {
      if (ctx->h_bounce) (void)hipHostFree(ctx->h_bounce);
      ctx->h_bounce = nullptr;
      ctx->cap_bounce = 0;
      RTK_HIP(hipHostMalloc((void**)&ctx->h_bounce, need, hipHostMallocDefault));
      ctx->cap_bounce = need;
    }
  }
  float* const land_rgb = bounce_rgb ? reinterpret_cast<float*>(ctx->h_bounce) : rgb_f32_out;
  uint8_t* const land_px = bounce_px ? ctx->h_bounce + pixels * 12u : rgba8_out;

  Frame f;
  if ((rc = frame_begin(f, ctx, h, camera, nullptr, scr)) != RTK_OK) return rc;
  float* d_rgb = rgb_f32_out ? ctx->d_rgb : nullptr;
  uint8_t* d_rgba8 = rgba8_out ? ctx->d_rgba8 : nullptr;
  struct Block {
    size_t first, count;  // pixels
  };
  Block blocks[rtk_context::kHostBlocks] = {};
  uint32_t n_queued = 0;
  for (uint32_t i = 0; i < n_blocks; ++i) {
    const uint32_t b = n_blocks - 1u - i;  // bottom block first
    const uint32_t row_a = b * per_block, row_b = std::min(rows, row_a + per_block);
    if (row_a >= row_b) continue;
    rtk_banding bd = {0u, 1u, row_b - row_a, row_a, row_b - row_a, 0u};
    rtk::Partition p;
    if ((rc = rtk::partition_from_banding(camera, &bd, p, "rtk_render")) != RTK_OK) return rc;
    if ((rc = frame_launch(f, p, d_rgb, d_rgba8, n_blocks == 1u)) != RTK_OK) return rc;
    RTK_HIP(hipEventRecord(ctx->ev_block[n_queued], f.st));
    RTK_HIP(hipStreamWaitEvent(ctx->copy_stream, ctx->ev_block[n_queued], 0));
    const size_t y0 = (size_t)row_a * RTK_TILE_H, y1 = std::min<size_t>(height, (size_t)row_b * RTK_TILE_H);
    const size_t first = y0 * width, count = (y1 - y0) * width;
    if (rgb_f32_out)
      RTK_HIP(hipMemcpyAsync(land_rgb + first * 3, d_rgb + first * 3, count * 3 * sizeof(float), hipMemcpyDeviceToHost,
                             ctx->copy_stream));
    if (rgba8_out)
      RTK_HIP(hipMemcpyAsync(land_px + first * 4, d_rgba8 + first * 4, count * 4, hipMemcpyDeviceToHost, ctx->copy_stream));
    RTK_HIP(hipEventRecord(ctx->ev_landed[n_queued], ctx->copy_stream));
    blocks[n_queued++] = {first, count};
  }
  if ((rc = frame_end(f, ResolveStep{}, nullptr)) != RTK_OK) return rc;

  hipError_t worker_error = hipSuccess;
  if (bounce_rgb || bounce_px) {
    // host threads take pieces of the blocks in turn (a block is cut into kPieces so that the last
    // block to land is not one thread's job): wait until the block has landed, copy the piece
    constexpr uint32_t kPieces = 4;
    const uint32_t hw = std::thread::hardware_concurrency();
    const uint32_t n_threads = std::max(1u, std::min({n_queued * kPieces, 8u, hw ? hw : 1u}));
    std::atomic<uint32_t> next{0};
    std::atomic<int> first_error{(int)hipSuccess};
    auto work = [&]() {
      (void)hipSetDevice(ctx->device);
      for (uint32_t item = next.fetch_add(1); item < n_queued * kPieces; item = next.fetch_add(1)) {
        const uint32_t k = item / kPieces, piece = item % kPieces;
        const hipError_t e = hipEventSynchronize(ctx->ev_landed[k]);
        if (e != hipSuccess) {
          int expected = (int)hipSuccess;
          first_error.compare_exchange_strong(expected, (int)e);
          continue;
        }
        const size_t a = blocks[k].first + blocks[k].count * piece / kPieces;
        const size_t n = blocks[k].first + blocks[k].count * (piece + 1) / kPieces - a;
        if (bounce_rgb) std::memcpy(rgb_f32_out + a * 3, land_rgb + a * 3, n * 12u);
        if (bounce_px) std::memcpy(rgba8_out + a * 4, land_px + a * 4, n * 4u);
      }
    };
    std::vector<std::thread> pool;
    try {
      pool.reserve(n_threads);
      for (uint32_t t = 1; t < n_threads; ++t) pool.emplace_back(work);
    } catch (...) {  // fewer threads than hoped for: the ones that started and this one do the work
    }
    work();
    for (std::thread& t : pool) t.join();
    worker_error = (hipError_t)first_error.load();
  }
  RTK_HIP(hipStreamSynchronize(ctx->copy_stream));
  RTK_HIP(hipStreamSynchronize(ctx->stream));
  if (worker_error != hipSuccess) return fail_hip(worker_error, "rtk_render: waiting for a block of the frame");
  return RTK_OK;
}

// The device copy of `scene` for rtk_render: the one kept from the previous call when the caller
// passes the same scene again (same counts, byte-identical objects and lights), else a fresh upload
// that replaces it.  The handle belongs to the context.
static int cached_scene(rtk_context* ctx, const rtk_scene* scene, rtk_scene_handle** out) {
  if ((scene->n_objects && !scene->objects) || (scene->n_lights && !scene->lights))
    return fail(RTK_ERR_INVALID, "rtk_scene_upload: null object or light array");
  const rtk_scene& c = ctx->cached_scene;
  // (a scene whose materials point to inner pattern transforms is never taken for the previous one:
  // the comparison below sees the pointers, not what they point to)
  if (ctx->cached_handle && !ctx->cached_has_pointers && c.n_objects == scene->n_objects && c.n_lights == scene->n_lights && c.max_depth == scene->max_depth &&
      std::memcmp(c.ambient_color, scene->ambient_color, sizeof c.ambient_color) == 0 &&
      (scene->n_objects == 0 ||
       std::memcmp(ctx->cached_objects.data(), scene->objects, (size_t)scene->n_objects * sizeof(rtk_object)) == 0) &&
      (scene->n_lights == 0 ||
       std::memcmp(ctx->cached_lights.data(), scene->lights, (size_t)scene->n_lights * sizeof(rtk_light)) == 0)) {
    *out = ctx->cached_handle;
    return RTK_OK;
  }
  // A new scene.  The caller's arrays are copied for the next call's comparison on a thread of their
  // own while this one flattens them; the device allocation of the previous scene is used again when
  // it is large enough (its last frame was synchronised before rtk_render returned).
  bool copied = true;
  std::thread copier;
  auto copy_scene = [&]() {
    try {
      ctx->cached_objects.assign(scene->objects, scene->objects + scene->n_objects);
      ctx->cached_lights.assign(scene->lights, scene->lights + scene->n_lights);
    } catch (const std::bad_alloc&) {
      copied = false;
    }
  };
  rtk_scene_handle* const previous = ctx->cached_handle;
  ctx->cached_handle = nullptr;
  try {
    copier = std::thread(copy_scene);
  } catch (const std::system_error&) {  // no thread to be had: copy here
    copy_scene();
  }
  rtk::FlatScene flat;
  flat.grow = context_staging;
  flat.user = ctx;
  rtk_scene_handle* h = nullptr;
  int rc = rtk::flatten_scene(scene, flat);
  if (rc == RTK_OK)
    rc = rtk::place_scene(ctx, flat, &h, previous);
  else if (previous)
    rtk_scene_release(ctx, previous);
  if (copier.joinable()) copier.join();
  if (rc != RTK_OK) {
    const std::string keep = rtk_last_error();
    ctx->cached_objects.clear();
    ctx->cached_lights.clear();
    return fail(rc, keep);
  }
  if (!copied) {
    rtk_scene_release(ctx, h);
    ctx->cached_objects.clear();
    ctx->cached_lights.clear();
    return fail(RTK_ERR_NOMEM, "rtk_render: out of host memory");
  }
  ctx->cached_scene = *scene;
  ctx->cached_has_pointers = false;
  for (uint32_t i = 0; i < scene->n_objects && !ctx->cached_has_pointers; ++i)
    ctx->cached_has_pointers = scene->objects[i].material.inner_pattern_transforms != nullptr;
  ctx->cached_handle = h;
  *out = h;
  return RTK_OK;
}

int rtk_render_sampled(rtk_context* ctx, const rtk_scene* scene, const rtk_camera* camera, uint32_t samples_per_axis,
                       float* rgb_f32_out, uint8_t* rgba8_out, rtk_stats* stats) {
  if (!ctx || !scene || !camera) return fail(RTK_ERR_INVALID, "rtk_render: null argument");
  if (samples_per_axis == 0 || samples_per_axis > RTK_MAX_SAMPLES_PER_AXIS)
    return fail(RTK_ERR_INVALID, "rtk_render_sampled: samples_per_axis must be 1.." + std::to_string(RTK_MAX_SAMPLES_PER_AXIS));
  if (camera->width == 0 || camera->height == 0) return fail(RTK_ERR_INVALID, "camera has zero size");
  rtk_scene_handle* h = nullptr;
  int rc = cached_scene(ctx, scene, &h);
  if (rc != RTK_OK) return rc;
  const size_t pixels = (size_t)camera->width * camera->height;
  if (pixels > ctx->cap_pixels) {
    if (ctx->d_rgb) (void)hipFree(ctx->d_rgb);
    if (ctx->d_rgba8) (void)hipFree(ctx->d_rgba8);
    ctx->d_rgb = nullptr;
    ctx->d_rgba8 = nullptr;
    ctx->cap_pixels = 0;
    hipError_t e = hipMalloc((void**)&ctx->d_rgb, pixels * 3 * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void**)&ctx->d_rgba8, pixels * 4);
    if (e != hipSuccess) return fail_hip(e, "rtk_render: framebuffer allocation");
    ctx->cap_pixels = pixels;
  }
  if (samples_per_axis == 1) {
    rc = render_to_host(ctx, h, camera, rgb_f32_out, rgba8_out);
  } else {
    rc = rtk_render_device_sampled(ctx, h, camera, samples_per_axis, nullptr, rgb_f32_out ? ctx->d_rgb : nullptr,
                                   rgba8_out ? ctx->d_rgba8 : nullptr, nullptr);
    hipError_t e = hipSuccess;
    if (rc == RTK_OK && rgb_f32_out)
      e = hipMemcpyAsync(rgb_f32_out, ctx->d_rgb, pixels * 3 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream);
    if (rc == RTK_OK && e == hipSuccess && rgba8_out)
      e = hipMemcpyAsync(rgba8_out, ctx->d_rgba8, pixels * 4, hipMemcpyDeviceToHost, ctx->stream);
    if (rc == RTK_OK && e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (rc == RTK_OK && e != hipSuccess) rc = fail_hip(e, "rtk_render: copy back");
  }
  if (rc == RTK_OK && stats) rc = rtk_get_stats(ctx, stats);
  return rc;
}

int rtk_trace_rays(rtk_context* ctx, rtk_scene_handle* scene, const double* origins, const double* directions, uint64_t n,
                   double* colors_out) {
  if (!ctx || !scene || (n && (!origins || !directions || !colors_out)))
    return fail(RTK_ERR_INVALID, "rtk_trace_rays: null argument");
  for (uint64_t i = 0; i < n; ++i)
    if (origins[4 * i + 3] != 1.0 || directions[4 * i + 3] != 0.0)
      return fail(RTK_ERR_UNSUPPORTED, "rtk_trace_rays: origins need w = 1 and directions w = 0");
  if (n == 0) return RTK_OK;
  RTK_HIP(hipSetDevice(ctx->device));
  double *d_o = nullptr, *d_d = nullptr, *d_c = nullptr;
  hipError_t e = hipMalloc((void**)&d_o, n * 4 * sizeof(double));
  if (e == hipSuccess) e = hipMalloc((void**)&d_d, n * 4 * sizeof(double));
  if (e == hipSuccess) e = hipMalloc((void**)&d_c, n * 3 * sizeof(double));
  hipStream_t st = ctx->stream;
  if (e == hipSuccess && ctx->n_timed && ctx->last_stream != st)  // behind the context's previous call (shared counters)
    e = hipStreamWaitEvent(st, ctx->ev_stop[(ctx->n_timed - 1) % RTK_TIMING_HISTORY], 0);
  if (e == hipSuccess) e = hipMemcpyAsync(d_o, origins, n * 4 * sizeof(double), hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipMemcpyAsync(d_d, directions, n * 4 * sizeof(double), hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipMemsetAsync(ctx->d_counters, 0, sizeof(rtk::DevCounters) * rtk::kCounterSlots, st);
  const int slot = (int)(ctx->n_timed % RTK_TIMING_HISTORY);
  ctx->ev_split[slot] = false;
  if (e == hipSuccess) e = hipEventRecord(ctx->ev_start[slot], st);
  if (e == hipSuccess) e = rtk::launch_trace_rays(st, scene->dev, d_o, d_d, n, d_c, ctx->d_counters);
  if (e == hipSuccess) e = hipEventRecord(ctx->ev_stop[slot], st);
  if (e == hipSuccess) e = hipMemcpyAsync(colors_out, d_c, n * 3 * sizeof(double), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (d_o) (void)hipFree(d_o);
  if (d_d) (void)hipFree(d_d);
  if (d_c) (void)hipFree(d_c);
  if (e != hipSuccess) return fail_hip(e, "rtk_trace_rays");
  ctx->last_stream = st;
  ctx->last_counters = ctx->d_counters;
  ++ctx->n_timed;
  return RTK_OK;
}

static int untile_checked(rtk_context* ctx, uint32_t width, uint32_t height, uint32_t world, uint64_t rgb_stride_bytes,
                          uint64_t rgba8_stride_bytes, const float* d_rgb_f32_compact, float* d_rgb_f32_frame,
                          const uint8_t* d_rgba8_compact, uint8_t* d_rgba8_frame, void* hip_stream, const char* who) {
  if (!ctx || width == 0 || height == 0 || world == 0) return fail(RTK_ERR_INVALID, std::string(who) + ": bad argument");
  if ((d_rgb_f32_compact == nullptr) != (d_rgb_f32_frame == nullptr) || (d_rgba8_compact == nullptr) != (d_rgba8_frame == nullptr))
    return fail(RTK_ERR_INVALID, std::string(who) + ": compact and frame pointers must come in pairs");
  if (rgb_stride_bytes % 4u != 0 || rgba8_stride_bytes % 4u != 0)
    return fail(RTK_ERR_INVALID, std::string(who) + ": rank strides must be multiples of 4 bytes");
  const uint64_t largest = rtk_compact_pixels(width, height, 0, world);  // rank 0 owns the most tiles
  if ((d_rgb_f32_compact && world > 1 && rgb_stride_bytes < largest * 12u) ||
      (d_rgba8_compact && world > 1 && rgba8_stride_bytes < largest * 4u))
    return fail(RTK_ERR_INVALID, std::string(who) + ": rank stride smaller than a rank's compact buffer");
  RTK_HIP(hipSetDevice(ctx->device));
  hipStream_t st = hip_stream ? (hipStream_t)hip_stream : ctx->stream;
  RTK_HIP(rtk::launch_untile(st, width, height, world, rgb_stride_bytes / 4u, rgba8_stride_bytes / 4u, d_rgb_f32_compact,
                             d_rgb_f32_frame, d_rgba8_compact, d_rgba8_frame));
  return RTK_OK;
}

int rtk_untile(rtk_context* ctx, uint32_t width, uint32_t height, uint32_t world, uint64_t slot_pixels,
               const float* d_rgb_f32_compact, float* d_rgb_f32_frame, const uint8_t* d_rgba8_compact,
               uint8_t* d_rgba8_frame, void* hip_stream) {
  return untile_checked(ctx, width, height, world, slot_pixels * 12u, slot_pixels * 4u, d_rgb_f32_compact, d_rgb_f32_frame,
                        d_rgba8_compact, d_rgba8_frame, hip_stream, "rtk_untile");
}

int rtk_untile_strided(rtk_context* ctx, uint32_t width, uint32_t height, uint32_t world, uint64_t rgb_rank_stride_bytes,
                       uint64_t rgba8_rank_stride_bytes, const float* d_rgb_f32_compact, float* d_rgb_f32_frame,
                       const uint8_t* d_rgba8_compact, uint8_t* d_rgba8_frame, void* hip_stream) {
  return untile_checked(ctx, width, height, world, rgb_rank_stride_bytes, rgba8_rank_stride_bytes, d_rgb_f32_compact,
                        d_rgb_f32_frame, d_rgba8_compact, d_rgba8_frame, hip_stream, "rtk_untile_strided");
}

int rtk_untile_sampled(rtk_context* ctx, uint32_t width, uint32_t height, uint32_t s, uint32_t world,
                       uint64_t rgb_rank_stride_bytes, uint64_t rgba8_rank_stride_bytes, const float* d_rgb_f32_compact,
                       float* d_rgb_f32_frame, const uint8_t* d_rgba8_compact, uint8_t* d_rgba8_frame, void* hip_stream) {
  const char* who = "rtk_untile_sampled";
  if (s == 1)
    return untile_checked(ctx, width, height, world, rgb_rank_stride_bytes, rgba8_rank_stride_bytes, d_rgb_f32_compact,
                          d_rgb_f32_frame, d_rgba8_compact, d_rgba8_frame, hip_stream, who);
  if (!ctx || width == 0 || height == 0 || world == 0) return fail(RTK_ERR_INVALID, std::string(who) + ": bad argument");
  if (s == 0 || s > 8 || 8u % s != 0) return fail(RTK_ERR_UNSUPPORTED, std::string(who) + ": samples_per_axis must be 1, 2, 4 or 8");
  if ((uint64_t)width * s > 0xffffffffull || (uint64_t)height * s > 0xffffffffull)
    return fail(RTK_ERR_UNSUPPORTED, std::string(who) + ": sample frame wider or taller than 2^32 - 1");
  if ((d_rgb_f32_compact == nullptr) != (d_rgb_f32_frame == nullptr) || (d_rgba8_compact == nullptr) != (d_rgba8_frame == nullptr))
    return fail(RTK_ERR_INVALID, std::string(who) + ": compact and frame pointers must come in pairs");
  if (rgb_rank_stride_bytes % 4u != 0 || rgba8_rank_stride_bytes % 4u != 0)
    return fail(RTK_ERR_INVALID, std::string(who) + ": rank strides must be multiples of 4 bytes");
  const uint64_t largest = rtk_compact_pixels_sampled(width, height, s, 0, world);
  if (world > 1 && ((d_rgb_f32_compact && rgb_rank_stride_bytes < largest * 12u) ||
                    (d_rgba8_compact && rgba8_rank_stride_bytes < largest * 4u)))
    return fail(RTK_ERR_INVALID, std::string(who) + ": rank stride smaller than a rank's compact buffer");
  if (world == 1) return fail(RTK_ERR_INVALID, std::string(who) + ": a single rank renders row-major frames; nothing to rebuild");
  RTK_HIP(hipSetDevice(ctx->device));
  hipStream_t st = hip_stream ? (hipStream_t)hip_stream : ctx->stream;
  RTK_HIP(rtk::launch_untile_sampled(st, width, height, s, world, rgb_rank_stride_bytes / 4u, rgba8_rank_stride_bytes / 4u,
                                     d_rgb_f32_compact, d_rgb_f32_frame, d_rgba8_compact, d_rgba8_frame));
  return RTK_OK;
}

}  // extern "C"
