// rtk_api.hip -- the C ABI of include/rtk.h: validation, scene flattening, HBM residency
// and kernel launches.  Host code only; the kernels live in rtk_kernels.hip.
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

// The f64 matrix code below (cofactor inverse, look_at) must round like the reference's: no
// fused multiply-adds, whatever the host architecture or -march offers.
#pragma clang fp contract(off)

namespace {
thread_local std::string g_error;
}

namespace rtk {
int fail(int code, const std::string& msg) {
  g_error = msg;
  return code;
}
int fail_hip(hipError_t e, const char* what) {
  g_error = std::string(what) + ": " + hipGetErrorString(e);
  return RTK_ERR_HIP;
}
}  // namespace rtk

using rtk::fail;
using rtk::fail_hip;

// ---------------------------------------------------------------- static scratch (rtk_render_device_static)
// Device memory that exists without any allocation call: one array per GPU inside the library's
// code object.  Layout (offsets in bytes), sized by the limits in include/rtk.h:
//   counters      kCounterShards x DevCounters
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

namespace {

// ---------------------------------------------------------------- f64 matrix helpers
// Row-major 4x4, element (row, col) at [col + 4*row] (src/maths/matrices.rs:69-71).
struct M4 {
  double e[16];
  double at(int r, int c) const { return e[c + 4 * r]; }
  double& at(int r, int c) { return e[c + 4 * r]; }
};
M4 m4_identity() {
  M4 m{};
  for (int i = 0; i < 4; ++i) m.at(i, i) = 1.0;
  return m;
}
M4 m4_load(const double* p) {
  M4 m;
  std::memcpy(m.e, p, sizeof m.e);
  return m;
}
// Determinant of the 3x3 block of a 4x4 made of rows r[0..2] and columns c[0..2], by Laplace
// expansion along its first row exactly as the reference computes determinants
// (src/maths/matrices.rs:172-180, 256-264, 282-286): the sum starts at 0.0, every minor is
// m[0]*m[3] - m[1]*m[2], odd terms enter with the minor negated.  Same operations in the same
// order as the reference's recursion over copied sub-matrices, without the copies.
inline double det3_of(const double* e, const int* r, const int* c) {
  const double a0 = e[c[0] + 4 * r[0]], a1 = e[c[1] + 4 * r[0]], a2 = e[c[2] + 4 * r[0]];
  const double a3 = e[c[0] + 4 * r[1]], a4 = e[c[1] + 4 * r[1]], a5 = e[c[2] + 4 * r[1]];
  const double a6 = e[c[0] + 4 * r[2]], a7 = e[c[1] + 4 * r[2]], a8 = e[c[2] + 4 * r[2]];
  const double m0 = a4 * a8 - a5 * a7, m1 = a3 * a8 - a5 * a6, m2 = a3 * a7 - a4 * a6;
  double result = 0.0;
  result += a0 * m0;
  result += a1 * -m1;
  result += a2 * m2;
  return result;
}
constexpr int kOthers[4][3] = {{1, 2, 3}, {0, 2, 3}, {0, 1, 3}, {0, 1, 2}};  // the indices that are not i
// Inverse by cofactors / determinant (src/maths/matrices.rs:183-198).  false = singular.
bool m4_invert(const M4& m, M4& out) {
  double det = 0.0;
  for (int i = 0; i < 4; ++i) {
    const double minor = det3_of(m.e, kOthers[0], kOthers[i]);
    det += m.e[i] * ((i % 2 == 0) ? minor : -minor);
  }
  if (det == 0.0) return false;
  for (int row = 0; row < 4; ++row)
    for (int col = 0; col < 4; ++col) {
      const double minor = det3_of(m.e, kOthers[row], kOthers[col]);
      out.at(col, row) = (((row + col) % 2 == 0) ? minor : -minor) / det;
    }
  return true;
}
bool is_affine(const double* t) { return t[12] == 0.0 && t[13] == 0.0 && t[14] == 0.0 && t[15] == 1.0; }

// Largest singular value of the upper-left 3x3 (the radius of the smallest origin-centred
// ball containing the image of the unit sphere), never under-estimated.
double sigma_max3(const M4& t) {
  double a[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      a[i][j] = 0.0;
      for (int k = 0; k < 3; ++k) a[i][j] += t.at(k, i) * t.at(k, j);  // M^T M
    }
  const double tr = a[0][0] + a[1][1] + a[2][2];
  const double off = std::fabs(a[0][1]) + std::fabs(a[0][2]) + std::fabs(a[1][2]);
  const double dmax = std::fmax(a[0][0], std::fmax(a[1][1], a[2][2]));
  if (off <= 1e-14 * tr) return std::sqrt(dmax) * (1.0 + 1e-12);  // rotation * axis scaling up to round-off
  // cyclic Jacobi on the symmetric 3x3
  for (int sweep = 0; sweep < 32; ++sweep) {
    double o = std::fabs(a[0][1]) + std::fabs(a[0][2]) + std::fabs(a[1][2]);
    if (o <= 1e-18 * tr) break;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        if (a[p][q] == 0.0) continue;
        double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
        double tt = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        double c = 1.0 / std::sqrt(tt * tt + 1.0), s = tt * c;
        for (int k = 0; k < 3; ++k) {
          double akp = a[k][p], akq = a[k][q];
          a[k][p] = c * akp - s * akq;
          a[k][q] = s * akp + c * akq;
        }
        for (int k = 0; k < 3; ++k) {
          double apk = a[p][k], aqk = a[q][k];
          a[p][k] = c * apk - s * aqk;
          a[q][k] = s * apk + c * aqk;
        }
      }
  }
  double lam = std::fmax(a[0][0], std::fmax(a[1][1], a[2][2]));
  double o = std::fabs(a[0][1]) + std::fabs(a[0][2]) + std::fabs(a[1][2]);
  return std::sqrt(lam + o) * (1.0 + 1e-9);  // Gershgorin: leftover off-diagonal mass only enlarges
}

float round_up(double v) {
  float f = (float)v;
  if ((double)f < v) f = std::nextafterf(f, INFINITY);
  return f;
}

uint32_t round_up_to(uint32_t n, uint32_t m) { return (n + m - 1) / m * m; }

}  // namespace

extern "C" {

const char* rtk_last_error(void) { return g_error.c_str(); }

const char* rtk_version(void) { return "rtk 0.3 (gfx950, HIP)"; }

uint32_t rtk_abi_version(void) { return RTK_ABI_VERSION; }

int rtk_camera_init(rtk_camera* camera, uint32_t width, uint32_t height, double fov) {
  if (!camera || width == 0 || height == 0) return fail(RTK_ERR_INVALID, "rtk_camera_init: null camera or zero size");
  // src/scene/cameras.rs:19-49
  const double half_view = std::tan(fov / 2.0);
  const double aspect = (double)width / (double)height;
  double half_width, half_height;
  if (aspect >= 1.0) {
    half_width = half_view;
    half_height = half_width / aspect;
  } else {
    half_width = half_view * aspect;
    half_height = half_view;
  }
  camera->width = width;
  camera->height = height;
  camera->half_width = half_width;
  camera->half_height = half_height;
  camera->field_of_view = fov;
  camera->pixel_size = (half_width * 2.0) / (double)width;
  // look_at(point(0,1.5,-5), point(0,1,0), vec3(0,1,0)) (src/maths/transforms.rs:67-80)
  const double from[3] = {0.0, 1.5, -5.0}, to[3] = {0.0, 1.0, 0.0}, up[3] = {0.0, 1.0, 0.0};
  double f[3] = {to[0] - from[0], to[1] - from[1], to[2] - from[2]};
  double fm = std::sqrt(f[0] * f[0] + f[1] * f[1] + f[2] * f[2] + 0.0 * 0.0);
  for (double& v : f) v /= fm;
  double um = std::sqrt(up[0] * up[0] + up[1] * up[1] + up[2] * up[2] + 0.0 * 0.0);
  double u[3] = {up[0] / um, up[1] / um, up[2] / um};
  double left[3] = {f[1] * u[2] - f[2] * u[1], f[2] * u[0] - f[0] * u[2], f[0] * u[1] - f[1] * u[0]};
  double tu[3] = {left[1] * f[2] - left[2] * f[1], left[2] * f[0] - left[0] * f[2], left[0] * f[1] - left[1] * f[0]};
  M4 orientation = m4_identity();
  for (int c = 0; c < 3; ++c) {
    orientation.at(0, c) = left[c];
    orientation.at(1, c) = tu[c];
    orientation.at(2, c) = -f[c];
  }
  M4 tr = m4_identity();
  tr.at(0, 3) = -from[0];
  tr.at(1, 3) = -from[1];
  tr.at(2, 3) = -from[2];
  M4 out{};
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) {
      double sum = 0.0;
      for (int i = 0; i < 4; ++i) sum += orientation.at(r, i) * tr.at(i, c);
      out.at(r, c) = sum;
    }
  std::memcpy(camera->transform, out.e, sizeof out.e);
  return RTK_OK;
}

uint64_t rtk_tiles_for_rank(uint32_t width, uint32_t height, uint32_t rank, uint32_t world) {
  if (world == 0 || rank >= world) return 0;
  const uint64_t tiles = (uint64_t)((width + RTK_TILE_W - 1) / RTK_TILE_W) * ((height + RTK_TILE_H - 1) / RTK_TILE_H);
  return tiles / world + ((tiles % world) > rank ? 1 : 0);
}

uint64_t rtk_compact_pixels(uint32_t width, uint32_t height, uint32_t rank, uint32_t world) {
  return rtk_tiles_for_rank(width, height, rank, world) * (uint64_t)(RTK_TILE_W * RTK_TILE_H);
}

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
    w.rays = nullptr, w.outs = nullptr, w.cap_nodes = 0;
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

// A few host threads kept for the library's data-parallel host work (flattening a scene).  Starting a
// thread costs more than its share of flattening 10,000 objects does, so they are started once, on
// first use, and parked on a condition variable in between; they are never joined (the object lives
// until the process ends).  run(n, fn) calls fn(0) .. fn(n - 1), each once, on the parked threads
// and the caller, and returns when all have returned; fn must not throw.  Jobs from several threads
// take turns.  A forked child has none of the threads: there the caller does everything.
class HostPool {
 public:
  static HostPool& get() {
    static HostPool* pool = new HostPool();
    return *pool;
  }
  uint32_t width() const { return n_workers_ + 1u; }
  void run(uint32_t n, const std::function<void(uint32_t)>& fn) {
    if (n == 0) return;
    if (n == 1 || n_workers_ == 0) {
      for (uint32_t i = 0; i < n; ++i) fn(i);
      return;
    }
    std::lock_guard<std::mutex> one_job(run_mutex_);
    {
      std::lock_guard<std::mutex> lock(m_);
      job_ = &fn;
      job_n_ = n;
      next_.store(0, std::memory_order_relaxed);
      busy_ = n_workers_;
      ++generation_;
    }
    wake_.notify_all();
    work();
    std::unique_lock<std::mutex> lock(m_);
    done_.wait(lock, [&] { return busy_ == 0; });
    job_ = nullptr;
  }

 private:
  HostPool() {
    const uint32_t hw = std::thread::hardware_concurrency();
    const uint32_t want = std::min(8u, hw ? hw : 1u) - 1u;
    for (uint32_t t = 0; t < want; ++t) {
      try {
        std::thread([this] { worker(); }).detach();
        ++n_workers_;
      } catch (const std::system_error&) {
        break;
      }
    }
    (void)pthread_atfork(nullptr, nullptr, [] { HostPool::get().n_workers_ = 0; });
  }
  void work() {
    for (;;) {
      const uint32_t i = next_.fetch_add(1, std::memory_order_relaxed);
      if (i >= job_n_) break;
      (*job_)(i);
    }
  }
  void worker() {
    uint64_t seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lock(m_);
        wake_.wait(lock, [&] { return generation_ != seen; });
        seen = generation_;
      }
      work();
      bool last;
      {
        std::lock_guard<std::mutex> lock(m_);
        last = --busy_ == 0;
      }
      if (last) done_.notify_one();
    }
  }
  std::mutex run_mutex_, m_;
  std::condition_variable wake_, done_;
  const std::function<void(uint32_t)>* job_ = nullptr;
  uint32_t job_n_ = 0, busy_ = 0, n_workers_ = 0;
  uint64_t generation_ = 0;
  std::atomic<uint32_t> next_{0};
};

// A sphere on its way into sweep order: the bounding ball of its image and its object index.
struct SphereKey {
  double c[3];
  double r;
  uint32_t object;
};

// Reorders the spheres so that every aligned run of kBundle is spatially compact: recursive
// split along the longest axis of the centres' bounding box, the left part a multiple of
// kBundle.  Which spheres end up in which run does not depend on how the work is scheduled (ties
// by object index).  split_once does one split of [lo, hi) and returns where; cluster_split does
// all of them below a range.
static uint32_t split_once(SphereKey* keys, uint32_t lo, uint32_t hi) {
  double mn[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, mx[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
  for (uint32_t i = lo; i < hi; ++i)
    for (int a = 0; a < 3; ++a) mn[a] = std::fmin(mn[a], keys[i].c[a]), mx[a] = std::fmax(mx[a], keys[i].c[a]);
  int axis = 0;
  for (int a = 1; a < 3; ++a)
    if (mx[a] - mn[a] > mx[axis] - mn[axis]) axis = a;
  const uint32_t groups = (hi - lo + rtk::kBundle - 1) / rtk::kBundle;
  const uint32_t mid = lo + (groups / 2) * rtk::kBundle;
  std::nth_element(keys + lo, keys + mid, keys + hi, [axis](const SphereKey& p, const SphereKey& q) {
    return p.c[axis] < q.c[axis] || (p.c[axis] == q.c[axis] && p.object < q.object);
  });
  return mid;
}
static void cluster_split(SphereKey* keys, uint32_t lo, uint32_t hi) {
  while (hi - lo > (uint32_t)rtk::kBundle) {
    const uint32_t mid = split_once(keys, lo, hi);
    cluster_split(keys, lo, mid);
    lo = mid;
  }
}
// The whole order: the first three levels of the split one level at a time (the ranges of a level are
// independent), then the ranges left, each on one of the pool's threads.
static void cluster_order(SphereKey* keys, uint32_t n) {
  std::vector<std::pair<uint32_t, uint32_t>> ranges{{0u, n}};
  if (n >= 4096u) {
    for (int level = 0; level < 3; ++level) {
      std::vector<uint32_t> mids(ranges.size());
      HostPool::get().run((uint32_t)ranges.size(), [&](uint32_t r) {
        mids[r] = ranges[r].second - ranges[r].first > (uint32_t)rtk::kBundle ? split_once(keys, ranges[r].first, ranges[r].second)
                                                                              : ranges[r].second;
      });
      std::vector<std::pair<uint32_t, uint32_t>> next;
      for (size_t r = 0; r < ranges.size(); ++r) {
        next.push_back({ranges[r].first, mids[r]});
        if (mids[r] < ranges[r].second) next.push_back({mids[r], ranges[r].second});
      }
      ranges.swap(next);
    }
  }
  HostPool::get().run((uint32_t)ranges.size(), [&](uint32_t r) { cluster_split(keys, ranges[r].first, ranges[r].second); });
}

static int flatten_scene_checked(const rtk_scene* scene, rtk::FlatScene& out);

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

int rtk::flatten_scene(const rtk_scene* scene, rtk::FlatScene& out) {
  // the host-side record arrays are std::vectors: nothing may unwind through the C boundary
  try {
    return flatten_scene_checked(scene, out);
  } catch (const std::bad_alloc&) {
    return fail(RTK_ERR_NOMEM, "rtk_scene_upload: out of host memory");
  } catch (...) {
    return fail(RTK_ERR_INVALID, "rtk_scene_upload: unexpected failure while flattening the scene");
  }
}

// The light grids of a scene whose lights all have w = 0 (LightGrid in rtk_device.h), one per light,
// built side by side on the library's host threads.  keys: the spheres in sweep order; gd: per light the unit
// direction and rho = -1/|L|.  False when a grid would not pay (lists too long, too many lights):
// the scene then keeps the cluster sweeps for its shadow rays.
struct LightGrids {
  std::vector<rtk::LightGrid> header;
  std::vector<uint2> cells;
  std::vector<uint32_t> entries;
};
static bool build_light_grids(const SphereKey* keys, uint32_t n_sph, const double* center, const std::vector<double>& gd,
                              uint32_t n_lights, LightGrids& out) {
  if (n_lights == 0 || n_lights > 8u || n_sph == 0) return false;
  // the spheres as the FP32 filter sees them: recentred, rounded, radius grown by the rounding
  struct Disc {
    double c[3], r;
  };
  std::vector<Disc> discs(n_sph);
  double scale = 0.0;
  for (uint32_t j = 0; j < n_sph; ++j) {
    const double cd[3] = {keys[j].c[0] - center[0], keys[j].c[1] - center[1], keys[j].c[2] - center[2]};
    const float cf[3] = {(float)cd[0], (float)cd[1], (float)cd[2]};
    const double dx = cd[0] - cf[0], dy = cd[1] - cf[1], dz = cd[2] - cf[2];
    const double R = keys[j].r * (1.0 + 1e-9) + std::sqrt(dx * dx + dy * dy + dz * dz) * 1.0001;
    discs[j] = {{(double)cf[0], (double)cf[1], (double)cf[2]}, std::sqrt((double)round_up(R * R))};
    scale = std::fmax(scale, std::sqrt((double)cf[0] * cf[0] + (double)cf[1] * cf[1] + (double)cf[2] * cf[2]) + discs[j].r);
  }
  struct PerLight {
    rtk::LightGrid header{};
    std::vector<uint2> cells;
    std::vector<uint32_t> entries;
    bool ok = false;
  };
  std::vector<PerLight> per(n_lights);
  auto build = [&](uint32_t k) {
    PerLight& pl = per[k];
    const double g[3] = {gd[4 * k], gd[4 * k + 1], gd[4 * k + 2]}, arho = std::fabs(gd[4 * k + 3]);
    // basis of the plane perpendicular to g
    int a = 0;
    for (int i = 1; i < 3; ++i)
      if (std::fabs(g[i]) < std::fabs(g[a])) a = i;
    double ax[3] = {0, 0, 0};
    ax[a] = 1.0;
    double e1[3] = {ax[1] * g[2] - ax[2] * g[1], ax[2] * g[0] - ax[0] * g[2], ax[0] * g[1] - ax[1] * g[0]};
    const double n1 = std::sqrt(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]);
    for (double& v : e1) v /= n1;
    const double e2[3] = {g[1] * e1[2] - g[2] * e1[1], g[2] * e1[0] - g[0] * e1[2], g[0] * e1[1] - g[1] * e1[0]};
    const float e1f[3] = {(float)e1[0], (float)e1[1], (float)e1[2]}, e2f[3] = {(float)e2[0], (float)e2[1], (float)e2[2]};
    // every shading point on a sphere lies at or above the lowest point of the lowest sphere
    double h_min = DBL_MAX;
    for (const Disc& d : discs) h_min = std::fmin(h_min, d.c[0] * g[0] + d.c[1] * g[1] + d.c[2] * g[2] - d.r);
    const double margin = 1e-4 * (1.0 + scale);
    h_min -= 2.0 * margin;
    // a sphere's shadow at height h has radius^2 = R^2 (1 + R^2 rho^2 + 2 |rho| (c.g - h)): widest at h_min
    struct Shadow {
      double u, v, r;
    };
    std::vector<Shadow> sh(n_sph);
    double u0 = DBL_MAX, v0 = DBL_MAX, u1 = -DBL_MAX, v1 = -DBL_MAX, r_sum = 0.0;
    for (uint32_t j = 0; j < n_sph; ++j) {
      const Disc& d = discs[j];
      const double h = d.c[0] * g[0] + d.c[1] * g[1] + d.c[2] * g[2];
      const double r2 = d.r * d.r * (1.0 + d.r * d.r * arho * arho + 2.0 * arho * (h - h_min + d.r));
      const double r = std::sqrt(r2) * (1.0 + 1e-4) + 1e-5 * (1.0 + scale);
      // the device projects with the FP32 basis
      const double u = d.c[0] * e1f[0] + d.c[1] * e1f[1] + d.c[2] * e1f[2], v = d.c[0] * e2f[0] + d.c[1] * e2f[1] + d.c[2] * e2f[2];
      sh[j] = {u, v, r};
      u0 = std::fmin(u0, u - r), u1 = std::fmax(u1, u + r), v0 = std::fmin(v0, v - r), v1 = std::fmax(v1, v + r);
      r_sum += r;
    }
    if (!std::isfinite(u0) || !std::isfinite(u1) || !std::isfinite(v0) || !std::isfinite(v1) || !std::isfinite(r_sum)) return;
    const double du = u1 - u0, dv = v1 - v0;
    double cell = std::fmax(2.0 * r_sum / n_sph, std::sqrt(du * dv / (4.0 * n_sph)));
    cell = std::fmax(cell, std::fmax(du, dv) / 4000.0);
    if (!(cell > 0.0)) return;
    // the cell index is formed in FP32 on the device: take u0, v0 and 1/cell as it will see them
    const float u0f = (float)(u0 - 0.5 * cell), v0f = (float)(v0 - 0.5 * cell), inv = (float)(1.0 / cell);
    const uint32_t nx = (uint32_t)((u1 - (double)u0f) * (double)inv) + 2u, ny = (uint32_t)((v1 - (double)v0f) * (double)inv) + 2u;
    if (nx > 4096u || ny > 4096u) return;
    auto range = [&](double lo, double hi, double origin, uint32_t n, uint32_t& i0, uint32_t& i1) {
      // a hundredth of a cell either side covers the rounding of the device's index arithmetic
      const double a0 = (lo - origin) * (double)inv - 0.01, a1 = (hi - origin) * (double)inv + 0.01;
      i0 = a0 <= 0.0 ? 0u : std::min<uint32_t>((uint32_t)a0, n - 1u);
      i1 = a1 <= 0.0 ? 0u : std::min<uint32_t>((uint32_t)a1, n - 1u);
    };
    pl.cells.assign((size_t)nx * ny, make_uint2(0u, 0u));
    size_t total = 0;
    for (uint32_t j = 0; j < n_sph; ++j) {
      uint32_t i0, i1, j0, j1;
      range(sh[j].u - sh[j].r, sh[j].u + sh[j].r, (double)u0f, nx, i0, i1);
      range(sh[j].v - sh[j].r, sh[j].v + sh[j].r, (double)v0f, ny, j0, j1);
      for (uint32_t y = j0; y <= j1; ++y)
        for (uint32_t x = i0; x <= i1; ++x) ++pl.cells[(size_t)y * nx + x].y;
      total += (size_t)(i1 - i0 + 1u) * (j1 - j0 + 1u);
    }
    if (total > (size_t)48u * n_sph) return;  // lists this long: the grid would not beat the cluster sweeps
    uint32_t at = 0;
    for (uint2& c : pl.cells) {
      c.x = at;
      at += c.y;
      c.y = 0;
    }
    pl.entries.resize(total);
    for (uint32_t j = 0; j < n_sph; ++j) {
      uint32_t i0, i1, j0, j1;
      range(sh[j].u - sh[j].r, sh[j].u + sh[j].r, (double)u0f, nx, i0, i1);
      range(sh[j].v - sh[j].r, sh[j].v + sh[j].r, (double)v0f, ny, j0, j1);
      for (uint32_t y = j0; y <= j1; ++y)
        for (uint32_t x = i0; x <= i1; ++x) {
          uint2& c = pl.cells[(size_t)y * nx + x];
          pl.entries[c.x + c.y++] = j;
        }
    }
    rtk::LightGrid& G = pl.header;
    for (int i = 0; i < 3; ++i) G.e1[i] = e1f[i], G.e2[i] = e2f[i];
    G.h_min = round_up(h_min + margin);  // the device's (p - centre).g is off by far less than the margin
    G.u0 = u0f, G.v0 = v0f, G.inv_cell = inv;
    G.nx = nx, G.ny = ny;
    pl.ok = true;
  };
  HostPool::get().run(n_lights, [&](uint32_t k) {
    try {
      build(k);
    } catch (const std::bad_alloc&) {
      per[k].ok = false;
    }
  });
  out.header.clear(), out.cells.clear(), out.entries.clear();
  for (PerLight& pl : per) {
    if (!pl.ok) return false;
    pl.header.cell_base = (uint32_t)out.cells.size();
    const uint32_t entry_base = (uint32_t)out.entries.size();
    for (uint2& c : pl.cells) c.x += entry_base;
    out.header.push_back(pl.header);
    out.cells.insert(out.cells.end(), pl.cells.begin(), pl.cells.end());
    out.entries.insert(out.entries.end(), pl.entries.begin(), pl.entries.end());
  }
  return true;
}

static int flatten_scene_checked(const rtk_scene* scene, rtk::FlatScene& flat) {
  if ((scene->n_objects && !scene->objects) || (scene->n_lights && !scene->lights))
    return fail(RTK_ERR_INVALID, "rtk_scene_upload: null object or light array");
  if (scene->max_depth > RTK_MAX_DEPTH_LIMIT) return fail(RTK_ERR_UNSUPPORTED, "rtk_scene_upload: max_depth above RTK_MAX_DEPTH_LIMIT");
  if (scene->n_lights > RTK_MAX_LIGHTS) return fail(RTK_ERR_UNSUPPORTED, "rtk_scene_upload: more than RTK_MAX_LIGHTS (32,767) lights");

  using namespace rtk;
  // Upper bound of the staged bytes (every piece rounded up to 256): the object array is built in
  // place at its start, the smaller arrays are copied behind it once their sizes are known.
  bool any_pattern = false;
  for (uint32_t i = 0; i < scene->n_objects && !any_pattern; ++i) any_pattern = scene->objects[i].material.pattern_transformed != 0;
  size_t n_planes_in = 0;
  for (uint32_t i = 0; i < scene->n_objects; ++i) n_planes_in += scene->objects[i].kind == RTK_PLANE;
  const size_t n_all = scene->n_objects, n_padded = n_all + kBundle, n_l = scene->n_lights;
  const size_t bound = 20 * 256 + n_planes_in * n_l * sizeof(double2) + n_all * sizeof(DevObject) + n_l * (sizeof(DevLight) + sizeof(float4) + sizeof(uint32_t)) +
                       n_padded * (3 * sizeof(float4) + sizeof(uint32_t) + sizeof(SphRecord)) + n_all * (sizeof(double4) + sizeof(uint32_t)) +
                       (any_pattern ? n_all * 12 * sizeof(double) : 0) + ((n_l + 3) / 4) * n_padded * sizeof(float4) +
                       2 * (n_padded / kBundle + kBundle) * sizeof(float4) + (n_padded / kBundle + kBundle) * sizeof(unsigned long long);
  if (flat.grow) {
    flat.staged = flat.grow(flat.user, bound);
  } else {
    flat.own.reset(new unsigned char[bound]);
    flat.staged = flat.own.get();
  }
  if (!flat.staged) return fail(RTK_ERR_NOMEM, "rtk_scene_upload: no staging memory");
  DevObject* const objects = reinterpret_cast<DevObject*>(flat.staged);
  std::vector<uint32_t> plane_object;
  std::vector<double> pattern_inv;  // stays empty unless some pattern has a transform of its own
  bool track_n1 = false, any_secondary = false;
  float r2_max = 0.f, r_max = 0.f, c_max = 0.f;

  // Per-object work (validation, the reference's cofactor inverse, bounding ball) is independent:
  // it is split over the library's host threads.  Errors are reported for the lowest object index.
  if (any_pattern) pattern_inv.assign(12 * (size_t)scene->n_objects, 0.0);
  std::vector<double4> ball(scene->n_objects);
  struct Outcome {
    int status = RTK_OK;
    uint32_t at = 0xffffffffu;
    const char* message = "";
    bool track_n1 = false, any_secondary = false;
  };
  auto flatten = [&](uint32_t begin, uint32_t end_, Outcome& oc) {
    auto bad = [&](uint32_t i, int status, const char* message) {
      oc.status = status, oc.at = i, oc.message = message;
    };
    for (uint32_t i = begin; i < end_; ++i) {
      const rtk_object& src = scene->objects[i];
      if (src.kind != RTK_SPHERE && src.kind != RTK_PLANE) return bad(i, RTK_ERR_INVALID, "rtk_scene_upload: unknown object kind");
      if (src.material.texture_kind > RTK_TEX_STRIPED) return bad(i, RTK_ERR_INVALID, "rtk_scene_upload: unknown texture kind");
      for (double v : src.transform)
        if (!std::isfinite(v))  // the reference panics on the NaN distances such a transform produces (src/scene.rs:164-166)
          return bad(i, RTK_ERR_INVALID, "rtk_scene_upload: object transform has a non-finite element");
      if (!is_affine(src.transform))
        return bad(i, RTK_ERR_UNSUPPORTED, "rtk_scene_upload: object transform is not affine (last row must be 0 0 0 1)");
      if (src.kind == RTK_PLANE && src.plane_normal[3] != 0.0)
        return bad(i, RTK_ERR_UNSUPPORTED, "rtk_scene_upload: plane_normal[3] must be 0");
      M4 t = m4_load(src.transform), inv;
      bool invertible = m4_invert(t, inv);
      if (!invertible) {  // src/scene.rs:55-62: identity inverse
        inv = m4_identity();
        t = m4_identity();
      }
      DevObject& o = objects[i];
      std::memcpy(o.inv, inv.e, sizeof o.inv);
      std::memcpy(o.color_a, src.material.color_a, sizeof o.color_a);
      std::memcpy(o.color_b, src.material.color_b, sizeof o.color_b);
      o.ambient = src.material.ambient;
      o.diffuse = src.material.diffuse;
      o.specular = src.material.specular;
      o.shininess = src.material.shininess;
      o.transparency = src.material.transparency;
      o.reflectivity = src.material.reflectivity;
      o.refractivity = src.material.refractivity;
      std::memcpy(o.plane_normal, src.plane_normal, sizeof o.plane_normal);
      o.kind = src.kind;
      o.texture_kind = src.material.texture_kind;
      if (src.material.pattern_transformed) {
        // `TransformPattern` (src/graphics/patterns.rs:27-35): the reference inverts per sample and
        // samples the untransformed point when the matrix is singular
        M4 pinv;
        if (m4_invert(m4_load(src.material.pattern_transform), pinv)) {
          std::memcpy(&pattern_inv[12 * (size_t)i], pinv.e, 12 * sizeof(double));
          o.texture_kind |= 0x80000000u;
        }
      }
      if (o.refractivity != 1.0) oc.track_n1 = true;
      if (!(std::fabs(o.reflectivity) < 0.00001) || !(std::fabs(o.transparency) < 0.00001)) oc.any_secondary = true;
      // bounding ball of the transformed unit sphere: centre = translation, radius = sigma_max
      if (src.kind == RTK_SPHERE) ball[i] = make_double4(t.at(0, 3), t.at(1, 3), t.at(2, 3), sigma_max3(t));
    }
  };
  // chunks of 512 objects, handed to the pool's threads as they come free
  const uint32_t chunk = 512u, n_chunks = (scene->n_objects + chunk - 1u) / chunk;
  std::vector<Outcome> outcomes(n_chunks);
  HostPool::get().run(n_chunks, [&](uint32_t c) {
    flatten(c * chunk, std::min((c + 1u) * chunk, scene->n_objects), outcomes[c]);
  });
  for (const Outcome& oc : outcomes) {  // chunks are in object order: the first failure is the lowest index
    if (oc.status != RTK_OK) return fail(oc.status, oc.message);
    track_n1 = track_n1 || oc.track_n1;
    any_secondary = any_secondary || oc.any_secondary;
  }
  // spheres (with the bounding balls of their images) and planes, in object order
  std::vector<SphereKey> keys;
  keys.reserve(n_all);
  for (uint32_t i = 0; i < scene->n_objects; ++i) {
    if (objects[i].kind == RTK_PLANE)
      plane_object.push_back(i);
    else
      keys.push_back({{ball[i].x, ball[i].y, ball[i].z}, ball[i].w, i});
  }
  // Sweep order of the spheres: spatial clusters of kBundle (a k-d split down to kBundle
  // members), so that the wavefront-level tests can skip whole clusters.  Closest-hit ties
  // are decided by object index (closest_consider), not by sweep order.
  const uint32_t n_sph = (uint32_t)keys.size();
  cluster_order(keys.data(), n_sph);

  // the FP32 records are recentred on the middle of the sphere centres' bounding box
  const uint32_t n_pad = round_up_to(n_sph, kBundle);
  double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
  for (uint32_t i = 0; i < n_sph; ++i)
    for (int a = 0; a < 3; ++a) {
      if (i == 0 || keys[i].c[a] < lo[a]) lo[a] = keys[i].c[a];
      if (i == 0 || keys[i].c[a] > hi[a]) hi[a] = keys[i].c[a];
    }
  const double center[3] = {0.5 * (lo[0] + hi[0]), 0.5 * (lo[1] + hi[1]), 0.5 * (lo[2] + hi[2])};

  // Lights with w == 0 (what scene files produce, src/loader.rs:35-40): the shadow-ray line
  // direction G = L.xyz does not depend on the shading point, so the per-(sphere, light)
  // part of the filter, K = c.g - R'^2 rho, is tabulated here.
  // Lights with w == 1 (built from points): plain segments from the shading point to the light;
  // the bundle of a patch is a cone with its apex at the light.  Their record is the light's
  // position, recentred like the sphere records.  lights_bundled: every light is one or the other.
  bool lights_uniform = scene->n_lights > 0, lights_bundled = scene->n_lights > 0;
  std::vector<double> gd(4 * (size_t)scene->n_lights);
  for (uint32_t k = 0; k < scene->n_lights; ++k) {
    const double* L = scene->lights[k].position;
    const double gg = L[0] * L[0] + L[1] * L[1] + L[2] * L[2];
    if (L[3] == 1.0 && std::isfinite(gg)) {
      lights_uniform = false;
      continue;
    }
    if (L[3] != 0.0 || !(gg > 0.0) || !std::isfinite(gg)) {
      lights_uniform = false;
      lights_bundled = false;
      continue;
    }
    const double inv = 1.0 / std::sqrt(gg);
    gd[4 * k + 0] = L[0] * inv;
    gd[4 * k + 1] = L[1] * inv;
    gd[4 * k + 2] = L[2] * inv;
    gd[4 * k + 3] = -inv;  // rho = (w - 1)/|G|
  }
  const uint32_t k_width = 4u;  // lights per table row and per sweep batch
  const bool tabulated = lights_uniform && n_pad;
  const uint32_t batches = (scene->n_lights + k_width - 1) / k_width;
  // light grids for the shadow rays of incoherent shading points: only where rays recurse, the
  // spheres are many enough for a cluster hierarchy, and every light has w = 0
  LightGrids grids;
  const bool shadow_grid = any_secondary && lights_uniform && n_sph >= 256u &&
                           build_light_grids(keys.data(), n_sph, center, gd, scene->n_lights, grids);

  // One device allocation and one copy for all record arrays (each 256-byte aligned inside it):
  // the arrays are laid out first and then written where they will be copied from.  The
  // per-camera records, which only kernels write, follow at the end of the allocation.
  const uint32_t n_clusters = n_pad / kBundle, n_clusters_pad = round_up_to(n_clusters, kBundle);
  size_t total = 0;
  auto reserve = [&](void* rtk_scene_handle::*dst, size_t bytes) -> size_t {
    const size_t at = total;
    if (bytes == 0) return at;
    flat.pieces.push_back({dst, bytes, total});
    total += (bytes + 255u) & ~(size_t)255u;
    return at;
  };
  const size_t at_objects = reserve(&rtk_scene_handle::d_objects, (size_t)scene->n_objects * sizeof(DevObject));  // already in place
  const size_t at_lights = reserve(&rtk_scene_handle::d_lights, (size_t)scene->n_lights * sizeof(DevLight));
  const size_t at_filter = reserve(&rtk_scene_handle::d_filter, (size_t)n_pad * sizeof(float4));
  const size_t at_exact = reserve(&rtk_scene_handle::d_exact, (size_t)n_sph * sizeof(double4));
  const size_t at_sph_object = reserve(&rtk_scene_handle::d_sph_object, (size_t)n_pad * sizeof(uint32_t));
  const size_t at_sph_record = reserve(&rtk_scene_handle::d_sph_record, (size_t)n_pad * sizeof(SphRecord));
  const size_t at_plane_object = reserve(&rtk_scene_handle::d_plane_object, plane_object.size() * sizeof(uint32_t));
  const size_t at_plane_light = reserve(&rtk_scene_handle::d_plane_light, plane_object.size() * scene->n_lights * sizeof(double2));
  const size_t at_pattern_inv = reserve(&rtk_scene_handle::d_pattern_inv, pattern_inv.size() * sizeof(double));
  const size_t at_light_filter = reserve(&rtk_scene_handle::d_light_filter, (size_t)scene->n_lights * sizeof(float4));
  const size_t at_light_kind = reserve(&rtk_scene_handle::d_light_kind, (size_t)scene->n_lights * sizeof(uint32_t));
  const size_t at_k_table = reserve(&rtk_scene_handle::d_k_table, tabulated ? (size_t)batches * n_pad * k_width * sizeof(float) : 0);
  const size_t at_sph_tab = reserve(&rtk_scene_handle::d_sph_tab, tabulated ? (size_t)n_pad * sizeof(float4) : 0);
  const size_t at_sph_ball = reserve(&rtk_scene_handle::d_sph_ball, (size_t)n_pad * sizeof(float4));
  const size_t at_cluster_ball = reserve(&rtk_scene_handle::d_cluster_ball, n_pad ? (size_t)n_clusters_pad * sizeof(float4) : 0);
  const size_t at_cluster_plain = reserve(&rtk_scene_handle::d_cluster_plain, n_pad ? (size_t)n_clusters_pad * sizeof(unsigned long long) : 0);
  const uint32_t n_supers = (n_clusters + kSuper - 1) / kSuper;
  const size_t at_super_ball = reserve(&rtk_scene_handle::d_super_ball, (size_t)n_supers * sizeof(float4));
  const size_t copied = total;
  // built apart from the staged bytes (their size is not known when the staging is sized), copied behind them
  const size_t extra_begin = total;
  size_t at_light_grid = 0, at_grid_cells = 0, at_grid_entries = 0;
  if (shadow_grid) {
    at_light_grid = reserve(&rtk_scene_handle::d_light_grid, grids.header.size() * sizeof(rtk::LightGrid)) - extra_begin;
    at_grid_cells = reserve(&rtk_scene_handle::d_grid_cells, grids.cells.size() * sizeof(uint2)) - extra_begin;
    at_grid_entries = reserve(&rtk_scene_handle::d_grid_entries, grids.entries.size() * sizeof(uint32_t)) - extra_begin;
    flat.extra.assign(total - extra_begin, 0);
    std::memcpy(flat.extra.data() + at_light_grid, grids.header.data(), grids.header.size() * sizeof(rtk::LightGrid));
    std::memcpy(flat.extra.data() + at_grid_cells, grids.cells.data(), grids.cells.size() * sizeof(uint2));
    if (!grids.entries.empty())
      std::memcpy(flat.extra.data() + at_grid_entries, grids.entries.data(), grids.entries.size() * sizeof(uint32_t));
  }
  reserve(&rtk_scene_handle::d_prim, (size_t)n_pad * sizeof(float4));
  reserve(&rtk_scene_handle::d_prim_ball, (size_t)n_pad * sizeof(float4));
  reserve(&rtk_scene_handle::d_prim_cluster, n_pad ? (size_t)n_clusters_pad * sizeof(float4) : 0);
  flat.total = total;
  flat.staged_bytes = copied;
  if (at_objects != 0 || copied > bound) return fail(RTK_ERR_INVALID, "rtk_scene_upload: staging bound exceeded");  // cannot happen
  unsigned char* const base = flat.staged;
  DevLight* const lights = reinterpret_cast<DevLight*>(base + at_lights);
  float4* const filter = reinterpret_cast<float4*>(base + at_filter);
  double4* const sph_exact = reinterpret_cast<double4*>(base + at_exact);
  uint32_t* const sph_object = reinterpret_cast<uint32_t*>(base + at_sph_object);
  SphRecord* const sph_record = reinterpret_cast<SphRecord*>(base + at_sph_record);
  float4* const light_filter = reinterpret_cast<float4*>(base + at_light_filter);
  uint32_t* const light_kind = reinterpret_cast<uint32_t*>(base + at_light_kind);
  float* const k_table = reinterpret_cast<float*>(base + at_k_table);
  float4* const sph_tab = reinterpret_cast<float4*>(base + at_sph_tab);
  float4* const sph_ball = reinterpret_cast<float4*>(base + at_sph_ball);
  float4* const cluster_ball = reinterpret_cast<float4*>(base + at_cluster_ball);
  float4* const super_ball = reinterpret_cast<float4*>(base + at_super_ball);
  unsigned long long* const cluster_plain = reinterpret_cast<unsigned long long*>(base + at_cluster_plain);
  if (!plane_object.empty()) std::memcpy(base + at_plane_object, plane_object.data(), plane_object.size() * sizeof(uint32_t));
  if (!pattern_inv.empty()) std::memcpy(base + at_pattern_inv, pattern_inv.data(), pattern_inv.size() * sizeof(double));

  for (uint32_t k = 0; k < scene->n_lights; ++k) {
    std::memcpy(lights[k].pos, scene->lights[k].position, sizeof lights[k].pos);
    std::memcpy(lights[k].intensity, scene->lights[k].intensity, sizeof lights[k].intensity);
    lights[k].reserved_ = 0.0;
    const double* L = scene->lights[k].position;
    const double gg = L[0] * L[0] + L[1] * L[1] + L[2] * L[2];
    if (L[3] == 1.0 && std::isfinite(gg)) {
      light_kind[k] = 1u;
      light_filter[k] = make_float4((float)(L[0] - center[0]), (float)(L[1] - center[1]), (float)(L[2] - center[2]), 0.f);
    } else if (L[3] != 0.0 || !(gg > 0.0) || !std::isfinite(gg)) {
      light_kind[k] = 2u;
      light_filter[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
      light_kind[k] = 0u;
      light_filter[k] = make_float4((float)gd[4 * k], (float)gd[4 * k + 1], (float)gd[4 * k + 2], (float)gd[4 * k + 3]);
    }
  }

  // plane_may_block's per-(plane, light) part: row 1 of the plane's inverse applied to (L.xyz, L.w - 1)
  // and 1e-12 times the magnitudes of its terms (non-finite values make the test keep the pair)
  double2* const plane_light = reinterpret_cast<double2*>(base + at_plane_light);
  for (size_t ip = 0; ip < plane_object.size(); ++ip) {
    const double* m = objects[plane_object[ip]].inv;
    for (uint32_t k = 0; k < scene->n_lights; ++k) {
      const double* L = scene->lights[k].position;
      const double a = m[4] * L[0], b = m[5] * L[1], c = m[6] * L[2], d = m[7] * (L[3] - 1.0);
      plane_light[ip * scene->n_lights + k] = make_double2(a + b + c + d, 1.0e-12 * (std::fabs(a) + std::fabs(b) + std::fabs(c) + std::fabs(d)));
    }
  }

  // Every record of a sphere follows from its key alone, so the clusters are shared out among a few
  // host threads; each keeps the maxima the kernels' error bounds need.
  struct Maxima {
    float r2 = 0.f, r = 0.f, c = 0.f;
  };
  const double kappa = (double)(1.0f - 24.0f * rtk::kEps32);
  auto tabulate = [&](uint32_t j) {  // the per-(sphere, light) tables of sphere slot j, from its filter record
    const float4 f = filter[j];
    const double cc = kappa * ((double)f.x * f.x + (double)f.y * f.y + (double)f.z * f.z) - (double)f.w;
    float ccf = (float)cc;
    if ((double)ccf > cc) ccf = std::nextafterf(ccf, -INFINITY);
    sph_tab[j] = make_float4(f.x, f.y, f.z, std::nextafterf(ccf, -INFINITY));
    for (uint32_t b = 0; b < batches; ++b)
      for (uint32_t e = 0; e < k_width; ++e) {
        const uint32_t k = std::min(b * k_width + e, scene->n_lights - 1);  // pad a short batch with its last light
        const double v = (double)f.x * gd[4 * k] + (double)f.y * gd[4 * k + 1] + (double)f.z * gd[4 * k + 2] - (double)f.w * gd[4 * k + 3];
        k_table[((size_t)b * n_pad + j) * k_width + e] = (float)v;
      }
  };
  auto exact_record = [&](uint32_t j, uint32_t obj) {  // exact-test record of sweep slot j
    const double* m = objects[obj].inv;
    SphRecord& r = sph_record[j];
    r.scale[0] = m[0], r.scale[1] = m[5], r.scale[2] = m[10];
    r.shift[0] = m[3], r.shift[1] = m[7], r.shift[2] = m[11];
    r.object = obj;
    r.diagonal = (m[1] == 0.0 && m[2] == 0.0 && m[4] == 0.0 && m[6] == 0.0 && m[8] == 0.0 && m[9] == 0.0) ? 1u : 0u;
    r.reserved_ = 0;
  };
  auto cluster_records = [&](uint32_t c_begin, uint32_t c_end, Maxima& mx) {
    for (uint32_t c = c_begin; c < c_end; ++c) {
      const uint32_t j0 = c * kBundle, members = std::min<uint32_t>(kBundle, n_sph - j0);
      for (uint32_t j = j0; j < j0 + members; ++j) {
        const SphereKey& key = keys[j];
        sph_exact[j] = make_double4(key.c[0], key.c[1], key.c[2], key.r);
        sph_object[j] = key.object;
        // FP32 filter record: the rounded centre, and a radius that covers the rounding
        const double cd[3] = {key.c[0] - center[0], key.c[1] - center[1], key.c[2] - center[2]};
        const float cf[3] = {(float)cd[0], (float)cd[1], (float)cd[2]};
        const double dx = cd[0] - cf[0], dy = cd[1] - cf[1], dz = cd[2] - cf[2];
        const double shift = std::sqrt(dx * dx + dy * dy + dz * dz);  // actual displacement of the rounded centre
        const double R = key.r * (1.0 + 1e-9) + shift * 1.0001;
        const float r2 = round_up(R * R);
        filter[j] = make_float4(cf[0], cf[1], cf[2], r2);
        mx.r2 = std::fmax(mx.r2, r2);
        mx.r = std::fmax(mx.r, round_up(R));
        mx.c = std::fmax(mx.c, round_up(std::sqrt((double)cf[0] * cf[0] + (double)cf[1] * cf[1] + (double)cf[2] * cf[2]) * 1.0001));
        exact_record(j, key.object);
        // record of the wavefront-level test: centre and (not squared) inflated radius
        sph_ball[j] = make_float4(cf[0], cf[1], cf[2], round_up(std::sqrt((double)r2) * (1.0 + 1e-7)));
        if (tabulated) tabulate(j);
      }
      // Slots behind the last sphere repeat its records (and its object index): a repeated candidate
      // is re-evaluated to the same roots, which changes no closest-hit or shadow decision.  Their
      // wavefront-level ball is empty.
      for (uint32_t j = j0 + members; j < j0 + kBundle; ++j) {
        filter[j] = filter[n_sph - 1];
        sph_object[j] = sph_object[n_sph - 1];
        exact_record(j, sph_object[j]);
        sph_ball[j] = make_float4(0.f, 0.f, 0.f, -1.0e30f);
        if (tabulated) tabulate(j);
      }
      // members whose exact test is that of a sphere of one radius: what the kernels' "sphere behind the
      // shadow ray" test may be applied to
      unsigned long long plain = 0ull;
      for (uint32_t j = j0; j < j0 + members; ++j) {
        const SphRecord& r = sph_record[j];
        if (r.diagonal && r.scale[0] == r.scale[1] && r.scale[1] == r.scale[2] && r.scale[0] != 0.0 && std::isfinite(r.scale[0]))
          plain |= 1ull << (j - j0);
      }
      cluster_plain[c] = plain;
      // one ball per cluster, around the balls of its members
      double m[3] = {0, 0, 0};
      for (uint32_t j = j0; j < j0 + members; ++j) m[0] += sph_ball[j].x, m[1] += sph_ball[j].y, m[2] += sph_ball[j].z;
      const float mf[3] = {(float)(m[0] / members), (float)(m[1] / members), (float)(m[2] / members)};
      double reach = 0.0;
      for (uint32_t j = j0; j < j0 + members; ++j) {
        const double dx = (double)sph_ball[j].x - mf[0], dy = (double)sph_ball[j].y - mf[1], dz = (double)sph_ball[j].z - mf[2];
        reach = std::fmax(reach, std::sqrt(dx * dx + dy * dy + dz * dz) + (double)sph_ball[j].w);
      }
      cluster_ball[c] = make_float4(mf[0], mf[1], mf[2], round_up(reach * (1.0 + 1e-7)));
    }
  };
  for (uint32_t c = n_clusters; c < n_clusters_pad; ++c) cluster_ball[c] = make_float4(0.f, 0.f, 0.f, -1.0e30f), cluster_plain[c] = 0ull;
  {
    // runs of 8 clusters, handed to the pool's threads as they come free; the last cluster (the only
    // one that may have empty slots, which read the last sphere's records) runs here, after the others
    const uint32_t n_parallel = n_clusters ? n_clusters - 1 : 0;
    const uint32_t run = 8u, n_runs = (n_parallel + run - 1u) / run;
    std::vector<Maxima> maxima(std::max(1u, n_runs));
    HostPool::get().run(n_runs, [&](uint32_t r) { cluster_records(r * run, std::min((r + 1u) * run, n_parallel), maxima[r]); });
    if (n_clusters) cluster_records(n_clusters - 1, n_clusters, maxima[0]);
    for (const Maxima& mx : maxima) r2_max = std::fmax(r2_max, mx.r2), r_max = std::fmax(r_max, mx.r), c_max = std::fmax(c_max, mx.c);
  }
  // upper level: one ball around the balls of kSuper consecutive clusters (neighbours in the k-d split)
  for (uint32_t u = 0; u < n_supers; ++u) {
    const uint32_t c0 = u * kSuper, c1 = std::min(c0 + (uint32_t)kSuper, n_clusters);
    double m[3] = {0, 0, 0};
    for (uint32_t c = c0; c < c1; ++c) m[0] += cluster_ball[c].x, m[1] += cluster_ball[c].y, m[2] += cluster_ball[c].z;
    const float mf[3] = {(float)(m[0] / (c1 - c0)), (float)(m[1] / (c1 - c0)), (float)(m[2] / (c1 - c0))};
    double reach = 0.0;
    for (uint32_t c = c0; c < c1; ++c) {
      const double dx = (double)cluster_ball[c].x - mf[0], dy = (double)cluster_ball[c].y - mf[1], dz = (double)cluster_ball[c].z - mf[2];
      reach = std::fmax(reach, std::sqrt(dx * dx + dy * dy + dz * dz) + (double)cluster_ball[c].w);
    }
    super_ball[u] = make_float4(mf[0], mf[1], mf[2], round_up(reach * (1.0 + 1e-7)));
  }

  DevScene& d = flat.dev;
  d = DevScene{};
  d.n_clusters_padded = n_clusters_pad;
  d.lights_uniform = lights_uniform ? 1u : 0u;
  d.lights_bundled = lights_bundled ? 1u : 0u;
  d.k_width = k_width;
  d.shadow_grid = shadow_grid ? 1u : 0u;
  d.n_objects = scene->n_objects;
  d.n_lights = scene->n_lights;
  d.n_spheres = n_sph;
  d.n_spheres_padded = n_pad;
  d.n_spheres_grouped = round_up_to(n_sph, kGroup);
  d.n_planes = (uint32_t)plane_object.size();
  d.max_depth = scene->max_depth;
  d.track_n1 = track_n1 ? 1u : 0u;
  d.any_secondary = any_secondary ? 1u : 0u;
  std::memcpy(d.center, center, sizeof center);
  std::memcpy(d.ambient, scene->ambient_color, sizeof d.ambient);
  d.r2_max = r2_max;
  d.r_max = r_max;
  d.c_max = c_max;
  return RTK_OK;
}

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
  *out = h;
  return RTK_OK;
}

static int make_dev_camera(const rtk_camera* camera, rtk::DevCamera& c) {
  if (camera->width == 0 || camera->height == 0) return fail(RTK_ERR_INVALID, "camera has zero size");
  if (!is_affine(camera->transform))
    return fail(RTK_ERR_UNSUPPORTED, "camera transform is not affine (last row must be 0 0 0 1)");
  M4 inv;
  if (!m4_invert(m4_load(camera->transform), inv))
    return fail(RTK_ERR_SINGULAR_CAMERA, "Failed to invert camera transform");  // src/scene/cameras.rs:59-62
  c.width = camera->width;
  c.height = camera->height;
  c.half_width = camera->half_width;
  c.half_height = camera->half_height;
  c.pixel_size = camera->pixel_size;
  std::memcpy(c.inv, inv.e, sizeof c.inv);
  // inverse * point(0, 0, 0): the 4-term row sums collapse to the last column
  for (int r = 0; r < 3; ++r) c.origin[r] = inv.at(r, 0) * 0.0 + inv.at(r, 1) * 0.0 + inv.at(r, 2) * 0.0 + inv.at(r, 3) * 1.0;
  return RTK_OK;
}

// ---------------------------------------------------------------- partitions
static uint32_t tiles_across(uint32_t width) { return (width + RTK_TILE_W - 1) / RTK_TILE_W; }
static uint32_t tiles_down(uint32_t height) { return (height + RTK_TILE_H - 1) / RTK_TILE_H; }

int rtk::partition_from_tiling(const rtk_camera* camera, const rtk_tiling* tiling, rtk::Partition& p, const char* who) {
  rtk_tiling tl = {0u, 1u};
  if (tiling) tl = *tiling;
  if (tl.world == 0 || tl.rank >= tl.world) return fail(RTK_ERR_INVALID, std::string(who) + ": bad tiling (rank >= world)");
  p = rtk::Partition{};
  p.layout = tl.world > 1 ? 1u : 0u;
  p.step = tl.world;
  p.offset = tl.rank;
  p.n_local_tiles = rtk_tiles_for_rank(camera->width, camera->height, tl.rank, tl.world);
  return RTK_OK;
}

// Of `rows` tile rows, those in k * period + [offset, offset + width), k = 0, 1, ...
uint64_t rtk::rows_of_share(uint32_t rows, uint32_t period, uint32_t offset, uint32_t width) {
  if (period == 0 || width == 0) return 0;
  const uint64_t whole = rows / period, rest = rows % period;
  return whole * width + (rest > offset ? std::min<uint64_t>(width, rest - offset) : 0u);
}
// Tile rows of [first, first + n) that the bands dealt to `rank` hold.
static uint64_t band_rows_of_rank(uint32_t rows, uint32_t m, uint32_t rank, uint32_t world) {
  const uint64_t period = (uint64_t)m * world, offset = (uint64_t)m * rank;
  if (offset >= rows) return 0;
  if (period >= rows) return std::min<uint64_t>(m, rows - offset);  // a single round
  return rtk::rows_of_share(rows, (uint32_t)period, (uint32_t)offset, m);
}
static bool banding_rows(uint32_t height, const rtk_banding* bd, uint32_t& rows) {
  if (!bd || bd->world == 0 || bd->rank >= bd->world || bd->band_tile_rows == 0) return false;
  const uint32_t ty = tiles_down(height);
  if (bd->first_tile_row > ty) return false;
  rows = ty - bd->first_tile_row;
  if (bd->n_tile_rows) {
    if (bd->n_tile_rows > rows) return false;
    rows = bd->n_tile_rows;
  }
  return true;
}

int rtk::partition_from_banding(const rtk_camera* camera, const rtk_banding* banding, rtk::Partition& p, const char* who) {
  uint32_t rows = 0;
  if (!banding_rows(camera->height, banding, rows))
    return fail(RTK_ERR_INVALID, std::string(who) + ": bad banding (rank >= world, band_tile_rows == 0, or rows outside the frame)");
  p = rtk::Partition{};
  p.layout = banding->compact ? 2u : 0u;
  // a period or an offset beyond 32 bits lies beyond the frame: one round, or no rows at all
  const uint64_t period = (uint64_t)banding->band_tile_rows * banding->world, offset = (uint64_t)banding->band_tile_rows * banding->rank;
  p.band_rows = banding->band_tile_rows;
  p.band_period = (uint32_t)std::min<uint64_t>(period, 0xffffffffull);
  p.band_offset = offset > 0xffffffffull ? 0u : (uint32_t)offset;
  p.row0 = banding->first_tile_row;
  p.n_local_tiles = band_rows_of_rank(rows, banding->band_tile_rows, banding->rank, banding->world) * tiles_across(camera->width);
  return RTK_OK;
}

int rtk::partition_from_rows(const rtk_camera* camera, uint32_t first, uint32_t period, uint32_t offset, uint32_t width,
                             bool compact, rtk::Partition& p, const char* who) {
  const uint32_t ty = tiles_down(camera->height);
  if (first > ty || width == 0 || period == 0 || (uint64_t)offset + width > period)
    return fail(RTK_ERR_INVALID, std::string(who) + ": bad share of the frame's tile rows");
  p = rtk::Partition{};
  p.layout = compact ? 2u : 0u;
  p.band_rows = width;
  p.band_period = period;
  p.band_offset = offset;
  p.row0 = first;
  p.n_local_tiles = rtk::rows_of_share(ty - first, period, offset, width) * tiles_across(camera->width);
  return RTK_OK;
}

extern "C" {

uint64_t rtk_band_tile_rows(uint32_t height, const rtk_banding* banding) {
  uint32_t rows = 0;
  if (height == 0 || !banding_rows(height, banding, rows)) return 0;
  return band_rows_of_rank(rows, banding->band_tile_rows, banding->rank, banding->world);
}

uint64_t rtk_band_pixel_rows(uint32_t height, const rtk_banding* banding) {
  return rtk_band_tile_rows(height, banding) * (uint64_t)RTK_TILE_H;
}

}  // extern "C"

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
  const size_t want = std::min<size_t>(std::max<size_t>(pixels * w.nodes_per_pixel, floor_nodes), ctx->wavefront_max_nodes);
  if (want < floor_nodes) return nullptr;
  if (w.cap_nodes >= want) return &w;
  // an earlier frame's kernels may still use the old buffers
  if (hipDeviceSynchronize() != hipSuccess) return nullptr;
  if (w.rays) (void)hipFree(w.rays);
  if (w.outs) (void)hipFree(w.outs);
  w.rays = nullptr, w.outs = nullptr, w.cap_nodes = 0;
  if (!w.ctrl && hipMalloc((void**)&w.ctrl, sizeof(unsigned int) * rtk::kWavefrontCtrlWords) != hipSuccess) w.ctrl = nullptr;
  if (!w.counters && hipMalloc((void**)&w.counters, sizeof(rtk::DevCounters) * rtk::kCounterShards) != hipSuccess) w.counters = nullptr;
  if (!w.ctrl || !w.counters || hipMalloc((void**)&w.rays, want * sizeof(rtk::WfNodeRay)) != hipSuccess ||
      hipMalloc((void**)&w.outs, want * sizeof(rtk::WfNodeOut)) != hipSuccess) {
    (void)hipGetLastError();
    if (w.rays) (void)hipFree(w.rays);
    w.rays = nullptr, w.outs = nullptr;
    return nullptr;
  }
  w.cap_nodes = (uint32_t)want;
  return &w;
}

// Orders this call behind the context's previous one (they share per-camera records, counters
// and hit records), resets the counters, rebuilds the per-camera records and starts the timing.
static int frame_begin(Frame& f, rtk_context* ctx, rtk_scene_handle* scene, const rtk_camera* camera, void* hip_stream,
                       const Scratch& scr) {
  f.ctx = ctx, f.scene = scene, f.camera = camera, f.scr = scr;
  const int rc = make_dev_camera(camera, f.cam);
  if (rc != RTK_OK) return rc;
  RTK_HIP(hipSetDevice(ctx->device));
  f.st = hip_stream ? (hipStream_t)hip_stream : ctx->stream;
  f.opaque = !scene->dev.any_secondary;
  if (ctx->n_timed && ctx->last_stream != f.st)
    RTK_HIP(hipStreamWaitEvent(f.st, ctx->ev_stop[(ctx->n_timed - 1) % RTK_TIMING_HISTORY], 0));
  RTK_HIP(rtk::launch_prep_primary(f.st, scene->dev, f.cam, f.scr.prim, f.scr.prim_ball, f.scr.prim_cluster, f.scr.counters));
  f.slot = (int)(ctx->n_timed % RTK_TIMING_HISTORY);
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
  const bool by_level = ctx->trace_order == RTK_TRACE_BY_LEVEL ||
                        (ctx->trace_order == RTK_TRACE_AUTO && rtk::wavefront_applies(ds) && rtk::wavefront_pays(ds, launch_pixels));
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

int rtk_camera_supersample(const rtk_camera* camera, uint32_t s, rtk_camera* out) {
  if (!camera || !out) return fail(RTK_ERR_INVALID, "rtk_camera_supersample: null argument");
  if (s == 0 || s > RTK_MAX_SAMPLES_PER_AXIS)
    return fail(RTK_ERR_INVALID, "rtk_camera_supersample: samples_per_axis must be 1.." + std::to_string(RTK_MAX_SAMPLES_PER_AXIS));
  if ((uint64_t)camera->width * s > 0xffffffffull || (uint64_t)camera->height * s > 0xffffffffull)
    return fail(RTK_ERR_UNSUPPORTED, "rtk_camera_supersample: sample frame wider or taller than 2^32 - 1");
  rtk_camera c = *camera;
  c.width = camera->width * s;
  c.height = camera->height * s;
  c.pixel_size = camera->pixel_size / (double)s;
  *out = c;
  return RTK_OK;
}

uint64_t rtk_compact_pixels_sampled(uint32_t width, uint32_t height, uint32_t s, uint32_t rank, uint32_t world) {
  if (s == 0 || s > RTK_MAX_SAMPLES_PER_AXIS || (uint64_t)width * s > 0xffffffffull || (uint64_t)height * s > 0xffffffffull)
    return 0;
  if (world <= 1) return rank == 0 && world == 1 ? (uint64_t)width * height : 0;
  if (8u % s != 0) return 0;
  return rtk_tiles_for_rank(width * s, height * s, rank, world) * (uint64_t)(RTK_TILE_W * RTK_TILE_H / (s * s));
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
  if (ctx->cached_handle && c.n_objects == scene->n_objects && c.n_lights == scene->n_lights && c.max_depth == scene->max_depth &&
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
