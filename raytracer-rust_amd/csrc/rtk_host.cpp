// rtk_host.cpp -- the host half of the C ABI of include/rtk.h that needs no GPU: error text, the f64
// matrix arithmetic of the reference (cofactor inverse, look_at), validation and flattening of a scene
// into the device's record arrays (sweep order, bounding balls, filter tables, light grids), the camera
// the kernels take, and the arithmetic of tiles, bands and shares.  Plain C++: this file includes no HIP
// runtime header and is compiled by the host compiler, so it also builds into the sanitizer drivers
// (make asan / tsan) that run it without a device.  Everything that touches the device is in rtk_api.hip.

#include "rtk_host.h"

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <functional>
#include <mutex>
#include <new>
#include <string>
#include <system_error>
#include <thread>
#include <utility>
#include <vector>

#include <pthread.h>

// The f64 matrix code below (cofactor inverse, look_at) must round like the reference's: no fused
// multiply-adds, whatever the host architecture or -march offers (the file is built with
// -ffp-contract=off as well).
#if defined(__clang__)
#pragma clang fp contract(off)
#endif

namespace {
thread_local std::string g_error;
}

namespace rtk {
int fail(int code, const std::string& msg) {
  g_error = msg;
  return code;
}
}  // namespace rtk

using rtk::fail;


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
}  // extern "C"

// A few host threads kept for the library's data-parallel host work (flattening a scene).  Starting a
// thread costs more than its share of flattening 10,000 objects does, so they are started once, on
// first use, and parked on a condition variable in between; they are never joined (the object lives
// until the process ends).  run(n, fn) calls fn(0) .. fn(n - 1), each once, on the parked threads
// and the caller, and returns when all have returned; an exception thrown by fn on whatever thread
// (a failed allocation, say) is caught there and the first one is thrown again by run() on the
// caller's thread.  Jobs from several threads take turns.  A forked child has none of the threads:
// there the caller does everything.
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
    std::exception_ptr thrown;
    {
      std::unique_lock<std::mutex> lock(m_);
      done_.wait(lock, [&] { return busy_ == 0; });
      job_ = nullptr;
      std::swap(thrown, first_error_);
    }
    if (thrown) std::rethrow_exception(thrown);
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
      try {
        (*job_)(i);
      } catch (...) {
        std::lock_guard<std::mutex> lock(m_);
        if (!first_error_) first_error_ = std::current_exception();
      }
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
  std::exception_ptr first_error_;  // guarded by m_
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

// The FP32 filter record of a sphere: its centre relative to `center`, rounded, and a squared radius
// that covers the rounding (and a relative 1e-9 for the exact test's own rounding).
static float4 filter_record(const SphereKey& key, const double* center) {
  const double cd[3] = {key.c[0] - center[0], key.c[1] - center[1], key.c[2] - center[2]};
  const float cf[3] = {(float)cd[0], (float)cd[1], (float)cd[2]};
  const double dx = cd[0] - cf[0], dy = cd[1] - cf[1], dz = cd[2] - cf[2];
  const double shift = std::sqrt(dx * dx + dy * dy + dz * dz);  // actual displacement of the rounded centre
  const double R = key.r * (1.0 + 1e-9) + shift * 1.0001;
  return make_float4(cf[0], cf[1], cf[2], round_up(R * R));
}

// The light grids of a scene whose lights all have w = 0 (LightGrid in rtk_device.h), one per light,
// built side by side on the library's host threads.  keys: the spheres in sweep order; gd: per light the unit
// direction and rho = -1/|L|.  False when a grid would not pay (lists too long, too many lights):
// the scene then keeps the cluster sweeps for its shadow rays.
struct LightGrids {
  std::vector<rtk::LightGrid> header;
  std::vector<uint2> cells;
  std::vector<uint32_t> entries;  // sweep positions ...
  std::vector<float4> filter;     // ... and filter records of the listed spheres
};
static bool build_light_grids(const SphereKey* keys, uint32_t n_sph, const double* center, const std::vector<double>& gd,
                              uint32_t n_lights, LightGrids& out) {
  if (n_lights == 0 || n_lights > 8u || n_sph == 0) return false;
  // the spheres as the FP32 filter sees them: recentred, rounded, radius grown by the rounding
  struct Disc {
    double c[3], r;
  };
  std::vector<Disc> discs(n_sph);
  std::vector<float4> records(n_sph);
  double scale = 0.0;
  for (uint32_t j = 0; j < n_sph; ++j) {
    const float4 f = records[j] = filter_record(keys[j], center);
    discs[j] = {{(double)f.x, (double)f.y, (double)f.z}, std::sqrt((double)f.w)};
    scale = std::fmax(scale, std::sqrt((double)f.x * f.x + (double)f.y * f.y + (double)f.z * f.z) + discs[j].r);
  }
  struct PerLight {
    rtk::LightGrid header{};
    std::vector<uint2> cells;
    std::vector<uint32_t> entries;
    bool ok = false;
  };
  std::vector<PerLight> per(n_lights);
  auto build = [&](uint32_t k, bool deep) {
    PerLight& pl = per[k];
    pl = PerLight{};
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
    // every shading point on a sphere lies at or above the lowest point of the lowest sphere; the floor
    // is lowered to the lowest corner of the box around the spheres (deep == true), so that points of a
    // plane underneath or between the spheres are in range as well
    double h_min = DBL_MAX;
    for (const Disc& d : discs) h_min = std::fmin(h_min, d.c[0] * g[0] + d.c[1] * g[1] + d.c[2] * g[2] - d.r);
    if (deep) {
      double b_lo[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, b_hi[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
      for (const Disc& d : discs)
        for (int i = 0; i < 3; ++i) b_lo[i] = std::fmin(b_lo[i], d.c[i] - d.r), b_hi[i] = std::fmax(b_hi[i], d.c[i] + d.r);
      double h_corner = 0.0;
      for (int i = 0; i < 3; ++i) h_corner += std::fmin(b_lo[i] * g[i], b_hi[i] * g[i]);
      h_min = std::fmin(h_min, h_corner);
    }
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
    // the discs are 1e-5 (1 + scale) larger than the shadows: enough against the rounding of the reference's
    // discriminant, about 64 x 2^-53 x D^2, up to D = 1e-5 (1 + scale) / 8.4e-8; 100 (1 + scale) to be safe
    const double d_max = 100.0 * (1.0 + scale);
    G.d_max2 = (float)std::fmin(d_max * d_max, 1.0e37);
    pl.ok = true;
  };
  HostPool::get().run(n_lights, [&](uint32_t k) {
    try {
      build(k, true);
      if (!per[k].ok) build(k, false);  // lists too long with the lower floor: the spheres' own floor
    } catch (const std::bad_alloc&) {
      per[k].ok = false;
    }
  });
  out.header.clear(), out.cells.clear(), out.entries.clear(), out.filter.clear();
  for (PerLight& pl : per) {
    if (!pl.ok) return false;
    pl.header.cell_base = (uint32_t)out.cells.size();
    const uint32_t entry_base = (uint32_t)out.entries.size();
    for (uint2& c : pl.cells) c.x += entry_base;
    out.header.push_back(pl.header);
    out.cells.insert(out.cells.end(), pl.cells.begin(), pl.cells.end());
    out.entries.insert(out.entries.end(), pl.entries.begin(), pl.entries.end());
  }
  out.filter.resize(out.entries.size());
  for (size_t i = 0; i < out.entries.size(); ++i) out.filter[i] = records[out.entries[i]];
  if (std::getenv("RTK_GRID_STATS"))  // a diagnostic: what the grids came to
    for (const rtk::LightGrid& G : out.header)
      std::fprintf(stderr, "rtk: light grid %u x %u cells of %.4g, h_min %.4g\n", G.nx, G.ny, 1.0 / (double)G.inv_cell, (double)G.h_min);
  if (std::getenv("RTK_GRID_STATS"))
    std::fprintf(stderr, "rtk: light grids: %zu cells, %zu entries for %u spheres\n", out.cells.size(), out.entries.size(), n_sph);
  return true;
}

// The cell grid that rays with a direction of their own (reflected and refracted rays) walk cell by
// cell (trace-grid fields of DevScene): cubic cells over the box around all bounding balls, each cell
// listing the spheres whose ball, grown by a margin, reaches into it.  The margin (3 % of a cell plus
// 1e-5 of the box's size) is what lets the device walk the cells in FP32 and stop at the first cell
// that begins behind the closest hit: a point of the ray is never further from a visited cell, and a
// computed root never further from the ball, than a small part of it.  keys: the spheres in sweep
// order (list entries are sweep positions).  False when the lists would be long (spheres much larger
// than the cells that many small ones ask for): the scene keeps the cluster sweeps.
struct TraceGridBuild {
  double lo[3] = {0, 0, 0}, inv_h = 0.0;
  uint32_t n[3] = {0, 0, 0};
  std::vector<uint2> cells;
  std::vector<float4> filter;  // per list entry: the sphere's filter record ...
  std::vector<uint32_t> slot;  // ... and its sweep position
  std::vector<rtk::TraceCell> records;  // the cells as the device reads them
};
static bool build_trace_grid(const SphereKey* keys, uint32_t n_sph, const double* center, TraceGridBuild& out) {
  if (n_sph == 0) return false;
  struct Ball {
    double c[3], r;
  };
  std::vector<Ball> balls(n_sph);
  double lo[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, hi[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX}, r_sum = 0.0;
  for (uint32_t j = 0; j < n_sph; ++j) {
    const float4 f = filter_record(keys[j], center);
    Ball& b = balls[j];
    b.c[0] = center[0] + (double)f.x, b.c[1] = center[1] + (double)f.y, b.c[2] = center[2] + (double)f.z;
    b.r = std::sqrt((double)f.w);
    r_sum += b.r;
    for (int a = 0; a < 3; ++a) lo[a] = std::fmin(lo[a], b.c[a] - b.r), hi[a] = std::fmax(hi[a], b.c[a] + b.r);
  }
  double extent[3], e_max = 0.0, diag2 = 0.0;
  for (int a = 0; a < 3; ++a) {
    extent[a] = hi[a] - lo[a];
    if (!std::isfinite(lo[a]) || !std::isfinite(hi[a]) || !(extent[a] >= 0.0)) return false;
    e_max = std::fmax(e_max, extent[a]);
    diag2 += extent[a] * extent[a];
  }
  if (!(e_max > 0.0) || !std::isfinite(diag2) || !std::isfinite(r_sum)) return false;
  const double diag = std::sqrt(diag2);
  constexpr uint32_t kAxisMax = 256u;
  // cell size: about two cells per sphere, not below three quarters of the mean radius
  auto cells_at = [&](double h) {
    double c = 1.0;
    for (int a = 0; a < 3; ++a) c *= std::fmax(1.0, std::ceil(extent[a] / h));
    return c;
  };
  double h = e_max / kAxisMax;
  double per_sphere = 2.0, least = 0.75;
  if (const char* v = std::getenv("RTK_GRID_CELLS_PER_SPHERE")) per_sphere = std::fmax(0.05, std::atof(v));  // tuning knobs
  if (const char* v = std::getenv("RTK_GRID_LEAST_CELL")) least = std::fmax(0.05, std::atof(v));
  while (cells_at(h) > per_sphere * n_sph) h *= 1.1;
  h = std::fmax(h, least * r_sum / n_sph);
  for (int attempt = 0; attempt < 5; ++attempt, h *= 1.6) {
    const double pad = 0.01 * h;  // the box begins a little outside the balls
    uint32_t n[3];
    double g_lo[3];
    for (int a = 0; a < 3; ++a) {
      g_lo[a] = lo[a] - pad;
      n[a] = (uint32_t)std::fmin((double)kAxisMax, std::fmax(1.0, std::ceil((extent[a] + 2.0 * pad) / h)));
    }
    // the last cell of an axis that was capped (or rounded) must still reach the far side
    bool covers = true;
    for (int a = 0; a < 3; ++a) covers = covers && g_lo[a] + n[a] * h >= hi[a] + 0.5 * pad;
    if (!covers) continue;
    const double inv_h = 1.0 / h, margin = 0.03 * h + 1e-5 * diag;
    auto range = [&](const Ball& b, int a, uint32_t& i0, uint32_t& i1) {
      const double a0 = (b.c[a] - b.r - margin - g_lo[a]) * inv_h, a1 = (b.c[a] + b.r + margin - g_lo[a]) * inv_h;
      i0 = a0 <= 0.0 ? 0u : std::min<uint32_t>((uint32_t)a0, n[a] - 1u);
      i1 = a1 <= 0.0 ? 0u : std::min<uint32_t>((uint32_t)a1, n[a] - 1u);
    };
    auto reaches = [&](const Ball& b, uint32_t x, uint32_t y, uint32_t z) {  // the grown ball and the cell's box
      const uint32_t i[3] = {x, y, z};
      double d2 = 0.0;
      for (int a = 0; a < 3; ++a) {
        const double c_lo = g_lo[a] + i[a] * h, c_hi = c_lo + h;
        const double d = std::fmax(0.0, std::fmax(c_lo - b.c[a], b.c[a] - c_hi));
        d2 += d * d;
      }
      const double reach = b.r + margin;
      return d2 <= reach * reach * (1.0 + 1e-9);
    };
    out.cells.assign((size_t)n[0] * n[1] * n[2], make_uint2(0u, 0u));
    auto each_cell = [&](uint32_t j, auto&& fn) {
      uint32_t x0, x1, y0, y1, z0, z1;
      range(balls[j], 0, x0, x1), range(balls[j], 1, y0, y1), range(balls[j], 2, z0, z1);
      for (uint32_t z = z0; z <= z1; ++z)
        for (uint32_t y = y0; y <= y1; ++y)
          for (uint32_t x = x0; x <= x1; ++x)
            if (reaches(balls[j], x, y, z)) fn(out.cells[((size_t)z * n[1] + y) * n[0] + x]);
    };
    size_t total = 0;
    const size_t limit = (size_t)24u * n_sph + 64u;
    for (uint32_t j = 0; j < n_sph && total <= limit; ++j) each_cell(j, [&](uint2& c) { ++c.y, ++total; });
    if (total > limit) continue;  // long lists: try larger cells
    uint32_t at = 0;
    for (uint2& c : out.cells) c.x = at, at += c.y, c.y = 0;
    out.filter.resize(total), out.slot.resize(total);
    for (uint32_t j = 0; j < n_sph; ++j) {
      const float4 f = filter_record(keys[j], center);
      each_cell(j, [&](uint2& c) {
        out.filter[c.x + c.y] = f;
        out.slot[c.x + c.y] = j;
        ++c.y;
      });
    }
    out.records.assign(out.cells.size(), rtk::TraceCell{});
    for (size_t c = 0; c < out.cells.size(); ++c) {
      rtk::TraceCell& r = out.records[c];
      r.first = out.cells[c].x, r.count = out.cells[c].y;
      for (uint32_t k = 0; k < 2u; ++k) {
        r.slot[k] = k < r.count ? out.slot[r.first + k] : 0u;
        r.filter[k] = k < r.count ? out.filter[r.first + k] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    for (int a = 0; a < 3; ++a) out.lo[a] = g_lo[a], out.n[a] = n[a];
    out.inv_h = inv_h;
    if (std::getenv("RTK_GRID_STATS"))  // a diagnostic: what the grid came to
      std::fprintf(stderr, "rtk: cell grid %u x %u x %u cells of %.4g, %zu entries for %u spheres\n", n[0], n[1], n[2], h, total, n_sph);
    return true;
  }
  return false;
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
  size_t chain_matrices = 0;  // of patterns wrapped more than once
  for (uint32_t i = 0; i < scene->n_objects; ++i) {
    const uint32_t levels = scene->objects[i].material.pattern_transformed;
    any_pattern = any_pattern || levels != 0;
    if (levels > RTK_MAX_PATTERN_LEVELS)
      return fail(RTK_ERR_UNSUPPORTED, "rtk_scene_upload: more than RTK_MAX_PATTERN_LEVELS nested pattern transforms");
    if (levels >= 2) {
      if (!scene->objects[i].material.inner_pattern_transforms)
        return fail(RTK_ERR_INVALID, "rtk_scene_upload: pattern_transformed > 1 without inner_pattern_transforms");
      chain_matrices += levels;
    }
  }
  size_t n_planes_in = 0;
  for (uint32_t i = 0; i < scene->n_objects; ++i) n_planes_in += scene->objects[i].kind == RTK_PLANE;
  const size_t n_all = scene->n_objects, n_padded = n_all + kBundle, n_l = scene->n_lights;
  const size_t bound = 20 * 256 + n_planes_in * n_l * sizeof(double2) + n_all * sizeof(DevObject) + n_l * (sizeof(DevLight) + sizeof(float4) + sizeof(uint32_t)) +
                       n_padded * (3 * sizeof(float4) + sizeof(uint32_t) + sizeof(SphRecord)) + n_all * (sizeof(double4) + sizeof(uint32_t)) +
                       (any_pattern ? n_all * 12 * sizeof(double) : 0) + chain_matrices * 16 * sizeof(double) + ((n_l + 3) / 4) * n_padded * sizeof(float4) +
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
  std::vector<double> pattern_inv;    // stays empty unless some pattern has a transform of its own
  std::vector<double> pattern_chain;  // the inverses (16 doubles each, outer to inner) of patterns wrapped more than once
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
      if (src.material.pattern_transformed == 1u) {  // (nested wrappers: below, once the objects are done)
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
  // Patterns inside several TransformPatterns (the type nests, src/graphics/patterns.rs:7-35): every
  // wrapper multiplies the point by its own inverse, the outermost first, and one whose matrix is
  // singular leaves the point as it is -- so the singular ones drop out of the chain here.  One
  // wrapper left: the per-object record of the single-wrapper case.  Several: their full inverses (the
  // transforms need not be affine, so w travels along the chain) go to pattern_chain, and the object
  // carries the chain's length (bits 8-11 of texture_kind) and its first matrix (bits 12-30).
  if (chain_matrices) {
    for (uint32_t i = 0; i < scene->n_objects; ++i) {
      const rtk_material& mat = scene->objects[i].material;
      if (mat.pattern_transformed < 2u) continue;
      std::vector<M4> chain;
      for (uint32_t k = 0; k < mat.pattern_transformed; ++k) {
        M4 pinv;
        if (m4_invert(m4_load(k == 0 ? mat.pattern_transform : mat.inner_pattern_transforms + 16 * (size_t)(k - 1)), pinv))
          chain.push_back(pinv);
      }
      if (chain.empty()) continue;
      DevObject& o = objects[i];
      o.texture_kind |= 0x80000000u;
      if (chain.size() == 1) {
        // alone, a wrapper sees the object-space point (w = 1): rows 0..2 of its inverse are all it takes
        std::memcpy(&pattern_inv[12 * (size_t)i], chain[0].e, 12 * sizeof(double));
        continue;
      }
      const size_t first = pattern_chain.size() / 16;
      if (first + chain.size() > (1u << 19))
        return fail(RTK_ERR_UNSUPPORTED, "rtk_scene_upload: more than 524,288 nested pattern transforms in one scene");
      o.texture_kind |= ((uint32_t)chain.size() << 8) | ((uint32_t)first << 12);
      for (const M4& m : chain) pattern_chain.insert(pattern_chain.end(), m.e, m.e + 16);
    }
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
  bool lights_uniform = scene->n_lights > 0, lights_bundled = scene->n_lights > 0, lights_point = scene->n_lights > 0;
  std::vector<double> gd(4 * (size_t)scene->n_lights);
  for (uint32_t k = 0; k < scene->n_lights; ++k) {
    const double* L = scene->lights[k].position;
    const double gg = L[0] * L[0] + L[1] * L[1] + L[2] * L[2];
    if (L[3] == 1.0 && std::isfinite(gg)) {
      lights_uniform = false;
      continue;
    }
    lights_point = false;
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

  // the cell grid for rays with a direction of their own: where rays recurse and the spheres are too
  // many to be intersected one by one
  TraceGridBuild tgrid;
  const bool trace_grid = any_secondary && scene->max_depth >= 2 && n_sph >= 32u && build_trace_grid(keys.data(), n_sph, center, tgrid);

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
  const size_t at_pattern_chain = reserve(&rtk_scene_handle::d_pattern_chain, pattern_chain.size() * sizeof(double));
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
  {
    struct Extra {
      size_t at;
      const void* from;
      size_t bytes;
    };
    std::vector<Extra> extras;
    auto extra = [&](void* rtk_scene_handle::*dst, const void* from, size_t bytes) {
      if (bytes) extras.push_back({reserve(dst, bytes) - extra_begin, from, bytes});
    };
    if (shadow_grid) {
      extra(&rtk_scene_handle::d_light_grid, grids.header.data(), grids.header.size() * sizeof(rtk::LightGrid));
      extra(&rtk_scene_handle::d_grid_cells, grids.cells.data(), grids.cells.size() * sizeof(uint2));
      extra(&rtk_scene_handle::d_grid_entries, grids.entries.data(), grids.entries.size() * sizeof(uint32_t));
      extra(&rtk_scene_handle::d_grid_filter, grids.filter.data(), grids.filter.size() * sizeof(float4));
    }
    if (trace_grid) {
      extra(&rtk_scene_handle::d_tg_cells, tgrid.records.data(), tgrid.records.size() * sizeof(rtk::TraceCell));
      extra(&rtk_scene_handle::d_tg_filter, tgrid.filter.data(), tgrid.filter.size() * sizeof(float4));
      extra(&rtk_scene_handle::d_tg_slot, tgrid.slot.data(), tgrid.slot.size() * sizeof(uint32_t));
    }
    flat.extra.assign(total - extra_begin, 0);
    for (const Extra& e : extras) std::memcpy(flat.extra.data() + e.at, e.from, e.bytes);
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
  if (!pattern_chain.empty()) std::memcpy(base + at_pattern_chain, pattern_chain.data(), pattern_chain.size() * sizeof(double));

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
        filter[j] = filter_record(key, center);
        const float cf[3] = {filter[j].x, filter[j].y, filter[j].z}, r2 = filter[j].w;
        mx.r2 = std::fmax(mx.r2, r2);
        mx.r = std::fmax(mx.r, round_up(std::sqrt((double)r2)));
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
  d.trace_grid = trace_grid ? 1u : 0u;
  d.lights_point = lights_point ? 1u : 0u;
  if (trace_grid) {
    for (int a = 0; a < 3; ++a) d.tg_lo[a] = tgrid.lo[a], d.tg_n[a] = tgrid.n[a];
    d.tg_inv_h = tgrid.inv_h;
  }
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

int rtk::make_dev_camera(const rtk_camera* camera, rtk::DevCamera& c) {
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
using rtk::tiles_across;
using rtk::tiles_down;

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

extern "C" {

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
}  // extern "C"

// ---------------------------------------------------------------- digest of a flattened scene
uint64_t rtk::flat_scene_digest(const rtk::FlatScene& flat) {
  uint64_t h = 0xcbf29ce484222325ull;
  auto bytes = [&](const void* p, size_t n) {
    const unsigned char* b = static_cast<const unsigned char*>(p);
    for (size_t i = 0; i < n; ++i) h = (h ^ b[i]) * 0x100000001b3ull;
  };
  auto word = [&](uint64_t v) { bytes(&v, sizeof v); };
  word(flat.total), word(flat.staged_bytes), word(flat.extra.size()), word(flat.pieces.size());
  for (const rtk::FlatScene::Piece& p : flat.pieces) {
    word(p.bytes), word(p.offset);
    // the staged pieces only: what lies between them is alignment padding that nobody writes
    if (p.offset + p.bytes <= flat.staged_bytes) bytes(flat.staged + p.offset, p.bytes);
  }
  if (!flat.extra.empty()) bytes(flat.extra.data(), flat.extra.size());
  const rtk::DevScene& d = flat.dev;
  for (uint32_t v : {d.n_objects, d.n_lights, d.n_spheres, d.n_spheres_padded, d.n_planes, d.n_clusters_padded, d.n_spheres_grouped,
                     d.max_depth, d.track_n1, d.any_secondary, d.lights_uniform, d.lights_bundled, d.k_width, d.shadow_grid,
                     d.trace_grid, d.lights_point, d.tg_n[0], d.tg_n[1], d.tg_n[2]})
    word(v);
  bytes(d.tg_lo, sizeof d.tg_lo), bytes(&d.tg_inv_h, sizeof d.tg_inv_h);
  bytes(d.center, sizeof d.center), bytes(d.ambient, sizeof d.ambient);
  bytes(&d.r2_max, sizeof d.r2_max), bytes(&d.r_max, sizeof d.r_max), bytes(&d.c_max, sizeof d.c_max);
  return h;
}

extern "C" int rtk_flatten_digest(const rtk_scene* scene, uint64_t* digest, uint64_t* device_bytes) {
  if (!scene || !digest) return fail(RTK_ERR_INVALID, "rtk_flatten_digest: null argument");
  rtk::FlatScene flat;
  const int rc = rtk::flatten_scene(scene, flat);
  if (rc != RTK_OK) return rc;
  *digest = rtk::flat_scene_digest(flat);
  if (device_bytes) *device_bytes = flat.total;
  return RTK_OK;
}
