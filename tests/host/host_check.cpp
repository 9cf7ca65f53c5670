// host_check -- runs the host half of the library (csrc/rtk_host.cpp: validation, the reference's matrix
// arithmetic, flattening into the device's record arrays, camera and partition arithmetic) without a
// device, so that it can run under AddressSanitizer, UndefinedBehaviorSanitizer and ThreadSanitizer
// (make -C raytracer-rust_amd/csrc asan tsan).  Test infrastructure: built and run by
// tests/test_host_sanitizers.py.
//
//   host_check [--threads N] [--repeats K] scene.bin ...
//
// Every scene file (written by the test: a header and the raw rtk_object / rtk_light arrays) is flattened
// once on the main thread and then K times on each of N threads at once; all digests of a scene must
// agree.  One line per scene goes to stdout: "scene <file> digest <hex> bytes <n>".  Then the argument
// checks and the arithmetic self-tests run.  Exit status 0 when everything holds.
#include "rtk_host.h"

#include <atomic>
#include <cinttypes>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <thread>
#include <vector>

namespace {

int failures = 0;
#define CHECK(cond)                                                       \
  do {                                                                    \
    if (!(cond)) {                                                        \
      std::fprintf(stderr, "%s:%d: check failed: %s\n", __FILE__, __LINE__, #cond); \
      ++failures;                                                         \
    }                                                                     \
  } while (0)

struct LoadedScene {
  std::vector<rtk_object> objects;
  std::vector<rtk_light> lights;
  rtk_scene scene{};
};

bool load_scene(const char* path, LoadedScene& out) {
  std::FILE* f = std::fopen(path, "rb");
  if (!f) return false;
  struct Header {
    char magic[4];
    uint32_t n_objects, n_lights, max_depth;
    double ambient[3];
  } h;
  bool ok = std::fread(&h, sizeof h, 1, f) == 1 && std::memcmp(h.magic, "RTKS", 4) == 0;
  if (ok) {
    out.objects.resize(h.n_objects);
    out.lights.resize(h.n_lights);
    ok = (h.n_objects == 0 || std::fread(out.objects.data(), sizeof(rtk_object), h.n_objects, f) == h.n_objects) &&
         (h.n_lights == 0 || std::fread(out.lights.data(), sizeof(rtk_light), h.n_lights, f) == h.n_lights);
  }
  std::fclose(f);
  if (!ok) return false;
  for (rtk_object& o : out.objects) {  // a pointer of the process that wrote the file means nothing here
    if (o.material.pattern_transformed > 1u) return false;
    o.material.inner_pattern_transforms = nullptr;
  }
  out.scene.objects = out.objects.data();
  out.scene.n_objects = h.n_objects;
  out.scene.lights = out.lights.data();
  out.scene.n_lights = h.n_lights;
  out.scene.max_depth = h.max_depth;
  std::memcpy(out.scene.ambient_color, h.ambient, sizeof h.ambient);
  return true;
}

rtk_object unit_sphere() {
  rtk_object o{};
  o.kind = RTK_SPHERE;
  for (int i = 0; i < 4; ++i) o.transform[5 * i] = 1.0;
  o.material.color_a[0] = o.material.color_a[1] = o.material.color_a[2] = 1.0;
  o.material.ambient = 0.1, o.material.diffuse = 0.9, o.material.specular = 0.9, o.material.shininess = 200.0;
  o.material.refractivity = 1.0;
  return o;
}

void argument_checks() {
  uint64_t digest = 0, bytes = 0;
  CHECK(rtk_flatten_digest(nullptr, &digest, &bytes) == RTK_ERR_INVALID);
  rtk_object o = unit_sphere();
  rtk_light l{};
  l.position[1] = 10.0;
  l.intensity[0] = l.intensity[1] = l.intensity[2] = 1.0;
  rtk_scene s{};
  s.objects = &o, s.n_objects = 1, s.lights = &l, s.n_lights = 1, s.max_depth = RTK_REFERENCE_MAX_DEPTH;
  CHECK(rtk_flatten_digest(&s, &digest, &bytes) == RTK_OK && bytes > 0);
  const uint64_t good = digest;
  CHECK(rtk_flatten_digest(&s, &digest, nullptr) == RTK_OK && digest == good);
  o.kind = 7;
  CHECK(rtk_flatten_digest(&s, &digest, &bytes) == RTK_ERR_INVALID && std::strstr(rtk_last_error(), "object kind"));
  o = unit_sphere();
  o.material.texture_kind = 9;
  CHECK(rtk_flatten_digest(&s, &digest, &bytes) == RTK_ERR_INVALID);
  o = unit_sphere();
  o.transform[3] = std::numeric_limits<double>::quiet_NaN();
  CHECK(rtk_flatten_digest(&s, &digest, &bytes) == RTK_ERR_INVALID);
  o = unit_sphere();
  o.transform[12] = 0.5;  // projective
  CHECK(rtk_flatten_digest(&s, &digest, &bytes) == RTK_ERR_UNSUPPORTED);
  o = unit_sphere();
  o.kind = RTK_PLANE;
  o.plane_normal[1] = 1.0, o.plane_normal[3] = 1.0;
  CHECK(rtk_flatten_digest(&s, &digest, &bytes) == RTK_ERR_UNSUPPORTED);
  o = unit_sphere();
  s.max_depth = RTK_MAX_DEPTH_LIMIT + 1;
  CHECK(rtk_flatten_digest(&s, &digest, &bytes) == RTK_ERR_UNSUPPORTED);
  s.max_depth = RTK_REFERENCE_MAX_DEPTH;
  s.objects = nullptr;
  CHECK(rtk_flatten_digest(&s, &digest, &bytes) == RTK_ERR_INVALID);
  s.objects = &o;
  // a singular object transform silently gets the identity inverse (src/scene.rs:55-62)
  rtk_object flat_o = unit_sphere();
  flat_o.transform[0] = 0.0;
  s.objects = &flat_o;
  uint64_t singular = 0;
  CHECK(rtk_flatten_digest(&s, &singular, nullptr) == RTK_OK && singular != 0);
  // nested pattern transforms: the count needs its matrices, and has a limit
  {
    rtk_object nested = unit_sphere();
    nested.material.texture_kind = RTK_TEX_CHECKER;
    nested.material.pattern_transformed = 3;
    for (int i = 0; i < 4; ++i) nested.material.pattern_transform[5 * i] = 1.0;
    s.objects = &nested;
    CHECK(rtk_flatten_digest(&s, &digest, &bytes) == RTK_ERR_INVALID);
    double inner[2 * 16] = {};
    for (int k = 0; k < 2; ++k)
      for (int i = 0; i < 4; ++i) inner[16 * k + 5 * i] = 2.0 + k;
    nested.material.inner_pattern_transforms = inner;
    uint64_t with_chain = 0;
    CHECK(rtk_flatten_digest(&s, &with_chain, &bytes) == RTK_OK);
    inner[16 + 5] = 0.0;  // the innermost wrapper singular: a chain of two is left
    uint64_t shorter = 0;
    CHECK(rtk_flatten_digest(&s, &shorter, &bytes) == RTK_OK && shorter != with_chain);
    nested.material.pattern_transformed = RTK_MAX_PATTERN_LEVELS + 1;
    CHECK(rtk_flatten_digest(&s, &digest, &bytes) == RTK_ERR_UNSUPPORTED);
    s.objects = &o;
  }
  // an empty scene is a scene
  rtk_scene empty{};
  empty.max_depth = RTK_REFERENCE_MAX_DEPTH;
  CHECK(rtk_flatten_digest(&empty, &digest, &bytes) == RTK_OK);
}

void camera_checks() {
  rtk_camera cam;
  CHECK(rtk_camera_init(&cam, 200, 125, std::acos(-1.0) / 2.0) == RTK_OK);
  CHECK(std::fabs(cam.pixel_size - 0.01) < 1e-12);  // src/scene/cameras.rs:95-104
  CHECK(rtk_camera_init(&cam, 125, 200, std::acos(-1.0) / 2.0) == RTK_OK && std::fabs(cam.pixel_size - 0.01) < 1e-12);
  CHECK(rtk_camera_init(nullptr, 1, 1, 1.0) == RTK_ERR_INVALID && rtk_camera_init(&cam, 0, 1, 1.0) == RTK_ERR_INVALID);
  CHECK(rtk_camera_init(&cam, 64, 36, 1.0) == RTK_OK);
  rtk::DevCamera dc;
  CHECK(rtk::make_dev_camera(&cam, dc) == RTK_OK && dc.width == 64 && dc.height == 36);
  // default view: look_at((0, 1.5, -5), ...): the origin comes back out of the inverse
  CHECK(std::fabs(dc.origin[0]) < 1e-12 && std::fabs(dc.origin[1] - 1.5) < 1e-12 && std::fabs(dc.origin[2] + 5.0) < 1e-12);
  rtk_camera bad = cam;
  bad.transform[0] = bad.transform[1] = bad.transform[2] = 0.0;
  CHECK(rtk::make_dev_camera(&bad, dc) == RTK_ERR_SINGULAR_CAMERA);
  bad = cam;
  bad.transform[13] = 1.0;
  CHECK(rtk::make_dev_camera(&bad, dc) == RTK_ERR_UNSUPPORTED);
  rtk_camera fine;
  CHECK(rtk_camera_supersample(&cam, 4, &fine) == RTK_OK && fine.width == 256 && fine.pixel_size == cam.pixel_size / 4.0);
  CHECK(rtk_camera_supersample(&cam, 0, &fine) == RTK_ERR_INVALID && rtk_camera_supersample(&cam, 17, &fine) == RTK_ERR_INVALID);
}

// Tiles, bands and shares against a plain enumeration of the tile rows.
void partition_checks() {
  uint64_t state = 0x9E3779B97F4A7C15ull;
  auto next = [&]() {
    state += 0x9E3779B97F4A7C15ull;
    uint64_t z = state;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  };
  for (int trial = 0; trial < 4000; ++trial) {
    const uint32_t rows = (uint32_t)(next() % 300), period = 1 + (uint32_t)(next() % 64);
    const uint32_t width = 1 + (uint32_t)(next() % period), offset = (uint32_t)(next() % (period - width + 1));
    uint64_t want = 0;
    for (uint32_t r = 0; r < rows; ++r) want += (r % period) >= offset && (r % period) < offset + width;
    CHECK(rtk::rows_of_share(rows, period, offset, width) == want);
  }
  for (int trial = 0; trial < 2000; ++trial) {
    const uint32_t w = 1 + (uint32_t)(next() % 700), h = 1 + (uint32_t)(next() % 500), world = 1 + (uint32_t)(next() % 9);
    uint64_t tiles = 0, pixels_rows = 0;
    const uint32_t m = 1 + (uint32_t)(next() % 12);
    for (uint32_t rank = 0; rank < world; ++rank) {
      tiles += rtk_tiles_for_rank(w, h, rank, world);
      rtk_banding bd = {rank, world, m, 0u, 0u, 1u};
      pixels_rows += rtk_band_tile_rows(h, &bd);
      CHECK(rtk_band_pixel_rows(h, &bd) == rtk_band_tile_rows(h, &bd) * RTK_TILE_H);
      rtk_camera cam;
      CHECK(rtk_camera_init(&cam, w, h, 1.0) == RTK_OK);
      rtk::Partition p;
      CHECK(rtk::partition_from_banding(&cam, &bd, p, "check") == RTK_OK);
      CHECK(p.n_local_tiles == rtk_band_tile_rows(h, &bd) * rtk::tiles_across(w));
      rtk::Partition q;
      CHECK(rtk::partition_from_rows(&cam, 0u, m * world, m * rank, m, true, q, "check") == RTK_OK && q.n_local_tiles == p.n_local_tiles);
    }
    CHECK(tiles == (uint64_t)rtk::tiles_across(w) * rtk::tiles_down(h));
    CHECK(pixels_rows == rtk::tiles_down(h));
    CHECK(rtk_tiles_for_rank(w, h, world, world) == 0 && rtk_tiles_for_rank(w, h, 0, 0) == 0);
  }
  // band sizes whose products leave 32 bits: one round, or nothing
  rtk_banding huge = {0u, 4u, 0x80000000u, 0u, 0u, 0u};
  CHECK(rtk_band_tile_rows(2160, &huge) == 270);
  huge.rank = 1;
  CHECK(rtk_band_tile_rows(2160, &huge) == 0);
  rtk_banding bad = {2u, 2u, 1u, 0u, 0u, 0u};
  CHECK(rtk_band_tile_rows(2160, &bad) == 0);
  CHECK(rtk_compact_pixels_sampled(64, 36, 2, 0, 1) == 64u * 36u && rtk_compact_pixels_sampled(64, 36, 3, 0, 2) == 0);
}

}  // namespace

int main(int argc, char** argv) {
  unsigned n_threads = 8, repeats = 2;
  std::vector<const char*> files;
  for (int i = 1; i < argc; ++i) {
    if (!std::strcmp(argv[i], "--threads") && i + 1 < argc)
      n_threads = (unsigned)std::atoi(argv[++i]);
    else if (!std::strcmp(argv[i], "--repeats") && i + 1 < argc)
      repeats = (unsigned)std::atoi(argv[++i]);
    else
      files.push_back(argv[i]);
  }
  for (const char* path : files) {
    LoadedScene ls;
    if (!load_scene(path, ls)) {
      std::fprintf(stderr, "cannot read %s\n", path);
      return 2;
    }
    uint64_t digest = 0, bytes = 0;
    const int rc = rtk_flatten_digest(&ls.scene, &digest, &bytes);
    if (rc != RTK_OK) {
      std::fprintf(stderr, "%s: %s\n", path, rtk_last_error());
      return 2;
    }
    // the same scene from several threads at once: the library's parked host threads serve one job at a
    // time, the callers' own shares run side by side
    std::atomic<unsigned> disagree{0};
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < n_threads; ++t)
      pool.emplace_back([&]() {
        for (unsigned k = 0; k < repeats; ++k) {
          uint64_t d = 0, b = 0;
          if (rtk_flatten_digest(&ls.scene, &d, &b) != RTK_OK || d != digest || b != bytes) ++disagree;
        }
      });
    for (std::thread& t : pool) t.join();
    CHECK(disagree.load() == 0);
    std::printf("scene %s digest %016" PRIx64 " bytes %" PRIu64 "\n", path, digest, bytes);
  }
  // a thread that fails validation sets its own error text only
  {
    std::thread other([]() {
      uint64_t d = 0;
      (void)rtk_flatten_digest(nullptr, &d, nullptr);
    });
    other.join();
  }
  argument_checks();
  camera_checks();
  partition_checks();
  if (failures) {
    std::fprintf(stderr, "%d check(s) failed\n", failures);
    return 1;
  }
  std::printf("HOST_CHECK_OK\n");
  return 0;
}
