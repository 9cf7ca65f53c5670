/*
 * oracle.h -- CPU oracle for the per-pixel ray-trace loop of mattkleiny/raytracer-rust.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE.  This is a literal f64, 4-component
 * restatement of the reference's algorithm (file:line citations sit on every function
 * in oracle.cpp).  Only the test-suite, the smoke check and the benchmark's
 * `cpu_baseline` leg may load it; the shipped library (librtk.so) never links,
 * loads or calls anything in this directory.
 *
 * Parity pins: the reference's own unit-test expectations (tests/golden/kat.json) and
 * the reference's rendered image output.png (tests/golden/output_stride4.npy plus the
 * sha256 of the full decoded frame).  The reference itself is Rust and cannot be built
 * in this environment (no rustc/cargo), so there is no oracle/_ref build.
 *
 * The scene and camera structs are the ones of include/rtk.h, so the oracle and the
 * device path always see byte-identical inputs.
 */
#ifndef RTK_ORACLE_H
#define RTK_ORACLE_H

#include "../include/rtk.h"

#ifdef __cplusplus
extern "C" {
#endif

/* One entry of a hit list: object index into rtk_scene.objects and ray parameter. */
typedef struct oracle_hit {
  uint32_t object;
  uint32_t reserved_;
  double distance;
} oracle_hit;

/* `LightingData` (src/scene/lighting.rs:24-36). */
typedef struct oracle_lighting_data {
  uint32_t object;
  uint32_t inside;
  double world_position[4];
  double over_position[4];
  double under_position[4];
  double object_position[4];
  double eye[4];
  double normal[4];
  double reflect_direction[4];
  double distance;
  double refractivity[2];
} oracle_lighting_data;

/* --- maths -------------------------------------------------------------------- */
int oracle_matrix_invert(const double m[16], double out[16]); /* 1 = inverted, 0 = singular */
void oracle_matrix_mul(const double a[16], const double b[16], double out[16]);
void oracle_matrix_mul_vector(const double m[16], const double v[4], double out[4]);
void oracle_matrix_transpose(const double m[16], double out[16]);
void oracle_translate(double x, double y, double z, double out[16]);
void oracle_scale(double x, double y, double z, double out[16]);
void oracle_rotate_x(double r, double out[16]);
void oracle_rotate_y(double r, double out[16]);
void oracle_rotate_z(double r, double out[16]);
void oracle_look_at(const double from[4], const double to[4], const double up[4], double out[16]);
void oracle_reflect(const double v[4], const double normal[4], double out[4]);
uint8_t oracle_cast_u8(double c);

/* --- shapes, patterns, camera --------------------------------------------------- */
/* `SceneNode::intersect` for one object; writes up to 2 distances, returns the count. */
int oracle_object_intersect(const rtk_object* object, const double origin[4], const double direction[4],
                            double distances[2]);
void oracle_object_normal_at(const rtk_object* object, const double world_point[4], double out[4]);
void oracle_texture_sample_at(const rtk_material* material, const double point[4], double out[3]);
int oracle_camera_init(rtk_camera* camera, uint32_t width, uint32_t height, double field_of_view);
int oracle_camera_ray_for_pixel(const rtk_camera* camera, uint64_t x, uint64_t y, double origin[4],
                                double direction[4]); /* 0 = singular camera */

/* --- scene ---------------------------------------------------------------------- */
/* `Scene::intersect`: all hits, stably sorted by distance.  Returns the total count;
 * writes at most `capacity` entries. */
uint64_t oracle_scene_intersect(const rtk_scene* scene, const double origin[4], const double direction[4],
                                oracle_hit* hits, uint64_t capacity);
/* `HitList::closest_hit`: index into `hits` of the winning entry, or -1. */
int64_t oracle_closest_hit(const oracle_hit* hits, uint64_t n);
int oracle_is_shadowed(const rtk_scene* scene, const double point[4]);
void oracle_phong_lighting(const rtk_light* light, const rtk_material* material, const double world_position[4],
                           const double object_position[4], const double eye[4], const double normal[4],
                           int in_shadow, double out[3]);
void oracle_lighting_data_calculate(const rtk_scene* scene, const double origin[4], const double direction[4],
                                    const oracle_hit* hits, uint64_t n_hits, uint64_t which,
                                    oracle_lighting_data* out);
void oracle_apply_lighting(const rtk_scene* scene, const double origin[4], const double direction[4],
                           const oracle_hit* hits, uint64_t n_hits, uint64_t which, uint32_t depth,
                           double out[3]);
void oracle_reflected_color(const rtk_scene* scene, const oracle_lighting_data* data, uint32_t depth, double out[3]);
void oracle_refracted_color(const rtk_scene* scene, const oracle_lighting_data* data, uint32_t depth, double out[3]);
double oracle_shlick(const oracle_lighting_data* data);
/* `Scene::trace`. */
void oracle_trace(const rtk_scene* scene, const double origin[4], const double direction[4], double out[3]);

/* `Camera::render` over the pixel rectangle [x0, x1) x [y0, y1) of the frame.
 * rgb_f64 ((x1-x0)*(y1-y0)*3) and rgba8 ((x1-x0)*(y1-y0)*4) are row-major inside the
 * rectangle; either may be NULL.  threads <= 1 runs the reference's single-threaded
 * loop; otherwise rows are split over `threads` std::threads.  stats (may be NULL)
 * receives the ray counts of the rectangle.  Returns 0, or -2 for a singular camera. */
int oracle_render(const rtk_scene* scene, const rtk_camera* camera, uint32_t x0, uint32_t y0, uint32_t x1,
                  uint32_t y1, double* rgb_f64, uint8_t* rgba8, rtk_stats* stats, int threads);

/* The same with samples_per_axis^2 samples per pixel, as include/rtk.h defines them (the
 * reference has no counterpart): rgb_f64 receives the f64 mean of the samples' FP32 colours. */
int oracle_render_sampled(const rtk_scene* scene, const rtk_camera* camera, uint32_t samples_per_axis, uint32_t x0,
                          uint32_t y0, uint32_t x1, uint32_t y1, double* rgb_f64, uint8_t* rgba8, rtk_stats* stats,
                          int threads);

#ifdef __cplusplus
}
#endif
#endif
