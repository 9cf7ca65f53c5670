/* The C ABI without any host library: one sphere over a plane, one light, rendered on the GPU
 * and written as a binary PPM.
 *
 *   gcc -std=c99 -Iinclude examples/minimal.c -Lraytracer-rust_amd/csrc -lrtk -lm -o minimal
 *   LD_LIBRARY_PATH=raytracer-rust_amd/csrc ./minimal out.ppm
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rtk.h"

static void identity(double m[16]) {
  memset(m, 0, 16 * sizeof(double));
  m[0] = m[5] = m[10] = m[15] = 1.0;
}

static rtk_material default_material(void) { /* src/scene/materials.rs:34-48 */
  rtk_material m;
  memset(&m, 0, sizeof m);
  m.texture_kind = RTK_TEX_SOLID;
  m.color_a[0] = m.color_a[1] = m.color_a[2] = 1.0;
  m.ambient = 0.1, m.diffuse = 0.9, m.specular = 0.9, m.shininess = 200.0;
  m.refractivity = 1.0;
  identity(m.pattern_transform);
  return m;
}

int main(int argc, char** argv) {
  const char* path = argc > 1 ? argv[1] : "minimal.ppm";
  rtk_object objects[2];
  memset(objects, 0, sizeof objects);
  objects[0].kind = RTK_PLANE; /* y = 0, checker */
  identity(objects[0].transform);
  objects[0].plane_normal[1] = 1.0;
  objects[0].material = default_material();
  objects[0].material.texture_kind = RTK_TEX_CHECKER;
  objects[0].material.specular = 0.0;
  objects[1].kind = RTK_SPHERE; /* translate(0, 1, 0) */
  identity(objects[1].transform);
  objects[1].transform[7] = 1.0;
  objects[1].material = default_material();
  objects[1].material.color_a[1] = 0.2, objects[1].material.color_a[2] = 0.2;

  rtk_light light = {{-10.0, 10.0, -10.0, 0.0}, {1.0, 1.0, 1.0}}; /* w = 0, as scene files give it */
  rtk_scene scene;
  memset(&scene, 0, sizeof scene);
  scene.objects = objects, scene.n_objects = 2;
  scene.lights = &light, scene.n_lights = 1;
  scene.max_depth = RTK_REFERENCE_MAX_DEPTH;

  rtk_camera camera;
  if (rtk_camera_init(&camera, 640, 360, 3.14159265358979323846 / 3.0) != RTK_OK) {
    fprintf(stderr, "%s\n", rtk_last_error());
    return 1;
  }
  rtk_context* ctx = NULL;
  if (rtk_create(0, &ctx) != RTK_OK) {
    fprintf(stderr, "%s\n", rtk_last_error());
    return 1;
  }
  unsigned char* rgba = malloc((size_t)camera.width * camera.height * 4);
  rtk_stats stats;
  int rc = rgba ? rtk_render(ctx, &scene, &camera, NULL, rgba, &stats) : RTK_ERR_NOMEM;
  if (rc != RTK_OK) {
    fprintf(stderr, "rtk_render: %s\n", rtk_last_error());
    rtk_destroy(ctx);
    return 1;
  }
  FILE* f = fopen(path, "wb");
  if (!f) return 1;
  fprintf(f, "P6\n%u %u\n255\n", camera.width, camera.height);
  for (size_t i = 0; i < (size_t)camera.width * camera.height; ++i) fwrite(rgba + 4 * i, 1, 3, f);
  fclose(f);
  printf("%llu rays in %.3f ms -> %s\n",
         (unsigned long long)(stats.rays_primary + stats.rays_shadow + stats.rays_secondary), stats.kernel_ms, path);
  free(rgba);
  rtk_destroy(ctx);
  return 0;
}
