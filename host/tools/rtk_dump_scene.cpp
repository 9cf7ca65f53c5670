// rtk_dump_scene -- prints the flat rtk_scene a scene file packs to, as JSON with full
// f64 precision.  Needs no GPU; used to compare the scene-file reader with other readers.
//   rtk_dump_scene scene.yaml
#include <cstdio>

#include "../include/rtk_host.hpp"

static void arr(const double* v, int n) {
  std::printf("[");
  for (int i = 0; i < n; ++i) std::printf("%s%.17g", i ? ", " : "", v[i]);
  std::printf("]");
}

int main(int argc, char** argv) {
  using namespace rtkhost;
  if (argc != 2) {
    std::fprintf(stderr, "usage: rtk_dump_scene scene.yaml\n");
    return 2;
  }
  try {
    const Scene scene = PackedScene::from_yaml_file(argv[1]).build();
    Scene::Packed p;
    scene.pack(p);
    std::printf("{\"max_depth\": %u, \"ambient_color\": ", p.scene.max_depth);
    arr(p.scene.ambient_color, 3);
    std::printf(", \"lights\": [");
    for (uint32_t k = 0; k < p.scene.n_lights; ++k) {
      std::printf("%s{\"position\": ", k ? ", " : "");
      arr(p.lights[k].position, 4);
      std::printf(", \"intensity\": ");
      arr(p.lights[k].intensity, 3);
      std::printf("}");
    }
    std::printf("], \"objects\": [");
    for (uint32_t i = 0; i < p.scene.n_objects; ++i) {
      const rtk_object& o = p.objects[i];
      const rtk_material& m = o.material;
      std::printf("%s{\"kind\": %u, \"transform\": ", i ? ", " : "", o.kind);
      arr(o.transform, 16);
      std::printf(", \"plane_normal\": ");
      arr(o.plane_normal, 4);
      std::printf(", \"texture_kind\": %u, \"color_a\": ", m.texture_kind);
      arr(m.color_a, 3);
      std::printf(", \"color_b\": ");
      arr(m.color_b, 3);
      const double s[7] = {m.ambient, m.diffuse, m.specular, m.shininess, m.transparency, m.reflectivity, m.refractivity};
      std::printf(", \"scalars\": ");
      arr(s, 7);
      std::printf("}");
    }
    std::printf("]}\n");
  } catch (const Error& e) {
    std::fprintf(stderr, "rtk_dump_scene: %s\n", e.what());
    return 1;
  }
  return 0;
}
