// rtk_render -- the reference's entry point (src/main.rs:12-24) on the GPU:
//   load a scene file, Camera::new(1920, 1080, PI / 3), render, save_to_png("./output.png").
//
//   rtk_render [scene.yaml] [output.png] [--width W] [--height H] [--fov RADIANS] [--device N] [--samples S]
// --samples S averages S x S samples per pixel (the reference has one, src/scene/cameras.rs:53-54).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "../include/rtk_host.hpp"

int main(int argc, char** argv) {
  using namespace rtkhost;
  std::string scene_path = "assets/scenes/test01.yaml", out_path = "./output.png";
  uint32_t width = 1920, height = 1080;
  double fov = PI / 3.0;
  int device_id = 0, positional = 0;
  uint32_t samples = 1;
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    auto value = [&]() -> const char* {
      if (i + 1 >= argc) {
        std::fprintf(stderr, "rtk_render: %s needs a value\n", a.c_str());
        std::exit(2);
      }
      return argv[++i];
    };
    if (a == "--width") width = (uint32_t)std::strtoul(value(), nullptr, 10);
    else if (a == "--height") height = (uint32_t)std::strtoul(value(), nullptr, 10);
    else if (a == "--fov") fov = std::strtod(value(), nullptr);
    else if (a == "--device") device_id = std::atoi(value());
    else if (a == "--samples") samples = (uint32_t)std::strtoul(value(), nullptr, 10);
    else if (positional == 0) scene_path = a, ++positional;
    else if (positional == 1) out_path = a, ++positional;
    else {
      std::fprintf(stderr, "usage: rtk_render [scene.yaml] [output.png] [--width W] [--height H] [--fov R] [--device N] [--samples S]\n");
      return 2;
    }
  }
  try {
    const Scene scene = PackedScene::from_yaml_file(scene_path).build();
    const Camera camera(width, height, fov);
    Device device(device_id);
    const Canvas canvas = camera.render(scene, device, samples);
    canvas.save_to_png(out_path);
    const rtk_stats& s = canvas.stats;
    std::printf("%s: %ux%u, %zu objects, %zu lights -> %s\n", scene_path.c_str(), width, height, scene.nodes.size(),
                scene.lights.size(), out_path.c_str());
    std::printf("rays: %llu primary, %llu shadow, %llu secondary; kernels %.3f ms\n", (unsigned long long)s.rays_primary,
                (unsigned long long)s.rays_shadow, (unsigned long long)s.rays_secondary, s.kernel_ms);
  } catch (const Error& e) {
    std::fprintf(stderr, "rtk_render: %s\n", e.what());
    return 1;
  }
  return 0;
}
