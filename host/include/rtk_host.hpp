// rtk_host.hpp -- C++ host side above the C ABI of include/rtk.h.
//
// Same names and argument meaning as the reference's object model, so that code written
// against the reference reads the same here:
//   point / vec3 / rgb           src/maths/vectors.rs:12-24, src/maths/colors.rs:8-10
//   Matrix4x4 + transforms       src/maths/matrices.rs, src/maths/transforms.rs
//   Texture, Material::with_*    src/scene/materials.rs:7-95
//   PointLight                   src/scene/lighting.rs:8-21
//   Sphere(), Plane(normal)      src/scene/shapes/spheres.rs:10-15, planes.rs:14-19, src/scene.rs:36-68
//   Scene                        src/scene.rs:102-133
//   Camera, Camera::render       src/scene/cameras.rs:7-85
//   Canvas, save_to_png          src/graphics.rs:12-81
//   PackedScene                  src/loader.rs:22-200
// Nothing here computes a pixel: the classes only build the flat f64 structs of the ABI and
// hand them to librtk.so.  Errors of the library surface as rtkhost::Error.
#ifndef RTK_HOST_HPP
#define RTK_HOST_HPP

#include <array>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/rtk.h"

namespace rtkhost {

constexpr double PI = 3.14159265358979323846;

struct Error : std::runtime_error {
  int code;
  Error(int code_, const std::string& what) : std::runtime_error(what), code(code_) {}
};

struct Vector {
  double x, y, z, w;
};
inline Vector point(double x, double y, double z) { return {x, y, z, 1.0}; }
inline Vector vec3(double x, double y, double z) { return {x, y, z, 0.0}; }
inline Vector vec4(double x, double y, double z, double w) { return {x, y, z, w}; }

struct Color {
  double r, g, b;
};
inline Color rgb(double r, double g, double b) { return {r, g, b}; }
constexpr Color BLACK{0, 0, 0}, WHITE{1, 1, 1};

// Row-major, element (row, column) at [column + 4 * row] (src/maths/matrices.rs:69-71).
struct Matrix4x4 {
  std::array<double, 16> e{};
  static Matrix4x4 identity();
  static Matrix4x4 translate(double x, double y, double z);
  static Matrix4x4 scale(double x, double y, double z);
  static Matrix4x4 rotate_x(double r);
  static Matrix4x4 rotate_y(double r);
  static Matrix4x4 rotate_z(double r);
  static Matrix4x4 shear(double x1, double x2, double y1, double y2, double z1, double z2);
  static Matrix4x4 look_at(Vector from, Vector to, Vector up);
  double at(int row, int col) const { return e[col + 4 * row]; }
  Matrix4x4 operator*(const Matrix4x4& rhs) const;
};

struct Texture {
  uint32_t kind = RTK_TEX_SOLID;
  Color a = WHITE, b = BLACK;
  bool transformed = false;  // wrapped in a `TransformPattern` (src/graphics/patterns.rs:7-35)
  Matrix4x4 transform = Matrix4x4::identity();  // ... the outermost one's
  std::vector<Matrix4x4> inner;                 // the transforms of the TransformPatterns inside it, outer to inner
  static Texture solid(Color c) { return make(RTK_TEX_SOLID, c, BLACK); }
  static Texture checker(Color a, Color b) { return make(RTK_TEX_CHECKER, a, b); }
  static Texture gradient(Color a, Color b) { return make(RTK_TEX_GRADIENT, a, b); }
  static Texture ring(Color a, Color b) { return make(RTK_TEX_RING, a, b); }
  static Texture striped(Color a, Color b) { return make(RTK_TEX_STRIPED, a, b); }
  // `TransformPattern::with_transform`: repeated calls multiply on the right (patterns.rs:22-24)
  Texture with_transform(const Matrix4x4& t) const {
    Texture r = *this;
    r.transformed = true;
    r.transform = transform * t;
    return r;
  }
  // `TransformPattern::new(self)`: one more wrapper, with the identity transform, around what is there
  // (the type is generic over its pattern, so wrappers nest)
  Texture wrap() const {
    Texture r = *this;
    if (transformed) r.inner.insert(r.inner.begin(), transform);
    r.transformed = true;
    r.transform = Matrix4x4::identity();
    return r;
  }

 private:
  static Texture make(uint32_t kind, Color a, Color b) {
    Texture t;
    t.kind = kind, t.a = a, t.b = b;
    return t;
  }
};

// Defaults of src/scene/materials.rs:34-48.
struct Material {
  Texture texture = Texture::solid(WHITE);
  double ambient = 0.1, diffuse = 0.9, specular = 0.9, shininess = 200.0;
  double transparency = 0.0, reflectivity = 0.0, refractivity = 1.0;
  static Material default_material() { return Material{}; }
  Material with_color(Color c) const { Material m = *this; m.texture = Texture::solid(c); return m; }
  Material with_pattern(Texture t) const { Material m = *this; m.texture = t; return m; }
  Material with_ambient(double v) const { Material m = *this; m.ambient = v; return m; }
  Material with_diffuse(double v) const { Material m = *this; m.diffuse = v; return m; }
  Material with_specular(double v) const { Material m = *this; m.specular = v; return m; }
  Material with_shininess(double v) const { Material m = *this; m.shininess = v; return m; }
  Material with_transparency(double v) const { Material m = *this; m.transparency = v; return m; }
  Material with_reflective(double v) const { Material m = *this; m.reflectivity = v; return m; }
  Material with_refractivity(double v) const { Material m = *this; m.refractivity = v; return m; }
};

// position keeps its w: vec3(..) gives the w = 0 light a scene file produces
// (src/loader.rs:35-40), point(..) a w = 1 light.
struct PointLight {
  Vector position;
  Color intensity = WHITE;
};

struct SceneNode {
  uint32_t kind = RTK_SPHERE;
  Vector plane_normal{0, 0, 0, 0};
  Material material;
  Matrix4x4 transform = Matrix4x4::identity();
  SceneNode with_transform(const Matrix4x4& t) const { SceneNode n = *this; n.transform = t; return n; }
  SceneNode with_material(const Material& m) const { SceneNode n = *this; n.material = m; return n; }
};
inline SceneNode Sphere() { return SceneNode{}; }
inline SceneNode Plane(Vector normal = vec3(0, 1, 0)) {
  SceneNode n;
  n.kind = RTK_PLANE;
  n.plane_normal = normal;
  return n;
}

class Scene {
 public:
  static constexpr uint32_t MAX_DEPTH = RTK_REFERENCE_MAX_DEPTH;  // src/scene.rs:112
  Color ambient_color = BLACK;
  uint32_t max_depth = MAX_DEPTH;
  std::vector<SceneNode> nodes;
  std::vector<PointLight> lights;
  void add_object(const SceneNode& n) { nodes.push_back(n); }
  void add_light(const PointLight& l) { lights.push_back(l); }

  // Flat view for the ABI; valid while `storage` lives and the scene is not modified.
  struct Packed {
    std::vector<rtk_object> objects;
    std::vector<rtk_light> lights;
    std::vector<std::vector<double>> inner_transforms;  // what rtk_material.inner_pattern_transforms points into
    rtk_scene scene;
  };
  void pack(Packed& storage) const;
};

class Canvas {
 public:
  Canvas(uint32_t width, uint32_t height)
      : width_(width), height_(height), rgb_((size_t)width * height * 3, 0.f), rgba8_((size_t)width * height * 4, 0) {}
  uint32_t width() const { return width_; }
  uint32_t height() const { return height_; }
  Color get_pixel(uint32_t x, uint32_t y) const {  // linear colour, index x + y * width (src/graphics.rs:39-46)
    const float* p = &rgb_[((size_t)x + (size_t)y * width_) * 3];
    return {p[0], p[1], p[2]};
  }
  const std::vector<float>& rgb() const { return rgb_; }
  const std::vector<uint8_t>& to_image() const { return rgba8_; }  // RGBA8, the cast of src/graphics.rs:59-74
  float* rgb_data() { return rgb_.data(); }
  uint8_t* rgba8_data() { return rgba8_.data(); }
  void save_to_png(const std::string& path) const;  // 8-bit RGBA, non-interlaced (src/graphics.rs:77-81)
  rtk_stats stats{};

 private:
  uint32_t width_, height_;
  std::vector<float> rgb_;
  std::vector<uint8_t> rgba8_;
};

// One GPU.  There is no CPU path: construction fails when no device can be opened.
class Device {
 public:
  explicit Device(int device_id = 0);
  ~Device();
  Device(const Device&) = delete;
  Device& operator=(const Device&) = delete;
  rtk_context* context() const { return ctx_; }

 private:
  rtk_context* ctx_ = nullptr;
};

class Camera {
 public:
  Camera(uint32_t width, uint32_t height, double field_of_view);  // src/scene/cameras.rs:19-49
  uint32_t width, height;
  double half_width, half_height, field_of_view, pixel_size;
  Matrix4x4 transform;  // public in the reference (cameras.rs:14)
  rtk_camera pack() const;
  // cameras.rs:72-85; samples_per_axis > 1 averages that many squared samples per pixel (the
  // reference has one sample, through the pixel centre: cameras.rs:53-54)
  Canvas render(const Scene& scene, Device& device, uint32_t samples_per_axis = 1) const;
  Camera supersampled(uint32_t samples_per_axis) const;  // the camera whose pixels are this camera's samples
};

// Scene files (src/loader.rs:22-200).  The YAML subset a scene file uses: block mappings and
// sequences, flow sequences of numbers, plain scalars, comments.
class PackedScene {
 public:
  static PackedScene from_yaml_file(const std::string& path);
  static PackedScene from_yaml(const std::string& text);
  Scene build() const;
  size_t n_lights() const { return lights_.size(); }
  size_t n_objects() const { return objects_.size(); }

 private:
  std::vector<PointLight> lights_;
  std::vector<SceneNode> objects_;
};

[[noreturn]] void throw_last(int code, const char* where);

}  // namespace rtkhost

#endif
