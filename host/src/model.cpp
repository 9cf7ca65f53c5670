// model.cpp -- matrices, scene packing, camera and device of the C++ host.
#include <cmath>
#include <cstring>

#include "../include/rtk_host.hpp"

namespace rtkhost {

void throw_last(int code, const char* where) {
  const char* msg = rtk_last_error();
  throw Error(code, std::string(where) + ": " + (msg && *msg ? msg : "error") + " (code " + std::to_string(code) + ")");
}

// ---- Matrix4x4 (src/maths/matrices.rs, src/maths/transforms.rs) -------------------------------
static Matrix4x4 from_rows(std::initializer_list<double> v) {
  Matrix4x4 m;
  int i = 0;
  for (double x : v) m.e[i++] = x;
  return m;
}
Matrix4x4 Matrix4x4::identity() { return from_rows({1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}); }
Matrix4x4 Matrix4x4::translate(double x, double y, double z) { return from_rows({1, 0, 0, x, 0, 1, 0, y, 0, 0, 1, z, 0, 0, 0, 1}); }
Matrix4x4 Matrix4x4::scale(double x, double y, double z) { return from_rows({x, 0, 0, 0, 0, y, 0, 0, 0, 0, z, 0, 0, 0, 0, 1}); }
Matrix4x4 Matrix4x4::rotate_x(double r) {
  const double c = std::cos(r), s = std::sin(r);
  return from_rows({1, 0, 0, 0, 0, c, -s, 0, 0, s, c, 0, 0, 0, 0, 1});
}
Matrix4x4 Matrix4x4::rotate_y(double r) {
  const double c = std::cos(r), s = std::sin(r);
  return from_rows({c, 0, s, 0, 0, 1, 0, 0, -s, 0, c, 0, 0, 0, 0, 1});
}
Matrix4x4 Matrix4x4::rotate_z(double r) {
  const double c = std::cos(r), s = std::sin(r);
  return from_rows({c, -s, 0, 0, s, c, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1});
}
Matrix4x4 Matrix4x4::shear(double x1, double x2, double y1, double y2, double z1, double z2) {
  return from_rows({1, x1, x2, 0, y1, 1, y2, 0, z1, z2, 1, 0, 0, 0, 0, 1});
}
// the reference normalises over all four components (src/maths/vectors.rs:52-66)
static Vector normalize4(Vector v) {
  const double m = std::sqrt(v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w);
  return {v.x / m, v.y / m, v.z / m, v.w / m};
}
static Vector cross3(Vector a, Vector b) { return vec3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
Matrix4x4 Matrix4x4::look_at(Vector from, Vector to, Vector up) {  // src/maths/transforms.rs:67-80
  const Vector forward = normalize4({to.x - from.x, to.y - from.y, to.z - from.z, to.w - from.w});
  const Vector left = cross3(forward, normalize4(up));
  const Vector true_up = cross3(left, forward);
  const Matrix4x4 orientation = from_rows({left.x, left.y, left.z, 0, true_up.x, true_up.y, true_up.z, 0, -forward.x,
                                           -forward.y, -forward.z, 0, 0, 0, 0, 1});
  return orientation * translate(-from.x, -from.y, -from.z);
}
Matrix4x4 Matrix4x4::operator*(const Matrix4x4& rhs) const {  // summation order of src/maths/matrices.rs:93-114
  Matrix4x4 out;
  for (int row = 0; row < 4; ++row)
    for (int col = 0; col < 4; ++col) {
      double s = 0.0;
      for (int i = 0; i < 4; ++i) s += e[i + row * 4] * rhs.e[col + i * 4];
      out.e[col + row * 4] = s;
    }
  return out;
}

// ---- Scene -> rtk_scene ---------------------------------------------------------------------
void Scene::pack(Packed& st) const {
  st.objects.assign(nodes.size(), rtk_object{});
  st.lights.assign(lights.size(), rtk_light{});
  st.inner_transforms.clear();
  st.inner_transforms.reserve(nodes.size());  // the pointers handed out below must stay put
  for (size_t i = 0; i < nodes.size(); ++i) {
    const SceneNode& n = nodes[i];
    rtk_object& o = st.objects[i];
    o.kind = n.kind;
    std::memcpy(o.transform, n.transform.e.data(), sizeof o.transform);
    const double pn[4] = {n.plane_normal.x, n.plane_normal.y, n.plane_normal.z, n.plane_normal.w};
    std::memcpy(o.plane_normal, pn, sizeof pn);
    rtk_material& m = o.material;
    m.texture_kind = n.material.texture.kind;
    const Color a = n.material.texture.a, b = n.material.texture.b;
    m.color_a[0] = a.r, m.color_a[1] = a.g, m.color_a[2] = a.b;
    m.color_b[0] = b.r, m.color_b[1] = b.g, m.color_b[2] = b.b;
    const Texture& tex = n.material.texture;
    m.pattern_transformed = tex.transformed ? 1u + (uint32_t)tex.inner.size() : 0u;
    std::memcpy(m.pattern_transform, tex.transform.e.data(), sizeof m.pattern_transform);
    m.inner_pattern_transforms = nullptr;
    if (tex.transformed && !tex.inner.empty()) {  // the struct points into storage the flattened scene keeps
      st.inner_transforms.emplace_back();
      for (const Matrix4x4& t : tex.inner) st.inner_transforms.back().insert(st.inner_transforms.back().end(), t.e.begin(), t.e.end());
      m.inner_pattern_transforms = st.inner_transforms.back().data();
    }
    m.ambient = n.material.ambient, m.diffuse = n.material.diffuse, m.specular = n.material.specular;
    m.shininess = n.material.shininess, m.transparency = n.material.transparency;
    m.reflectivity = n.material.reflectivity, m.refractivity = n.material.refractivity;
  }
  for (size_t k = 0; k < lights.size(); ++k) {
    const PointLight& l = lights[k];
    rtk_light& d = st.lights[k];
    d.position[0] = l.position.x, d.position[1] = l.position.y, d.position[2] = l.position.z, d.position[3] = l.position.w;
    d.intensity[0] = l.intensity.r, d.intensity[1] = l.intensity.g, d.intensity[2] = l.intensity.b;
  }
  st.scene = rtk_scene{};
  st.scene.objects = st.objects.data();
  st.scene.n_objects = (uint32_t)st.objects.size();
  st.scene.lights = st.lights.data();
  st.scene.n_lights = (uint32_t)st.lights.size();
  st.scene.ambient_color[0] = ambient_color.r, st.scene.ambient_color[1] = ambient_color.g, st.scene.ambient_color[2] = ambient_color.b;
  st.scene.max_depth = max_depth;
}

// ---- Device, Camera -------------------------------------------------------------------------
Device::Device(int device_id) {
  const int rc = rtk_create(device_id, &ctx_);
  if (rc != RTK_OK) throw_last(rc, "rtk_create");
}
Device::~Device() {
  if (ctx_) rtk_destroy(ctx_);
}

Camera::Camera(uint32_t w, uint32_t h, double fov) {
  rtk_camera c{};
  const int rc = rtk_camera_init(&c, w, h, fov);
  if (rc != RTK_OK) throw_last(rc, "rtk_camera_init");
  width = c.width, height = c.height;
  half_width = c.half_width, half_height = c.half_height;
  field_of_view = c.field_of_view, pixel_size = c.pixel_size;
  std::memcpy(transform.e.data(), c.transform, sizeof c.transform);
}
rtk_camera Camera::pack() const {
  rtk_camera c{};
  c.width = width, c.height = height;
  c.half_width = half_width, c.half_height = half_height;
  c.field_of_view = field_of_view, c.pixel_size = pixel_size;
  std::memcpy(c.transform, transform.e.data(), sizeof c.transform);
  return c;
}
Camera Camera::supersampled(uint32_t samples_per_axis) const {
  const rtk_camera c = pack();
  rtk_camera fine{};
  const int rc = rtk_camera_supersample(&c, samples_per_axis, &fine);
  if (rc != RTK_OK) throw_last(rc, "rtk_camera_supersample");
  Camera out = *this;
  out.width = fine.width, out.height = fine.height, out.pixel_size = fine.pixel_size;
  return out;
}
Canvas Camera::render(const Scene& scene, Device& device, uint32_t samples_per_axis) const {
  Scene::Packed packed;
  scene.pack(packed);
  const rtk_camera cam = pack();
  Canvas canvas(width, height);
  const int rc = rtk_render_sampled(device.context(), &packed.scene, &cam, samples_per_axis, canvas.rgb_data(),
                                    canvas.rgba8_data(), &canvas.stats);
  if (rc != RTK_OK) throw_last(rc, "rtk_render");
  return canvas;
}

}  // namespace rtkhost
