// oracle.cpp -- CPU oracle: a literal f64 restatement of the reference's ray-trace loop.
//
// TEST INFRASTRUCTURE, NOT PRODUCT CODE (see oracle.h).  Every function names the
// reference lines it follows.  Arithmetic is kept 4-component and in the reference's
// operation order; build with -ffp-contract=off so no multiply-add is fused (rustc
// never contracts f64 arithmetic).
//
// Pinned by: tests/golden/kat.json (the reference's unit-test expectations) and
// tests/golden/output_stride4.npy (+ sha256 of the full decoded output.png).

#include "oracle.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

namespace {

// src/maths.rs:17
constexpr double EPSILON = 0.00001;

// src/maths.rs:24-28
inline bool is_approx(double a, double b) { return std::fabs(a - b) < EPSILON; }

// ------------------------------------------------------------------ Vector (src/maths/vectors.rs:28-33)
struct Vec {
  double x, y, z, w;
};
inline Vec point(double x, double y, double z) { return {x, y, z, 1.}; }  // vectors.rs:12-14
inline Vec vec3(double x, double y, double z) { return {x, y, z, 0.}; }   // vectors.rs:17-19
inline Vec load4(const double* p) { return {p[0], p[1], p[2], p[3]}; }
inline void store4(double* p, Vec v) {
  p[0] = v.x;
  p[1] = v.y;
  p[2] = v.z;
  p[3] = v.w;
}

// vectors.rs:52-59 -- the magnitude includes w
inline double magnitude(Vec v) {
  double x2 = v.x * v.x, y2 = v.y * v.y, z2 = v.z * v.z, w2 = v.w * v.w;
  return std::sqrt(x2 + y2 + z2 + w2);
}
// vectors.rs:62-71 -- w is divided as well
inline Vec normalize(Vec v) {
  double m = magnitude(v);
  return {v.x / m, v.y / m, v.z / m, v.w / m};
}
// vectors.rs:76-83 -- the dot product includes w
inline double dot(Vec a, Vec b) {
  double x = a.x * b.x, y = a.y * b.y, z = a.z * b.z, w = a.w * b.w;
  return x + y + z + w;
}
// vectors.rs:88-94
inline Vec cross(Vec a, Vec b) {
  return vec3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
inline Vec operator-(Vec a) { return {-a.x, -a.y, -a.z, -a.w}; }                              // vectors.rs:140-151
inline Vec operator+(Vec a, Vec b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }  // vectors.rs:153-164
inline Vec operator-(Vec a, Vec b) { return {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }  // vectors.rs:166-177
inline Vec operator*(Vec a, double s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }       // vectors.rs:179-190
// vectors.rs:97-99: self - normal * 2. * self.dot(normal)
inline Vec reflect(Vec v, Vec n) { return v - (n * 2.) * dot(v, n); }

// ------------------------------------------------------------------ Color (src/maths/colors.rs:14-18)
struct Col {
  double r, g, b;
};
inline Col operator+(Col a, Col b) { return {a.r + b.r, a.g + b.g, a.b + b.b}; }  // colors.rs:40-50
inline Col operator-(Col a, Col b) { return {a.r - b.r, a.g - b.g, a.b - b.b}; }  // colors.rs:52-62
inline Col operator*(Col a, Col b) { return {a.r * b.r, a.g * b.g, a.b * b.b}; }  // colors.rs:64-74
inline Col operator*(Col a, double s) { return {a.r * s, a.g * s, a.b * s}; }     // colors.rs:76-86
constexpr Col BLACK{0., 0., 0.};
inline Col load3(const double* p) { return {p[0], p[1], p[2]}; }

// ------------------------------------------------------------------ Matrix (src/maths/matrices.rs:17-19)
struct Mat4 {
  double e[16];
  double& at(int row, int col) { return e[col + row * 4]; }  // matrices.rs:69-71
  double at(int row, int col) const { return e[col + row * 4]; }
};
inline Mat4 load16(const double* p) {
  Mat4 m;
  std::memcpy(m.e, p, sizeof m.e);
  return m;
}
inline Mat4 identity() {
  Mat4 m{};
  m.at(0, 0) = m.at(1, 1) = m.at(2, 2) = m.at(3, 3) = 1.;
  return m;
}
// matrices.rs:93-114
inline Mat4 mul(const Mat4& a, const Mat4& b) {
  Mat4 r{};
  for (int row = 0; row < 4; ++row)
    for (int col = 0; col < 4; ++col) {
      double sum = 0.;
      for (int i = 0; i < 4; ++i) sum += a.at(row, i) * b.at(i, col);
      r.at(row, col) = sum;
    }
  return r;
}
// matrices.rs:35-45
inline Mat4 transpose(const Mat4& m) {
  Mat4 r{};
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) r.at(i, j) = m.at(j, i);
  return r;
}
// vectors.rs:205-223
inline Vec mul(const Mat4& m, Vec v) {
  double r[4];
  for (int row = 0; row < 4; ++row) {
    double x = m.at(row, 0) * v.x, y = m.at(row, 1) * v.y, z = m.at(row, 2) * v.z, w = m.at(row, 3) * v.w;
    r[row] = x + y + z + w;
  }
  return {r[0], r[1], r[2], r[3]};
}

// matrices.rs:282-286
inline double det2(const double m[4]) { return m[0] * m[3] - m[1] * m[2]; }
// matrices.rs:215-264: sub-matrix, minor, cofactor and determinant of a 3x3
inline double det3(const double m[9]) {
  double result = 0.;
  for (int i = 0; i < 3; ++i) {
    double sub[4];
    int k = 0;
    for (int y = 0; y < 3; ++y) {
      if (y == 0) continue;
      for (int x = 0; x < 3; ++x) {
        if (x == i) continue;
        sub[k++] = m[x + y * 3];
      }
    }
    double minor = det2(sub);
    double cof = ((0 + i) % 2 == 0) ? minor : -minor;
    result += m[i] * cof;
  }
  return result;
}
// matrices.rs:131-167
inline double cofactor4(const Mat4& m, int row, int col) {
  double sub[9];
  int k = 0;
  for (int y = 0; y < 4; ++y) {
    if (y == row) continue;
    for (int x = 0; x < 4; ++x) {
      if (x == col) continue;
      sub[k++] = m.at(y, x);
    }
  }
  double minor = det3(sub);
  return ((row + col) % 2 == 0) ? minor : -minor;
}
// matrices.rs:172-180
inline double det4(const Mat4& m) {
  double result = 0.;
  for (int i = 0; i < 4; ++i) result += m.at(0, i) * cofactor4(m, 0, i);
  return result;
}
// matrices.rs:183-198
inline bool invert(const Mat4& m, Mat4& out) {
  double d = det4(m);
  if (d == 0.) return false;
  for (int row = 0; row < 4; ++row)
    for (int col = 0; col < 4; ++col) out.at(col, row) = cofactor4(m, row, col) / d;
  return true;
}

// src/maths/transforms.rs:7-14
inline Mat4 translate(double x, double y, double z) {
  Mat4 m = identity();
  m.at(0, 3) = x;
  m.at(1, 3) = y;
  m.at(2, 3) = z;
  return m;
}
// transforms.rs:17-24
inline Mat4 scale(double x, double y, double z) {
  Mat4 m = identity();
  m.at(0, 0) = x;
  m.at(1, 1) = y;
  m.at(2, 2) = z;
  return m;
}
// transforms.rs:27-34
inline Mat4 rotate_x(double r) {
  Mat4 m = identity();
  m.at(1, 1) = std::cos(r);
  m.at(1, 2) = -std::sin(r);
  m.at(2, 1) = std::sin(r);
  m.at(2, 2) = std::cos(r);
  return m;
}
// transforms.rs:37-44
inline Mat4 rotate_y(double r) {
  Mat4 m = identity();
  m.at(0, 0) = std::cos(r);
  m.at(0, 2) = std::sin(r);
  m.at(2, 0) = -std::sin(r);
  m.at(2, 2) = std::cos(r);
  return m;
}
// transforms.rs:47-54
inline Mat4 rotate_z(double r) {
  Mat4 m = identity();
  m.at(0, 0) = std::cos(r);
  m.at(0, 1) = -std::sin(r);
  m.at(1, 0) = std::sin(r);
  m.at(1, 1) = std::cos(r);
  return m;
}
// transforms.rs:67-80
inline Mat4 look_at(Vec from, Vec to, Vec up) {
  Vec forward = normalize(to - from);
  Vec left = cross(forward, normalize(up));
  Vec true_up = cross(left, forward);
  Mat4 o = identity();
  o.at(0, 0) = left.x;
  o.at(0, 1) = left.y;
  o.at(0, 2) = left.z;
  o.at(1, 0) = true_up.x;
  o.at(1, 1) = true_up.y;
  o.at(1, 2) = true_up.z;
  o.at(2, 0) = -forward.x;
  o.at(2, 1) = -forward.y;
  o.at(2, 2) = -forward.z;
  return mul(o, translate(-from.x, -from.y, -from.z));
}

// ------------------------------------------------------------------ Ray (src/maths/rays.rs:11-14)
struct Ray {
  Vec origin, direction;
};
inline Vec position(const Ray& r, double t) { return r.origin + r.direction * t; }  // rays.rs:26-28
inline Ray mul(const Mat4& m, const Ray& r) { return {mul(m, r.origin), mul(m, r.direction)}; }  // rays.rs:31-41

// ------------------------------------------------------------------ patterns (src/graphics/patterns.rs)
// Rust's `%` on f64 is C's fmod (sign of the dividend).
inline Col sample_at(const rtk_material& m, Vec p) {  // src/scene/materials.rs:14-19
  // `TransformPattern::sample_at` (src/graphics/patterns.rs:27-35): inverted per sample; the type is generic
  // over its pattern, so a TransformPattern may hold another: the outermost acts first, then each inner one
  for (uint32_t level = 0; level < m.pattern_transformed; ++level) {
    Mat4 inverse;
    if (invert(load16(level == 0 ? m.pattern_transform : m.inner_pattern_transforms + 16 * (size_t)(level - 1)), inverse))
      p = mul(inverse, p);
  }
  Col a = load3(m.color_a), b = load3(m.color_b);
  switch (m.texture_kind) {
    case RTK_TEX_SOLID:
      return a;
    case RTK_TEX_STRIPED:  // patterns.rs:57-65
      return (std::fmod(std::floor(p.x), 2.) == 0.) ? a : b;
    case RTK_TEX_GRADIENT: {  // patterns.rs:81-88
      Col distance = b - a;
      double fraction = p.x - std::floor(p.x);
      return a + distance * fraction;
    }
    case RTK_TEX_RING: {  // patterns.rs:105-116
      double x2 = p.x * p.x, z2 = p.z * p.z;
      return (std::fmod(std::floor(std::sqrt(x2 + z2)), 2.) == 0.) ? a : b;
    }
    case RTK_TEX_CHECKER:  // patterns.rs:132-140
      return (std::fmod(std::floor(p.x) + std::floor(p.y) + std::floor(p.z), 2.) == 0.) ? a : b;
  }
  return a;
}

// ------------------------------------------------------------------ scene objects
// `SceneNode<S>` (src/scene.rs:36-99) with its pre-computed inverse.
struct Node {
  const rtk_object* src;
  Mat4 transform, inverse;
};
inline Node make_node(const rtk_object* o) {
  Node n;
  n.src = o;
  n.transform = load16(o->transform);
  // scene.rs:55-62: a singular transform silently keeps an identity inverse
  if (!invert(n.transform, n.inverse)) n.inverse = identity();
  return n;
}

struct Hit {
  uint32_t object;
  double distance;
};
// scene.rs:357-364: same object and approximately the same distance
inline bool hit_eq(const Hit& a, const Hit& b) { return is_approx(a.distance, b.distance) && a.object == b.object; }

// spheres.rs:18-35 and planes.rs:22-28, behind SceneNode::intersect (scene.rs:75-84)
inline int node_intersect(const Node& n, const Ray& world_ray, double out[2]) {
  Ray r = mul(n.inverse, world_ray);
  if (n.src->kind == RTK_PLANE) {
    if (std::fabs(r.direction.y) < 0.00001) return 0;
    out[0] = -r.origin.y / r.direction.y;
    return 1;
  }
  Vec sphere_to_ray = r.origin - point(0., 0., 0.);
  double a = dot(r.direction, r.direction);
  double b = 2. * dot(sphere_to_ray, r.direction);
  double c = dot(sphere_to_ray, sphere_to_ray) - 1.;
  double discriminant = b * b - 4. * a * c;
  if (discriminant >= 0.) {
    out[0] = (-b - std::sqrt(discriminant)) / (2. * a);
    out[1] = (-b + std::sqrt(discriminant)) / (2. * a);
    return 2;
  }
  return 0;
}

// scene.rs:86-90 with spheres.rs:37-44 / planes.rs:30-32
inline Vec node_normal_at(const Node& n, Vec world_point) {
  Vec object_point = mul(n.inverse, world_point);
  if (n.src->kind == RTK_PLANE) return load4(n.src->plane_normal);  // never transformed
  Vec object_normal = object_point - point(0., 0., 0.);
  Vec world_normal = mul(transpose(n.inverse), object_normal);
  world_normal.w = 0.;
  return normalize(world_normal);
}

struct Counters {
  uint64_t primary = 0, shadow = 0, secondary = 0;
};

// `Scene` (src/scene.rs:102-296)
struct Scene {
  Col ambient_color;
  std::vector<Node> nodes;
  const rtk_light* lights;
  uint32_t n_lights;
  uint32_t max_depth;
  bool nan_seen = false;

  explicit Scene(const rtk_scene* s) {
    ambient_color = load3(s->ambient_color);
    nodes.reserve(s->n_objects);
    for (uint32_t i = 0; i < s->n_objects; ++i) nodes.push_back(make_node(&s->objects[i]));
    lights = s->lights;
    n_lights = s->n_lights;
    max_depth = s->max_depth;
  }

  // scene.rs:156-169
  void intersect(const Ray& ray, std::vector<Hit>& hits) {
    hits.clear();
    double t[2];
    for (uint32_t i = 0; i < nodes.size(); ++i) {
      int n = node_intersect(nodes[i], ray, t);
      for (int k = 0; k < n; ++k) hits.push_back({i, t[k]});
    }
    for (const Hit& h : hits)
      if (std::isnan(h.distance)) {  // the reference panics here (partial_cmp().unwrap())
        nan_seen = true;
        return;
      }
    // Rust's sort_by is stable; so is an insertion sort and std::stable_sort.
    if (hits.size() <= 24) {
      for (size_t i = 1; i < hits.size(); ++i) {
        Hit key = hits[i];
        size_t j = i;
        while (j > 0 && key.distance < hits[j - 1].distance) {
          hits[j] = hits[j - 1];
          --j;
        }
        hits[j] = key;
      }
    } else {
      std::stable_sort(hits.begin(), hits.end(), [](const Hit& a, const Hit& b) { return a.distance < b.distance; });
    }
  }
};

// scene.rs:325-340
inline int64_t closest_hit(const Hit* hits, size_t n, double* closest_out) {
  double closest_t = DBL_MAX;
  int64_t closest = -1;
  for (size_t i = 0; i < n; ++i) {
    double t = hits[i].distance;
    if (t > 0. && t < closest_t) {
      closest_t = t;
      closest = (int64_t)i;
    }
  }
  if (closest_out) *closest_out = closest_t;
  return closest;
}

// `LightingData` (src/scene/lighting.rs:24-36)
struct LightingData {
  uint32_t object;
  Vec world_position, over_position, under_position, object_position, eye, normal, reflect_direction;
  double distance;
  bool inside;
  double refractivity[2];
};

// lighting.rs:81-116
inline void compute_refractivity(const Scene& sc, const Hit& hit, const Hit* hits, size_t n, double out[2]) {
  double n1 = 0., n2 = 0.;
  std::vector<const Hit*> containers;
  for (size_t k = 0; k < n; ++k) {
    const Hit* i = &hits[k];
    if (hit_eq(*i, hit))
      n1 = containers.empty() ? 1. : sc.nodes[containers.back()->object].src->material.refractivity;
    bool contained = false;
    for (const Hit* it : containers)
      if (hit_eq(*it, *i)) {
        contained = true;
        break;
      }
    if (contained)
      containers.erase(
          std::remove_if(containers.begin(), containers.end(), [&](const Hit* it) { return hit_eq(*it, *i); }),
          containers.end());
    else
      containers.push_back(i);
    if (hit_eq(*i, hit)) {
      n2 = containers.empty() ? 1. : sc.nodes[containers.back()->object].src->material.refractivity;
      break;
    }
  }
  out[0] = n1;
  out[1] = n2;
}

// lighting.rs:40-78
inline LightingData lighting_data(const Scene& sc, const Ray& ray, const Hit& hit, const Hit* hits, size_t n) {
  LightingData d;
  const Node& object = sc.nodes[hit.object];
  d.object = hit.object;
  d.world_position = position(ray, hit.distance);
  d.eye = -ray.direction;
  d.distance = hit.distance;
  Vec normal = node_normal_at(object, d.world_position);
  // the offsets, the texture point and the reflection use the normal BEFORE the flip
  d.over_position = d.world_position + normal * 0.0001;
  d.under_position = d.world_position - normal * 0.0001;
  d.object_position = mul(object.inverse, d.over_position);  // scene.rs:92-94
  d.reflect_direction = reflect(ray.direction, normal);
  d.inside = false;
  if (dot(normal, d.eye) < 0.) {
    normal = -normal;
    d.inside = true;
  }
  d.normal = normal;
  compute_refractivity(sc, hit, hits, n, d.refractivity);
  return d;
}

// lighting.rs:120-154
inline Col phong_lighting(const rtk_light& light, const rtk_material& material, Vec world_position,
                          Vec object_position, Vec eye, Vec normal, bool in_shadow) {
  Col intensity = load3(light.intensity);
  Col effective_color = sample_at(material, object_position) * intensity;
  Vec light_direction = normalize(load4(light.position) - world_position);
  Col ambient = effective_color * material.ambient;
  Col diffuse = BLACK, specular = BLACK;
  double light_dot_normal = dot(light_direction, normal);
  if (light_dot_normal >= 0.) {
    diffuse = effective_color * material.diffuse * light_dot_normal;
    Vec reflect_direction = -reflect(light_direction, normal);
    double reflect_dot_eye = dot(reflect_direction, eye);
    if (reflect_dot_eye >= 0.) {
      double factor = std::pow(reflect_dot_eye, material.shininess);
      specular = intensity * material.specular * factor;
    }
  }
  if (in_shadow) return ambient;
  return ambient + diffuse + specular;
}

struct Tracer {
  Scene& sc;
  Counters& cnt;

  Col trace_inner(const Ray& ray, uint32_t depth);

  // scene.rs:208-225
  bool is_shadowed(Vec p) {
    std::vector<Hit> hits;
    for (uint32_t k = 0; k < sc.n_lights; ++k) {
      Vec light_vector = load4(sc.lights[k].position) - p;
      double distance = magnitude(light_vector);
      Vec direction = normalize(light_vector);
      Ray ray{p, direction};
      ++cnt.shadow;
      sc.intersect(ray, hits);
      double t;
      if (closest_hit(hits.data(), hits.size(), &t) >= 0)
        if (t < distance) return true;
    }
    return false;
  }

  // scene.rs:228-241
  Col reflected_color(const LightingData& d, uint32_t depth) {
    const rtk_material& material = sc.nodes[d.object].src->material;
    if (is_approx(material.reflectivity, 0.)) return BLACK;
    Ray reflect_ray{d.over_position, d.reflect_direction};
    return trace_inner(reflect_ray, depth + 1) * material.reflectivity;
  }

  // scene.rs:244-272
  Col refracted_color(const LightingData& d, uint32_t depth) {
    if (depth >= sc.max_depth) return BLACK;
    const rtk_material& material = sc.nodes[d.object].src->material;
    double n1 = d.refractivity[0], n2 = d.refractivity[1];
    if (is_approx(material.transparency, 0.)) return BLACK;
    double n_ratio = n1 / n2;
    double cos_i = dot(d.eye, d.normal);
    double sin_t2 = n_ratio * n_ratio * (1. - cos_i * cos_i);
    if (sin_t2 > 1.) return BLACK;
    double cos_t = std::sqrt(1. - sin_t2);
    // note the `+`: normal * (n_ratio + cos_i - cos_t), as the reference has it
    Vec direction = d.normal * (n_ratio + cos_i - cos_t) - d.eye * n_ratio;
    Ray ray{d.under_position, direction};
    Col color = trace_inner(ray, depth + 1);
    return color * material.transparency;
  }

  // scene.rs:172-205
  Col apply_lighting(const Ray& ray, const Hit& hit, const Hit* hits, size_t n, uint32_t depth) {
    Col surface = sc.ambient_color;
    LightingData d = lighting_data(sc, ray, hit, hits, n);
    bool in_shadow = is_shadowed(d.over_position);
    const rtk_material& material = sc.nodes[d.object].src->material;
    for (uint32_t k = 0; k < sc.n_lights; ++k)
      surface = surface +
                phong_lighting(sc.lights[k], material, d.over_position, d.object_position, d.eye, d.normal, in_shadow);
    Col reflected = reflected_color(d, depth);
    Col refracted = refracted_color(d, depth);
    if (material.reflectivity > 0. && material.transparency > 0.) {
      double reflectance = shlick(d);
      return surface + reflected * reflectance + refracted * (1. - reflectance);
    }
    return surface + reflected + refracted;
  }

  // scene.rs:275-295
  static double shlick(const LightingData& d) {
    double n1 = d.refractivity[0], n2 = d.refractivity[1];
    double cosv = dot(d.eye, d.normal);
    if (n1 > n2) {
      double n = n1 / n2;
      double sin_t2 = n * n * (1. - cosv * cosv);
      if (sin_t2 > 1.) return 1.;
      double cos_t = std::sqrt(1. - sin_t2);
      cosv = cos_t;
    }
    double r0 = (n1 - n2) / (n1 + n2);
    double r02 = r0 * r0;
    double x = 1. - cosv;
    double x2 = x * x;
    double x5 = x * (x2 * x2);  // powi(5) by repeated squaring
    return r02 + (1. - r02) * x5;
  }
};

// scene.rs:141-153
Col Tracer::trace_inner(const Ray& ray, uint32_t depth) {
  if (depth >= sc.max_depth) return sc.ambient_color;
  if (depth == 0)
    ++cnt.primary;
  else
    ++cnt.secondary;
  std::vector<Hit> hits;
  sc.intersect(ray, hits);
  double t;
  int64_t k = closest_hit(hits.data(), hits.size(), &t);
  if (k >= 0) {
    Hit hit{hits[(size_t)k].object, t};
    return apply_lighting(ray, hit, hits.data(), hits.size(), depth);
  }
  return sc.ambient_color;
}

// src/scene/cameras.rs:19-49
inline void camera_new(rtk_camera* c, uint32_t width, uint32_t height, double fov) {
  double half_view = std::tan(fov / 2.);
  double aspect = (double)width / (double)height;
  double half_width, half_height;
  if (aspect >= 1.) {
    half_width = half_view;
    half_height = half_width / aspect;
  } else {
    half_width = half_view * aspect;
    half_height = half_view;
  }
  c->width = width;
  c->height = height;
  c->half_width = half_width;
  c->half_height = half_height;
  c->field_of_view = fov;
  c->pixel_size = (half_width * 2.) / (double)width;
  Mat4 t = look_at(point(0., 1.5, -5.), point(0., 1., 0.), vec3(0., 1., 0.));
  std::memcpy(c->transform, t.e, sizeof t.e);
}

// cameras.rs:52-69 -- the reference inverts the camera matrix for every pixel; so does this
inline bool ray_for_pixel(const rtk_camera& c, uint64_t x, uint64_t y, Ray& out) {
  double x_offset = ((double)x + 0.5) * c.pixel_size;
  double y_offset = ((double)y + 0.5) * c.pixel_size;
  double world_x = c.half_width - x_offset;
  double world_y = c.half_height - y_offset;
  Mat4 inverse;
  if (!invert(load16(c.transform), inverse)) return false;
  Vec pixel = mul(inverse, point(world_x, world_y, -1.));
  Vec origin = mul(inverse, point(0., 0., 0.));
  Vec direction = normalize(pixel - origin);
  out = {origin, direction};
  return true;
}

// src/graphics.rs:65-69: Rust's `as u8` truncates toward zero, saturates, and maps NaN to 0
inline uint8_t cast_u8(double v) {
  if (std::isnan(v)) return 0;
  if (v <= 0.) return 0;
  if (v >= 255.) return 255;
  return (uint8_t)v;
}

inline std::vector<Hit> to_hits(const oracle_hit* h, uint64_t n) {
  std::vector<Hit> v(n);
  for (uint64_t i = 0; i < n; ++i) v[i] = {h[i].object, h[i].distance};
  return v;
}
inline void export_data(const LightingData& d, oracle_lighting_data* o) {
  o->object = d.object;
  o->inside = d.inside ? 1u : 0u;
  store4(o->world_position, d.world_position);
  store4(o->over_position, d.over_position);
  store4(o->under_position, d.under_position);
  store4(o->object_position, d.object_position);
  store4(o->eye, d.eye);
  store4(o->normal, d.normal);
  store4(o->reflect_direction, d.reflect_direction);
  o->distance = d.distance;
  o->refractivity[0] = d.refractivity[0];
  o->refractivity[1] = d.refractivity[1];
}
inline LightingData import_data(const oracle_lighting_data* o) {
  LightingData d;
  d.object = o->object;
  d.inside = o->inside != 0;
  d.world_position = load4(o->world_position);
  d.over_position = load4(o->over_position);
  d.under_position = load4(o->under_position);
  d.object_position = load4(o->object_position);
  d.eye = load4(o->eye);
  d.normal = load4(o->normal);
  d.reflect_direction = load4(o->reflect_direction);
  d.distance = o->distance;
  d.refractivity[0] = o->refractivity[0];
  d.refractivity[1] = o->refractivity[1];
  return d;
}

}  // namespace

extern "C" {

int oracle_matrix_invert(const double m[16], double out[16]) {
  Mat4 r;
  if (!invert(load16(m), r)) return 0;
  std::memcpy(out, r.e, sizeof r.e);
  return 1;
}
void oracle_matrix_mul(const double a[16], const double b[16], double out[16]) {
  Mat4 r = mul(load16(a), load16(b));
  std::memcpy(out, r.e, sizeof r.e);
}
void oracle_matrix_mul_vector(const double m[16], const double v[4], double out[4]) {
  store4(out, mul(load16(m), load4(v)));
}
void oracle_matrix_transpose(const double m[16], double out[16]) {
  Mat4 r = transpose(load16(m));
  std::memcpy(out, r.e, sizeof r.e);
}
void oracle_translate(double x, double y, double z, double out[16]) { std::memcpy(out, translate(x, y, z).e, 128); }
void oracle_scale(double x, double y, double z, double out[16]) { std::memcpy(out, scale(x, y, z).e, 128); }
void oracle_rotate_x(double r, double out[16]) { std::memcpy(out, rotate_x(r).e, 128); }
void oracle_rotate_y(double r, double out[16]) { std::memcpy(out, rotate_y(r).e, 128); }
void oracle_rotate_z(double r, double out[16]) { std::memcpy(out, rotate_z(r).e, 128); }
void oracle_look_at(const double from[4], const double to[4], const double up[4], double out[16]) {
  std::memcpy(out, look_at(load4(from), load4(to), load4(up)).e, 128);
}
void oracle_reflect(const double v[4], const double normal[4], double out[4]) {
  store4(out, reflect(load4(v), load4(normal)));
}
uint8_t oracle_cast_u8(double c) { return cast_u8(c * 255.0); }

int oracle_object_intersect(const rtk_object* object, const double origin[4], const double direction[4],
                            double distances[2]) {
  Node n = make_node(object);
  return node_intersect(n, Ray{load4(origin), load4(direction)}, distances);
}
void oracle_object_normal_at(const rtk_object* object, const double world_point[4], double out[4]) {
  Node n = make_node(object);
  store4(out, node_normal_at(n, load4(world_point)));
}
void oracle_texture_sample_at(const rtk_material* material, const double p[4], double out[3]) {
  Col c = sample_at(*material, load4(p));
  out[0] = c.r;
  out[1] = c.g;
  out[2] = c.b;
}
int oracle_camera_init(rtk_camera* camera, uint32_t width, uint32_t height, double fov) {
  if (!camera || width == 0 || height == 0) return -1;
  camera_new(camera, width, height, fov);
  return 0;
}
int oracle_camera_ray_for_pixel(const rtk_camera* camera, uint64_t x, uint64_t y, double origin[4],
                                double direction[4]) {
  Ray r;
  if (!ray_for_pixel(*camera, x, y, r)) return 0;
  store4(origin, r.origin);
  store4(direction, r.direction);
  return 1;
}

uint64_t oracle_scene_intersect(const rtk_scene* scene, const double origin[4], const double direction[4],
                                oracle_hit* hits, uint64_t capacity) {
  Scene sc(scene);
  std::vector<Hit> v;
  sc.intersect(Ray{load4(origin), load4(direction)}, v);
  for (uint64_t i = 0; i < v.size() && i < capacity; ++i) {
    hits[i].object = v[i].object;
    hits[i].reserved_ = 0;
    hits[i].distance = v[i].distance;
  }
  return v.size();
}
int64_t oracle_closest_hit(const oracle_hit* hits, uint64_t n) {
  std::vector<Hit> v = to_hits(hits, n);
  return closest_hit(v.data(), v.size(), nullptr);
}
int oracle_is_shadowed(const rtk_scene* scene, const double p[4]) {
  Scene sc(scene);
  Counters c;
  Tracer t{sc, c};
  return t.is_shadowed(load4(p)) ? 1 : 0;
}
void oracle_phong_lighting(const rtk_light* light, const rtk_material* material, const double world_position[4],
                           const double object_position[4], const double eye[4], const double normal[4],
                           int in_shadow, double out[3]) {
  Col c = phong_lighting(*light, *material, load4(world_position), load4(object_position), load4(eye),
                         load4(normal), in_shadow != 0);
  out[0] = c.r;
  out[1] = c.g;
  out[2] = c.b;
}
void oracle_lighting_data_calculate(const rtk_scene* scene, const double origin[4], const double direction[4],
                                    const oracle_hit* hits, uint64_t n_hits, uint64_t which,
                                    oracle_lighting_data* out) {
  Scene sc(scene);
  std::vector<Hit> v = to_hits(hits, n_hits);
  LightingData d = lighting_data(sc, Ray{load4(origin), load4(direction)}, v[which], v.data(), v.size());
  export_data(d, out);
}
void oracle_apply_lighting(const rtk_scene* scene, const double origin[4], const double direction[4],
                           const oracle_hit* hits, uint64_t n_hits, uint64_t which, uint32_t depth,
                           double out[3]) {
  Scene sc(scene);
  Counters c;
  Tracer t{sc, c};
  std::vector<Hit> v = to_hits(hits, n_hits);
  Col col = t.apply_lighting(Ray{load4(origin), load4(direction)}, v[which], v.data(), v.size(), depth);
  out[0] = col.r;
  out[1] = col.g;
  out[2] = col.b;
}
void oracle_reflected_color(const rtk_scene* scene, const oracle_lighting_data* data, uint32_t depth, double out[3]) {
  Scene sc(scene);
  Counters c;
  Tracer t{sc, c};
  Col col = t.reflected_color(import_data(data), depth);
  out[0] = col.r;
  out[1] = col.g;
  out[2] = col.b;
}
void oracle_refracted_color(const rtk_scene* scene, const oracle_lighting_data* data, uint32_t depth, double out[3]) {
  Scene sc(scene);
  Counters c;
  Tracer t{sc, c};
  Col col = t.refracted_color(import_data(data), depth);
  out[0] = col.r;
  out[1] = col.g;
  out[2] = col.b;
}
double oracle_shlick(const oracle_lighting_data* data) { return Tracer::shlick(import_data(data)); }

void oracle_trace(const rtk_scene* scene, const double origin[4], const double direction[4], double out[3]) {
  Scene sc(scene);
  Counters c;
  Tracer t{sc, c};
  Col col = t.trace_inner(Ray{load4(origin), load4(direction)}, 0);  // scene.rs:136-138
  out[0] = col.r;
  out[1] = col.g;
  out[2] = col.b;
}

// cameras.rs:72-85 and graphics.rs:44-46, 59-74
int oracle_render(const rtk_scene* scene, const rtk_camera* camera, uint32_t x0, uint32_t y0, uint32_t x1,
                  uint32_t y1, double* rgb_f64, uint8_t* rgba8, rtk_stats* stats, int threads) {
  if (!scene || !camera || x1 > camera->width || y1 > camera->height || x0 > x1 || y0 > y1) return -1;
  Mat4 probe;
  if (!invert(load16(camera->transform), probe)) return -2;
  const uint32_t w = x1 - x0, h = y1 - y0;
  if (threads < 1) threads = 1;
  if ((uint32_t)threads > h && h > 0) threads = (int)h;
  std::vector<Counters> counters((size_t)threads);
  auto work = [&](int tid) {
    Scene sc(scene);
    Tracer tr{sc, counters[(size_t)tid]};
    // rows are dealt round-robin so that every thread sees sky and ground alike
    for (uint32_t y = y0 + (uint32_t)tid; y < y1; y += (uint32_t)threads)
      for (uint32_t x = x0; x < x1; ++x) {
        Ray ray;
        ray_for_pixel(*camera, x, y, ray);
        Col c = tr.trace_inner(ray, 0);
        size_t i = (size_t)(x - x0) + (size_t)(y - y0) * w;
        if (rgb_f64) {
          rgb_f64[3 * i + 0] = c.r;
          rgb_f64[3 * i + 1] = c.g;
          rgb_f64[3 * i + 2] = c.b;
        }
        if (rgba8) {
          rgba8[4 * i + 0] = cast_u8(c.r * 255.0);
          rgba8[4 * i + 1] = cast_u8(c.g * 255.0);
          rgba8[4 * i + 2] = cast_u8(c.b * 255.0);
          rgba8[4 * i + 3] = 255;
        }
      }
  };
  if (threads == 1) {
    work(0);
  } else {
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; ++t) pool.emplace_back(work, t);
    for (auto& t : pool) t.join();
  }
  if (stats) {
    std::memset(stats, 0, sizeof *stats);
    for (const Counters& c : counters) {
      stats->rays_primary += c.primary;
      stats->rays_shadow += c.shadow;
      stats->rays_secondary += c.secondary;
    }
  }
  return 0;
}

// More than one sample per pixel.  The reference has none of this (one ray through each
// pixel centre, cameras.rs:53-54); the definition restated here is the one of include/rtk.h:
// the samples of pixel (x, y) are the pixels (s x + i, s y + j) of the same camera at s times
// the resolution (pixel_size / s), each traced as the reference traces a pixel; their colours
// are narrowed to FP32, summed in f64 in row-major order and divided by s*s.
int oracle_render_sampled(const rtk_scene* scene, const rtk_camera* camera, uint32_t s, uint32_t x0, uint32_t y0,
                          uint32_t x1, uint32_t y1, double* rgb_f64, uint8_t* rgba8, rtk_stats* stats, int threads) {
  if (!scene || !camera || s == 0 || x1 > camera->width || y1 > camera->height || x0 > x1 || y0 > y1) return -1;
  rtk_camera fine = *camera;
  fine.width = camera->width * s;
  fine.height = camera->height * s;
  fine.pixel_size = camera->pixel_size / (double)s;
  const uint32_t w = x1 - x0, h = y1 - y0;
  std::vector<double> samples((size_t)w * s * h * s * 3);
  const int rc = oracle_render(scene, &fine, x0 * s, y0 * s, x1 * s, y1 * s, samples.data(), nullptr, stats, threads);
  if (rc != 0) return rc;
  const size_t row = (size_t)w * s * 3;
  for (uint32_t y = 0; y < h; ++y)
    for (uint32_t x = 0; x < w; ++x) {
      double sum[3] = {0.0, 0.0, 0.0};
      for (uint32_t j = 0; j < s; ++j)
        for (uint32_t i = 0; i < s; ++i)
          for (int c = 0; c < 3; ++c)
            sum[c] += (double)(float)samples[((size_t)y * s + j) * row + ((size_t)x * s + i) * 3 + (size_t)c];
      const size_t p = (size_t)x + (size_t)y * w;
      for (int c = 0; c < 3; ++c) {
        const double mean = sum[c] / (double)(s * s);
        if (rgb_f64) rgb_f64[3 * p + (size_t)c] = mean;
        if (rgba8) rgba8[4 * p + (size_t)c] = cast_u8(mean * 255.0);
      }
      if (rgba8) rgba8[4 * p + 3] = 255;
    }
  return 0;
}

}  // extern "C"
