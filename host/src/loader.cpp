// loader.cpp -- scene files of the reference (src/loader.rs:22-200) for the C++ host.
//
// The reference deserialises with serde_yaml; this is a reader for the YAML subset scene
// files use -- block mappings and sequences by indentation, flow sequences `[a, b, c]`
// (nested, possibly spanning lines), plain or quoted scalars, `#` comments -- followed by
// the same field rules: `lights` and `objects` are required, every Option<> field may be
// absent or null, unknown fields are ignored, textures are single-key mappings
// (`solid: [r,g,b]` or `checker|gradient|ring|striped: [[r,g,b],[r,g,b]]`).
#include <cmath>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <utility>

#include "../include/rtk_host.hpp"

namespace rtkhost {
namespace {

struct Value {
  enum Kind { Null, Scalar, Seq, Map } kind = Null;
  std::string scalar;
  std::vector<Value> seq;
  std::vector<std::pair<std::string, Value>> map;
  int line = 0;
  const Value* get(const std::string& key) const {
    for (const auto& kv : map)
      if (kv.first == key) return &kv.second;
    return nullptr;
  }
};

struct Line {
  int indent;
  std::string text;
  int number;
};

[[noreturn]] void bad(int line, const std::string& what) {
  throw Error(RTK_ERR_INVALID, "scene file, line " + std::to_string(line) + ": " + what);
}

std::string trim(const std::string& s) {
  size_t a = 0, b = s.size();
  while (a < b && (s[a] == ' ' || s[a] == '\t' || s[a] == '\r')) ++a;
  while (b > a && (s[b - 1] == ' ' || s[b - 1] == '\t' || s[b - 1] == '\r')) --b;
  return s.substr(a, b - a);
}

// removes a trailing comment: '#' at the start or after white space, outside quotes
std::string strip_comment(const std::string& s) {
  char quote = 0;
  for (size_t i = 0; i < s.size(); ++i) {
    const char c = s[i];
    if (quote) {
      if (c == quote) quote = 0;
    } else if (c == '"' || c == '\'') {
      quote = c;
    } else if (c == '#' && (i == 0 || s[i - 1] == ' ' || s[i - 1] == '\t')) {
      return s.substr(0, i);
    }
  }
  return s;
}

int bracket_balance(const std::string& s) {
  int depth = 0;
  for (char c : s) depth += (c == '[') - (c == ']');
  return depth;
}

std::vector<Line> split_lines(const std::string& text) {
  std::vector<Line> lines;
  std::istringstream in(text);
  std::string raw;
  int number = 0;
  while (std::getline(in, raw)) {
    ++number;
    std::string s = strip_comment(raw);
    if (trim(s).empty() || trim(s) == "---") continue;
    int indent = 0;
    while (indent < (int)s.size() && s[indent] == ' ') ++indent;
    if (indent < (int)s.size() && s[indent] == '\t') bad(number, "tab used for indentation");
    std::string body = trim(s);
    // a flow sequence may run over several lines: join until the brackets balance
    int depth = bracket_balance(body);
    const int first = number;
    while (depth > 0 && std::getline(in, raw)) {
      ++number;
      const std::string more = trim(strip_comment(raw));
      body += " " + more;
      depth += bracket_balance(more);
    }
    if (depth != 0) bad(first, "unbalanced '[' ']'");
    lines.push_back({indent, body, first});
  }
  return lines;
}

// position of the ':' that ends a mapping key (followed by a blank or the end of the line,
// outside brackets and quotes), or npos
size_t key_colon(const std::string& s) {
  if (s.empty() || s[0] == '[' || s[0] == '-') return std::string::npos;
  char quote = 0;
  int depth = 0;
  for (size_t i = 0; i < s.size(); ++i) {
    const char c = s[i];
    if (quote) {
      if (c == quote) quote = 0;
    } else if (c == '"' || c == '\'') {
      quote = c;
    } else if (c == '[') {
      ++depth;
    } else if (c == ']') {
      --depth;
    } else if (c == ':' && depth == 0 && (i + 1 == s.size() || s[i + 1] == ' ')) {
      return i;
    }
  }
  return std::string::npos;
}

std::string unquote(const std::string& s) {
  if (s.size() >= 2 && (s.front() == '"' || s.front() == '\'') && s.back() == s.front()) return s.substr(1, s.size() - 2);
  return s;
}

Value parse_inline(const std::string& text, int line) {
  const std::string s = trim(text);
  Value v;
  v.line = line;
  if (s.empty() || s == "~" || s == "null" || s == "Null" || s == "NULL") return v;
  if (s[0] == '[') {
    if (s.back() != ']') bad(line, "flow sequence does not end with ']'");
    v.kind = Value::Seq;
    int depth = 0;
    std::string item;
    bool any = false;
    for (size_t i = 1; i + 1 < s.size(); ++i) {
      const char c = s[i];
      if (c == '[') ++depth;
      if (c == ']') --depth;
      if (c == ',' && depth == 0) {
        v.seq.push_back(parse_inline(item, line));
        item.clear();
        any = true;
      } else {
        item += c;
      }
    }
    if (!trim(item).empty() || any) {
      if (trim(item).empty()) bad(line, "empty item in flow sequence");
      v.seq.push_back(parse_inline(item, line));
    }
    return v;
  }
  if (s[0] == '{') bad(line, "flow mappings are not supported");
  v.kind = Value::Scalar;
  v.scalar = unquote(s);
  return v;
}

class Parser {
 public:
  explicit Parser(std::vector<Line> lines) : lines_(std::move(lines)) {}
  Value document() {
    if (lines_.empty()) return Value{};
    size_t i = 0;
    Value v = block(i, lines_[0].indent);
    if (i != lines_.size()) bad(lines_[i].number, "unexpected indentation");
    return v;
  }

 private:
  std::vector<Line> lines_;

  static bool is_item(const std::string& s) { return s == "-" || (s.size() >= 2 && s[0] == '-' && s[1] == ' '); }

  Value block(size_t& i, int indent) {
    const Line& l = lines_[i];
    if (is_item(l.text)) return sequence(i, indent);
    if (key_colon(l.text) != std::string::npos) return mapping(i, indent);
    Value v = parse_inline(l.text, l.number);
    ++i;
    return v;
  }

  Value sequence(size_t& i, int indent) {
    Value v;
    v.kind = Value::Seq;
    v.line = lines_[i].number;
    while (i < lines_.size() && lines_[i].indent == indent && is_item(lines_[i].text)) {
      const std::string text = lines_[i].text;
      size_t skip = 1;
      while (skip < text.size() && text[skip] == ' ') ++skip;
      const std::string rest = text.substr(skip);
      if (rest.empty()) {
        ++i;
        if (i < lines_.size() && lines_[i].indent > indent)
          v.seq.push_back(block(i, lines_[i].indent));
        else
          v.seq.push_back(Value{});
      } else {
        // the item starts on the dash's line: treat what follows the dash as a line of its own
        lines_[i].indent = indent + (int)skip;
        lines_[i].text = rest;
        v.seq.push_back(block(i, lines_[i].indent));
      }
    }
    if (i < lines_.size() && lines_[i].indent > indent) bad(lines_[i].number, "unexpected indentation");
    return v;
  }

  Value mapping(size_t& i, int indent) {
    Value v;
    v.kind = Value::Map;
    v.line = lines_[i].number;
    while (i < lines_.size() && lines_[i].indent == indent) {
      const Line l = lines_[i];
      const size_t colon = key_colon(l.text);
      if (colon == std::string::npos) bad(l.number, "expected `key: value`");
      const std::string key = unquote(trim(l.text.substr(0, colon)));
      const std::string rest = trim(l.text.substr(colon + 1));
      ++i;
      Value child;
      if (!rest.empty()) {
        child = parse_inline(rest, l.number);
      } else if (i < lines_.size() && lines_[i].indent > indent) {
        child = block(i, lines_[i].indent);
      } else if (i < lines_.size() && lines_[i].indent == indent && is_item(lines_[i].text)) {
        child = sequence(i, indent);  // a sequence may sit at its key's indentation
      }
      child.line = child.line ? child.line : l.number;
      v.map.emplace_back(key, std::move(child));
    }
    if (i < lines_.size() && lines_[i].indent > indent) bad(lines_[i].number, "unexpected indentation");
    return v;
  }
};

double number(const Value& v, const char* what) {
  if (v.kind != Value::Scalar) bad(v.line, std::string(what) + ": expected a number");
  const std::string& s = v.scalar;
  if (s == ".inf" || s == "+.inf" || s == ".Inf") return INFINITY;
  if (s == "-.inf" || s == "-.Inf") return -INFINITY;
  if (s == ".nan" || s == ".NaN") return NAN;
  char* end = nullptr;
  const double d = std::strtod(s.c_str(), &end);
  if (end == s.c_str() || *end != '\0') bad(v.line, std::string(what) + ": `" + s + "` is not a number");
  return d;
}

std::array<double, 3> triple(const Value& v, const char* what) {
  if (v.kind != Value::Seq || v.seq.size() != 3) bad(v.line, std::string(what) + ": expected a sequence of 3 numbers");
  return {number(v.seq[0], what), number(v.seq[1], what), number(v.seq[2], what)};
}

const Value* optional(const Value& map, const char* key) {
  const Value* v = map.get(key);
  return (v && v->kind != Value::Null) ? v : nullptr;
}

Color color_of(const Value& v, const char* what) {
  const auto t = triple(v, what);
  return rgb(t[0], t[1], t[2]);
}

Texture build_texture(const Value& v) {  // src/loader.rs:137-175
  if (v.kind == Value::Scalar) bad(v.line, "texture: `" + v.scalar + "` needs its colours");
  if (v.kind != Value::Map || v.map.size() != 1) bad(v.line, "texture: expected a mapping with a single key");
  const std::string& kind = v.map[0].first;
  const Value& arg = v.map[0].second;
  if (kind == "solid") return Texture::solid(color_of(arg, "texture.solid"));
  if (kind == "checker" || kind == "gradient" || kind == "ring" || kind == "striped") {
    if (arg.kind != Value::Seq || arg.seq.size() != 2) bad(arg.line ? arg.line : v.line, "texture." + kind + ": expected two colours");
    const Color a = color_of(arg.seq[0], "texture colour"), b = color_of(arg.seq[1], "texture colour");
    if (kind == "checker") return Texture::checker(a, b);
    if (kind == "gradient") return Texture::gradient(a, b);
    if (kind == "ring") return Texture::ring(a, b);
    return Texture::striped(a, b);
  }
  bad(v.line, "unknown texture `" + kind + "`");
}

Material build_material(const Value& v) {  // src/loader.rs:111-135
  if (v.kind != Value::Map) bad(v.line, "material: expected a mapping");
  Material m = Material::default_material();
  if (const Value* t = optional(v, "texture")) m.texture = build_texture(*t);
  if (const Value* x = optional(v, "ambient")) m.ambient = number(*x, "ambient");
  if (const Value* x = optional(v, "diffuse")) m.diffuse = number(*x, "diffuse");
  if (const Value* x = optional(v, "specular")) m.specular = number(*x, "specular");
  if (const Value* x = optional(v, "shininess")) m.shininess = number(*x, "shininess");
  if (const Value* x = optional(v, "transparency")) m.transparency = number(*x, "transparency");
  if (const Value* x = optional(v, "reflectivity")) m.reflectivity = number(*x, "reflectivity");
  if (const Value* x = optional(v, "refractivity")) m.refractivity = number(*x, "refractivity");
  return m;
}

SceneNode build_object(const Value& v) {  // src/loader.rs:59-97
  if (v.kind != Value::Map) bad(v.line, "object: expected a mapping");
  const Value* kind = optional(v, "kind");
  if (!kind || kind->kind != Value::Scalar) bad(v.line, "object: missing field `kind`");
  Matrix4x4 transform = Matrix4x4::identity();
  if (const Value* p = optional(v, "position")) {
    const auto t = triple(*p, "position");
    transform = transform * Matrix4x4::translate(t[0], t[1], t[2]);
  }
  if (const Value* r = optional(v, "rotation")) {
    const auto t = triple(*r, "rotation");
    transform = transform * Matrix4x4::rotate_x(t[0]);
    transform = transform * Matrix4x4::rotate_y(t[1]);
    transform = transform * Matrix4x4::rotate_z(t[2]);
  }
  if (const Value* s = optional(v, "scale")) {
    const auto t = triple(*s, "scale");
    transform = transform * Matrix4x4::scale(t[0], t[1], t[2]);
  }
  Material material = Material::default_material();
  if (const Value* m = optional(v, "material")) material = build_material(*m);
  SceneNode node;
  if (kind->scalar == "sphere")
    node = Sphere();
  else if (kind->scalar == "plane")
    node = Plane(vec3(0.0, 1.0, 0.0));
  else
    bad(kind->line, "unknown variant `" + kind->scalar + "`, expected `sphere` or `plane`");
  return node.with_material(material).with_transform(transform);
}

PointLight build_light(const Value& v) {  // src/loader.rs:34-41: the position is a vec3, w = 0
  if (v.kind != Value::Map) bad(v.line, "light: expected a mapping");
  const Value* p = optional(v, "position");
  if (!p) bad(v.line, "light: missing field `position`");
  const auto pos = triple(*p, "light.position");
  Color c = WHITE;
  if (const Value* col = optional(v, "color")) c = color_of(*col, "light.color");
  return PointLight{vec3(pos[0], pos[1], pos[2]), c};
}

}  // namespace

PackedScene PackedScene::from_yaml(const std::string& text) {
  const Value doc = Parser(split_lines(text)).document();
  if (doc.kind != Value::Map) throw Error(RTK_ERR_INVALID, "scene file: expected a mapping with `lights` and `objects`");
  const Value* lights = doc.get("lights");
  const Value* objects = doc.get("objects");
  if (!lights) throw Error(RTK_ERR_INVALID, "scene file: missing field `lights`");
  if (!objects) throw Error(RTK_ERR_INVALID, "scene file: missing field `objects`");
  PackedScene packed;
  auto each = [](const Value& list, const char* what, auto&& fn) {
    if (list.kind == Value::Null) return;  // `lights:` with nothing under it
    if (list.kind != Value::Seq) bad(list.line, std::string(what) + ": expected a sequence");
    for (const Value& item : list.seq) fn(item);
  };
  each(*lights, "lights", [&](const Value& v) { packed.lights_.push_back(build_light(v)); });
  each(*objects, "objects", [&](const Value& v) { packed.objects_.push_back(build_object(v)); });
  return packed;
}

PackedScene PackedScene::from_yaml_file(const std::string& path) {
  std::ifstream f(path, std::ios::binary);
  if (!f) throw Error(RTK_ERR_INVALID, "cannot open scene file " + path);
  std::ostringstream text;
  text << f.rdbuf();
  return from_yaml(text.str());
}

Scene PackedScene::build() const {  // src/loader.rs:187-199
  Scene scene;
  for (const PointLight& l : lights_) scene.add_light(l);
  for (const SceneNode& n : objects_) scene.add_object(n);
  return scene;
}

}  // namespace rtkhost
