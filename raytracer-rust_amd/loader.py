"""Scene files: the YAML schema of the reference's loader (src/loader.rs:22-200).

    lights:  [{position: [x,y,z], color: [r,g,b]?}]
    objects: [{kind: sphere|plane, position?, rotation?, scale?, material?}]
    material: {texture?, ambient?, diffuse?, specular?, shininess?, transparency?,
               reflectivity?, refractivity?}
    texture:  {solid: [r,g,b]} | {checker|gradient|ring|striped: [[r,g,b],[r,g,b]]}

Defaults, transform order (translate * rotate_x * rotate_y * rotate_z * scale) and the
w = 0 light position follow src/loader.rs:35-40, 60-96, 111-135.
"""
import yaml

from . import host

_TEXTURES = {
    "checker": host.Texture.checker,
    "gradient": host.Texture.gradient,
    "ring": host.Texture.ring,
    "striped": host.Texture.striped,
}


def _triple(v, what):
    if not isinstance(v, (list, tuple)) or len(v) != 3:
        raise ValueError(f"{what}: expected a list of 3 numbers, got {v!r}")
    return [float(x) for x in v]


def _build_texture(packed):
    if packed is None:
        return host.Texture.solid(host.WHITE)  # loader.rs:115
    if not isinstance(packed, dict) or len(packed) != 1:
        raise ValueError(f"texture: expected a single-key mapping, got {packed!r}")
    (kind, value), = packed.items()
    if kind == "solid":
        return host.Texture.solid(_triple(value, "texture.solid"))
    if kind in _TEXTURES:
        if not isinstance(value, (list, tuple)) or len(value) != 2:
            raise ValueError(f"texture.{kind}: expected two colours")
        return _TEXTURES[kind](_triple(value[0], kind), _triple(value[1], kind))
    raise ValueError(f"unknown texture kind {kind!r}")


def _build_material(packed):
    if packed is None:
        return host.Material.default()
    d = host.Material.default()
    # fields the reference's structs do not name are ignored, as serde does (src/loader.rs:99-109)
    get = lambda k, dv: float(packed[k]) if packed.get(k) is not None else dv
    return host.Material(
        texture=_build_texture(packed.get("texture")),
        ambient=get("ambient", d.ambient),
        diffuse=get("diffuse", d.diffuse),
        specular=get("specular", d.specular),
        shininess=get("shininess", d.shininess),
        transparency=get("transparency", d.transparency),
        reflectivity=get("reflectivity", d.reflectivity),
        refractivity=get("refractivity", d.refractivity),
    )


def _build_object(packed):
    kind = packed.get("kind")
    transform = host.Matrix4x4.identity()
    if packed.get("position") is not None:
        transform = transform * host.Matrix4x4.translate(*_triple(packed["position"], "position"))
    if packed.get("rotation") is not None:
        x, y, z = _triple(packed["rotation"], "rotation")
        transform = transform * host.Matrix4x4.rotate_x(x)
        transform = transform * host.Matrix4x4.rotate_y(y)
        transform = transform * host.Matrix4x4.rotate_z(z)
    if packed.get("scale") is not None:
        transform = transform * host.Matrix4x4.scale(*_triple(packed["scale"], "scale"))
    material = _build_material(packed.get("material"))
    if kind == "sphere":
        node = host.Sphere()
    elif kind == "plane":
        node = host.Plane(host.vec3(0.0, 1.0, 0.0))
    else:
        raise ValueError(f"unknown object kind {kind!r}")
    return node.with_material(material).with_transform(transform)


class PackedScene:
    """`PackedScene` (src/loader.rs:22-26, 177-200)."""

    def __init__(self, lights, objects):
        self.lights, self.objects = lights, objects

    @staticmethod
    def from_yaml_file(path):
        with open(path, "r") as f:
            return PackedScene.from_yaml(f.read())

    @staticmethod
    def from_yaml(text):
        doc = yaml.safe_load(text)
        if not isinstance(doc, dict) or "lights" not in doc or "objects" not in doc:
            raise ValueError("scene file needs `lights` and `objects`")
        return PackedScene(list(doc["lights"] or []), list(doc["objects"] or []))

    def build(self):
        scene = host.Scene()
        for light in self.lights:
            color = light.get("color")
            color = _triple(color, "light.color") if color is not None else [1.0, 1.0, 1.0]
            scene.add_light(host.PointLight(host.vec3(*_triple(light["position"], "light.position")), color))
        for obj in self.objects:
            scene.add_object(_build_object(obj))
        return scene
