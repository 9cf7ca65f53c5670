"""Host-side mirror of the reference's object model for the ray-trace path.

Same names and argument meaning as the reference so that tests read like the
reference's own: `point`/`vec3`/`rgb` (src/maths/vectors.rs:12-24, colors.rs:8-10),
`Matrix4x4` (src/maths/matrices.rs, transforms.rs), `Material` and its `with_*`
builders (src/scene/materials.rs:23-95), `PointLight` (src/scene/lighting.rs:8-21),
`Sphere`/`Plane` scene nodes (src/scene.rs:36-68, shapes/*.rs), `Scene`
(src/scene.rs:102-133) and `Camera` (src/scene/cameras.rs:7-49).

Everything here is plain f64 Python; it only builds the flat structs of include/rtk.h.
Rendering itself happens in the device library (device.py); nothing here computes a
pixel.
"""
import ctypes as C
import math

from . import abi

PI = math.pi


def point(x, y, z):
    """(x, y, z) with w = 1 (src/maths/vectors.rs:12-14)."""
    return (float(x), float(y), float(z), 1.0)


def vec3(x, y, z):
    """(x, y, z) with w = 0 (src/maths/vectors.rs:17-19)."""
    return (float(x), float(y), float(z), 0.0)


def vec4(x, y, z, w):
    return (float(x), float(y), float(z), float(w))


def rgb(r, g, b):
    return (float(r), float(g), float(b))


BLACK = rgb(0, 0, 0)
WHITE = rgb(1, 1, 1)
RED = rgb(1, 0, 0)


def _sub(a, b):
    return tuple(x - y for x, y in zip(a, b))


def _magnitude(v):
    # the magnitude includes w (src/maths/vectors.rs:52-59)
    return math.sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3])


def _normalize(v):
    m = _magnitude(v)
    return tuple(c / m for c in v)


def _cross(a, b):
    return vec3(a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0])


class Matrix4x4:
    """Row-major 4x4 of f64, element (row, column) at [column + 4*row] (matrices.rs:69-71)."""

    __slots__ = ("e",)

    def __init__(self, elements=None):
        self.e = [0.0] * 16 if elements is None else [float(x) for x in elements]
        if len(self.e) != 16:
            raise ValueError("Matrix4x4 needs 16 elements")

    @staticmethod
    def identity():
        return Matrix4x4([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1])

    @staticmethod
    def translate(x, y, z):  # transforms.rs:7-14
        return Matrix4x4([1, 0, 0, x, 0, 1, 0, y, 0, 0, 1, z, 0, 0, 0, 1])

    @staticmethod
    def scale(x, y, z):  # transforms.rs:17-24
        return Matrix4x4([x, 0, 0, 0, 0, y, 0, 0, 0, 0, z, 0, 0, 0, 0, 1])

    @staticmethod
    def rotate_x(r):  # transforms.rs:27-34
        c, s = math.cos(r), math.sin(r)
        return Matrix4x4([1, 0, 0, 0, 0, c, -s, 0, 0, s, c, 0, 0, 0, 0, 1])

    @staticmethod
    def rotate_y(r):  # transforms.rs:37-44
        c, s = math.cos(r), math.sin(r)
        return Matrix4x4([c, 0, s, 0, 0, 1, 0, 0, -s, 0, c, 0, 0, 0, 0, 1])

    @staticmethod
    def rotate_z(r):  # transforms.rs:47-54
        c, s = math.cos(r), math.sin(r)
        return Matrix4x4([c, -s, 0, 0, s, c, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1])

    @staticmethod
    def shear(x1, x2, y1, y2, z1, z2):  # transforms.rs:57-64
        return Matrix4x4([1, x1, x2, 0, y1, 1, y2, 0, z1, z2, 1, 0, 0, 0, 0, 1])

    @staticmethod
    def look_at(frm, to, up):  # transforms.rs:67-80
        forward = _normalize(_sub(to, frm))
        left = _cross(forward, _normalize(up))
        true_up = _cross(left, forward)
        orientation = Matrix4x4([
            left[0], left[1], left[2], 0,
            true_up[0], true_up[1], true_up[2], 0,
            -forward[0], -forward[1], -forward[2], 0,
            0, 0, 0, 1,
        ])
        return orientation * Matrix4x4.translate(-frm[0], -frm[1], -frm[2])

    def __getitem__(self, rc):
        row, col = rc
        return self.e[col + row * 4]

    def __mul__(self, rhs):
        if isinstance(rhs, Matrix4x4):  # matrices.rs:93-114, same summation order
            out = [0.0] * 16
            for row in range(4):
                for col in range(4):
                    s = 0.0
                    for i in range(4):
                        s += self.e[i + row * 4] * rhs.e[col + i * 4]
                    out[col + row * 4] = s
            return Matrix4x4(out)
        v = rhs  # vectors.rs:205-223
        return tuple(
            self.e[0 + r * 4] * v[0] + self.e[1 + r * 4] * v[1] + self.e[2 + r * 4] * v[2] + self.e[3 + r * 4] * v[3]
            for r in range(4)
        )

    def transpose(self):
        return Matrix4x4([self.e[r + c * 4] for r in range(4) for c in range(4)])

    def __eq__(self, other):  # approximate, like matrices.rs:81-91
        return all(abs(a - b) < 1e-5 for a, b in zip(self.e, other.e))

    def __repr__(self):
        rows = [" ".join(f"{self.e[c + r * 4]: .5f}" for c in range(4)) for r in range(4)]
        return "Matrix4x4(\n  " + "\n  ".join(rows) + ")"


class Texture:
    """`Texture::{Solid, Pattern}` with the four patterns a scene file can name
    (src/scene/materials.rs:7-10, src/loader.rs:137-175)."""

    __slots__ = ("kind", "a", "b", "transform", "inner")

    def __init__(self, kind, a, b=BLACK, transform=None, inner=()):
        self.kind, self.a, self.b = kind, rgb(*a), rgb(*b)
        self.transform = transform  # None, or the Matrix4x4 of the outermost `TransformPattern`
        self.inner = tuple(inner)   # the transforms of the TransformPatterns inside it, outer to inner

    def with_transform(self, transform):
        """`TransformPattern::new(pattern).with_transform(t)` (src/graphics/patterns.rs:12-25):
        repeated calls multiply on the right.  Acts on the outermost wrapper."""
        current = self.transform if self.transform is not None else Matrix4x4.identity()
        return Texture(self.kind, self.a, self.b, current * transform, self.inner)

    def wrap(self):
        """`TransformPattern::new(self)`: one more wrapper around what is there, with the identity
        transform (the type is generic over its pattern, src/graphics/patterns.rs:7-35, so wrappers nest)."""
        if self.transform is None:
            return Texture(self.kind, self.a, self.b, Matrix4x4.identity(), self.inner)
        if 1 + len(self.inner) >= abi.RTK_MAX_PATTERN_LEVELS:
            raise ValueError(f"at most {abi.RTK_MAX_PATTERN_LEVELS} nested TransformPatterns")
        return Texture(self.kind, self.a, self.b, Matrix4x4.identity(), (self.transform,) + self.inner)

    @staticmethod
    def solid(color):
        return Texture(abi.RTK_TEX_SOLID, color)

    @staticmethod
    def checker(a, b):
        return Texture(abi.RTK_TEX_CHECKER, a, b)

    @staticmethod
    def gradient(a, b):
        return Texture(abi.RTK_TEX_GRADIENT, a, b)

    @staticmethod
    def ring(a, b):
        return Texture(abi.RTK_TEX_RING, a, b)

    @staticmethod
    def striped(a, b):
        return Texture(abi.RTK_TEX_STRIPED, a, b)


class Material:
    """`Material` with the reference's defaults (src/scene/materials.rs:34-48)."""

    FIELDS = ("texture", "ambient", "diffuse", "specular", "shininess", "transparency", "reflectivity", "refractivity")

    def __init__(self, texture=None, ambient=0.1, diffuse=0.9, specular=0.9, shininess=200.0, transparency=0.0,
                 reflectivity=0.0, refractivity=1.0):
        self.texture = texture if texture is not None else Texture.solid(WHITE)
        self.ambient, self.diffuse, self.specular, self.shininess = ambient, diffuse, specular, shininess
        self.transparency, self.reflectivity, self.refractivity = transparency, reflectivity, refractivity

    @staticmethod
    def default():
        return Material()

    def _with(self, **kw):
        d = {k: getattr(self, k) for k in self.FIELDS}
        d.update(kw)
        return Material(**d)

    def with_color(self, color):
        return self._with(texture=Texture.solid(color))

    def with_pattern(self, texture):
        return self._with(texture=texture)

    def with_ambient(self, v):
        return self._with(ambient=v)

    def with_diffuse(self, v):
        return self._with(diffuse=v)

    def with_specular(self, v):
        return self._with(specular=v)

    def with_shininess(self, v):
        return self._with(shininess=v)

    def with_transparency(self, v):
        return self._with(transparency=v)

    def with_reflective(self, v):
        return self._with(reflectivity=v)

    def with_refractivity(self, v):
        return self._with(refractivity=v)

    def to_abi(self):
        m = abi.rtk_material()
        m.texture_kind = self.texture.kind
        m.color_a[:] = self.texture.a
        m.color_b[:] = self.texture.b
        if self.texture.transform is not None:
            m.pattern_transformed = 1 + len(self.texture.inner)
            m.pattern_transform[:] = self.texture.transform.e
            if self.texture.inner:
                flat = [v for t in self.texture.inner for v in t.e]
                arr = (C.c_double * len(flat))(*flat)
                m.inner_pattern_transforms = C.cast(arr, C.POINTER(C.c_double))
                m._keep = arr  # the struct only holds a pointer
        m.ambient, m.diffuse, m.specular, m.shininess = self.ambient, self.diffuse, self.specular, self.shininess
        m.transparency, m.reflectivity, m.refractivity = self.transparency, self.reflectivity, self.refractivity
        return m


class PointLight:
    """`PointLight::new(position, intensity)`; pass `vec3(..)` to get the w = 0 light a
    scene file produces (src/loader.rs:35-40) or `point(..)` for w = 1."""

    def __init__(self, position, intensity=WHITE):
        self.position = vec4(*position)
        self.intensity = rgb(*intensity)

    def to_abi(self):
        l = abi.rtk_light()
        l.position[:] = self.position
        l.intensity[:] = self.intensity
        return l


class SceneNode:
    """`SceneNode<S>`: shape + material + transform (src/scene.rs:36-68)."""

    def __init__(self, kind, plane_normal=(0.0, 0.0, 0.0, 0.0)):
        self.kind = kind
        self.plane_normal = vec4(*plane_normal)
        self.material = Material.default()
        self.transform = Matrix4x4.identity()

    def with_transform(self, transform):
        n = SceneNode(self.kind, self.plane_normal)
        n.material, n.transform = self.material, transform
        return n

    def with_material(self, material):
        n = SceneNode(self.kind, self.plane_normal)
        n.material, n.transform = material, self.transform
        return n

    def to_abi(self):
        o = abi.rtk_object()
        o.kind = self.kind
        o.transform[:] = self.transform.e
        o.plane_normal[:] = self.plane_normal
        m = self.material.to_abi()
        o.material = m
        o._keep = getattr(m, "_keep", None)  # inner pattern transforms the material points to
        return o


def Sphere():
    """`Sphere::new()` (src/scene/shapes/spheres.rs:10-15)."""
    return SceneNode(abi.RTK_SPHERE)


def Plane(normal=(0.0, 1.0, 0.0, 0.0)):
    """`Plane::new(normal)` (src/scene/shapes/planes.rs:14-19)."""
    return SceneNode(abi.RTK_PLANE, normal)


class Scene:
    """`Scene` (src/scene.rs:102-133)."""

    MAX_DEPTH = abi.RTK_REFERENCE_MAX_DEPTH

    def __init__(self):
        self.ambient_color = BLACK
        self.nodes = []
        self.lights = []
        self.max_depth = Scene.MAX_DEPTH
        self._abi_cache = None

    def add_object(self, node):
        self.nodes.append(node)
        self._abi_cache = None

    def add_light(self, light):
        self.lights.append(light)
        self._abi_cache = None

    def to_abi(self):
        """Flat rtk_scene; the arrays stay alive as attributes of the returned struct."""
        objs = (abi.rtk_object * max(1, len(self.nodes)))()
        kept = []
        for i, n in enumerate(self.nodes):
            o = n.to_abi()
            objs[i] = o
            if o._keep is not None:
                kept.append(o._keep)
        lights = (abi.rtk_light * max(1, len(self.lights)))()
        for i, l in enumerate(self.lights):
            lights[i] = l.to_abi()
        s = abi.rtk_scene()
        s.objects = C.cast(objs, C.POINTER(abi.rtk_object))
        s.n_objects = len(self.nodes)
        s.lights = C.cast(lights, C.POINTER(abi.rtk_light))
        s.n_lights = len(self.lights)
        s.ambient_color[:] = self.ambient_color
        s.max_depth = self.max_depth
        s._keep = (objs, lights, kept)
        return s


class Camera:
    """`Camera::new(width, height, field_of_view)` (src/scene/cameras.rs:19-49);
    `transform` is public in the reference (cameras.rs:14) and assignable here."""

    def __init__(self, width, height, field_of_view):
        half_view = math.tan(field_of_view / 2.0)
        aspect = float(width) / float(height)
        if aspect >= 1.0:
            half_width = half_view
            half_height = half_width / aspect
        else:
            half_width = half_view * aspect
            half_height = half_view
        self.width, self.height = int(width), int(height)
        self.half_width, self.half_height = half_width, half_height
        self.field_of_view = float(field_of_view)
        self.pixel_size = (half_width * 2.0) / float(width)
        self.transform = Matrix4x4.look_at(point(0.0, 1.5, -5.0), point(0.0, 1.0, 0.0), vec3(0.0, 1.0, 0.0))

    def to_abi(self):
        c = abi.rtk_camera()
        c.width, c.height = self.width, self.height
        c.half_width, c.half_height = self.half_width, self.half_height
        c.field_of_view, c.pixel_size = self.field_of_view, self.pixel_size
        c.transform[:] = self.transform.e
        return c

    def supersampled(self, samples_per_axis):
        """The camera whose pixels are this camera's samples: the same view at samples_per_axis
        times the resolution (no counterpart in the reference, which samples pixel centres only)."""
        s = int(samples_per_axis)
        if s < 1 or s > abi.RTK_MAX_SAMPLES_PER_AXIS:
            raise ValueError(f"samples_per_axis must be 1..{abi.RTK_MAX_SAMPLES_PER_AXIS}")
        c = Camera.__new__(Camera)
        c.__dict__.update(self.__dict__)
        c.width, c.height = self.width * s, self.height * s
        c.pixel_size = self.pixel_size / float(s)
        return c

    def render(self, scene, device=None, samples_per_axis=1):
        """`Camera::render(&self, scene) -> Canvas` (cameras.rs:72-85) on the device;
        samples_per_axis > 1 averages that many squared samples per pixel."""
        from . import device as _device

        dev = device if device is not None else _device.default_device()
        return dev.render(scene, self, samples_per_axis)
