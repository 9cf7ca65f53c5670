"""Deterministic synthetic scenes for the throughput and parity configurations.

The reference ships one scene (assets/scenes/test01.yaml).  The larger configurations
are generated here from a fixed SplitMix64 stream so that every run, rank and language
sees the same objects.  Scenes are built from the same primitives the reference's
loader uses (src/loader.rs:60-96): `translate * scale` transforms, `Material::default()`
values (src/scene/materials.rs:34-48) and lights built with `vec3` (w = 0,
src/loader.rs:35-40).
"""
import math

from . import host

MASK64 = (1 << 64) - 1


class SplitMix64:
    def __init__(self, seed):
        self.s = seed & MASK64

    def next_u64(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & MASK64
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
        return z ^ (z >> 31)

    def uniform(self):
        return (self.next_u64() >> 11) * (1.0 / 9007199254740992.0)  # 2**-53


# name -> (spheres, plane?, lights, width, height, seed, glass variant)
CONFIGS = {
    "C2": (64, True, 2, 1920, 1080, 0xC0FFEE02, False),
    "C2g": (64, True, 2, 1920, 1080, 0xC0FFEE02, True),
    "C3": (1000, True, 4, 3840, 2160, 0xC0FFEE03, False),
    "C4": (10000, True, 4, 3840, 2160, 0xC0FFEE04, False),
    "C5": (10000, False, 8, 16384, 16384, 0xC0FFEE05, False),
}

PITCH = 1.5


def grid_side(n_spheres):
    return int(math.ceil(math.sqrt(n_spheres)))


def make_scene(n_spheres, with_plane, n_lights, seed, glass=False):
    """Grid of jittered spheres resting on the y = 0 plane, lit from above."""
    rng = SplitMix64(seed)
    g = grid_side(n_spheres)
    extent = g * PITCH
    scene = host.Scene()
    if with_plane:
        floor = host.Material.default().with_pattern(host.Texture.checker(host.WHITE, host.BLACK)).with_specular(0.0)
        if glass:
            floor = floor.with_reflective(0.7)
        scene.add_object(host.Plane().with_material(floor))
    for i in range(n_spheres):
        u_r, u_jx, u_jz = rng.uniform(), rng.uniform(), rng.uniform()
        u_cr, u_cg, u_cb = rng.uniform(), rng.uniform(), rng.uniform()
        r = 0.25 + 0.35 * u_r
        cx = (i % g - (g - 1) / 2.0) * PITCH + (u_jx - 0.5) * 0.3
        cy = r
        cz = (i // g) * PITCH + (u_jz - 0.5) * 0.3
        material = host.Material.default().with_color(host.rgb(0.2 + 0.8 * u_cr, 0.2 + 0.8 * u_cg, 0.2 + 0.8 * u_cb))
        if glass and i % 4 == 0:
            material = (material.with_ambient(0.2).with_diffuse(0.2).with_specular(0.3)
                        .with_transparency(0.9).with_reflective(0.9))
        transform = host.Matrix4x4.translate(cx, cy, cz) * host.Matrix4x4.scale(r, r, r)
        scene.add_object(host.Sphere().with_material(material).with_transform(transform))
    for k in range(n_lights):
        pos = host.vec3((k - (n_lights - 1) / 2.0) * 0.4 * extent, 0.6 * extent + 10.0, -0.2 * extent - 5.0 * (k % 2))
        scene.add_light(host.PointLight(pos, host.rgb(1.0 / n_lights, 1.0 / n_lights, 1.0 / n_lights)))
    return scene


def make_camera(n_spheres, width, height):
    extent = grid_side(n_spheres) * PITCH
    camera = host.Camera(width, height, host.PI / 3.0)
    camera.transform = host.Matrix4x4.look_at(
        host.point(0.0, 0.35 * extent, -0.25 * extent), host.point(0.0, 0.0, 0.45 * extent), host.vec3(0.0, 1.0, 0.0)
    )
    return camera


def config(name, width=None, height=None):
    """(scene, camera) of a named configuration, optionally at a reduced resolution."""
    n, plane, lights, w, h, seed, glass = CONFIGS[name]
    scene = make_scene(n, plane, lights, seed, glass)
    camera = make_camera(n, width or w, height or h)
    return scene, camera
