"""Parity of the HIP path against the CPU oracle and the reference's golden image.

Every call goes through the C ABI (ctypes -> librtk.so).  Bars (north star): linear FP32
colour within 1e-4 relative of the f64 result; 8-bit output within 1 LSB on >= 99.99 % of
pixels.  The path decides everything on f64 values, so in practice frames come out
bit-identical after the 8-bit cast; the tests also pin the ray counts, which must equal
the oracle's exactly (one ray = one `Scene::intersect` call, src/scene.rs:146, 217).
"""
import hashlib
import importlib
import json
import math
import os

import numpy as np
import pytest

from gpu_common import assert_parity

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev(rtk):
    device = importlib.import_module("raytracer-rust_amd.device")
    d = device.Device(0)
    yield d
    d.close()


def counts(st):
    return (st.rays_primary, st.rays_shadow, st.rays_secondary)


def check_against_oracle(dev, oracle, scene, camera, what, rect=None, expect_identical=True):
    canvas = dev.render(scene, camera)
    rgb, rgba, ost = oracle.render(scene, camera, rect=rect, threads=16)
    g_rgb, g_rgba = canvas.rgb, canvas.rgba8
    if rect:
        x0, y0, x1, y1 = rect
        g_rgb, g_rgba = g_rgb[y0:y1, x0:x1], g_rgba[y0:y1, x0:x1]
    else:
        assert counts(canvas.stats) == counts(ost), what
    max_rel, bad_rel, bad_lsb, not_equal = assert_parity(g_rgb, g_rgba, rgb, rgba, what)
    if expect_identical:
        assert not_equal == 0, f"{what}: {not_equal} pixels differ after the 8-bit cast"
        assert max_rel < 1e-6, f"{what}: max relative error {max_rel:.3e}"
    assert (g_rgba[..., 3] == 255).all()
    return canvas


# ---------------------------------------------------------------- C1: the bundled scene
def test_bundled_scene_matches_reference_image(rtk, dev, oracle, golden_dir):
    scene = rtk.PackedScene.from_yaml_file(os.path.join(golden_dir, "test01.yaml")).build()
    camera = rtk.Camera(1920, 1080, rtk.PI / 3.0)
    canvas = check_against_oracle(dev, oracle, scene, camera, "C1")
    meta = json.load(open(os.path.join(golden_dir, "output_meta.json")))
    golden = np.load(os.path.join(golden_dir, "output_stride4.npy"))
    d = np.abs(canvas.rgba8[::4, ::4, :3].astype(int) - golden.astype(int)).max(axis=2)
    assert (d > 1).sum() <= int(d.size * 1e-4)
    assert (d > 0).sum() == 0
    assert hashlib.sha256(canvas.rgba8.tobytes()).hexdigest() == meta["decoded_rgba_sha256"]
    assert canvas.stats.rays == 21939338


# ---------------------------------------------------------------- synthetic configurations
@pytest.mark.parametrize("name,w,h", [("C2", 1920, 1080), ("C2g", 960, 540), ("C3", 768, 432), ("C4", 256, 144),
                                      ("C5", 192, 192)])
def test_synthetic_config_reduced_resolution(rtk, dev, oracle, name, w, h):
    scene, camera = rtk.synthetic.config(name, w, h)
    check_against_oracle(dev, oracle, scene, camera, f"{name} {w}x{h}")


@pytest.mark.parametrize("name,rect", [("C3", (1792, 1180, 2048, 1324)), ("C4", (1800, 1000, 1992, 1108)),
                                       ("C4", (0, 2060, 160, 2160))])
def test_full_size_frame_crop(rtk, dev, oracle, name, rect):
    """The benchmark's full sizes (3840x2160): the whole frame is rendered on the device, a
    rectangle of it is checked against the oracle."""
    scene, camera = rtk.synthetic.config(name)
    check_against_oracle(dev, oracle, scene, camera, f"{name} crop {rect}", rect=rect)


def test_full_size_properties(rtk, dev):
    """Size-independent properties at full size: determinism, alpha, ray-count identities."""
    scene, camera = rtk.synthetic.config("C4")
    a = dev.render(scene, camera)
    b = dev.render(scene, camera)
    assert np.array_equal(a.rgb, b.rgb) and np.array_equal(a.rgba8, b.rgba8)
    assert counts(a.stats) == counts(b.stats)
    st = a.stats
    assert st.rays_primary == 3840 * 2160 and st.rays_secondary == 0
    n_lights = len(scene.lights)
    assert st.rays_primary <= st.rays_shadow + (a.rgba8[..., :3].max(axis=2) == 0).sum()  # every hit pixel casts >= 1 shadow ray
    assert st.rays_shadow <= n_lights * st.rays_primary
    # the cast is the reference's: truncation of colour * 255, saturated
    v = np.clip(np.trunc(a.rgb.astype(np.float64) * 255.0), 0, 255).astype(np.uint8)
    assert (np.abs(v.astype(int) - a.rgba8[..., :3].astype(int)) <= 1).all()


# ---------------------------------------------------------------- Scene::trace known answers
@pytest.fixture()
def test_scene(rtk):
    scene = rtk.Scene()
    scene.add_light(rtk.PointLight(rtk.vec3(-10, 10, -10), rtk.WHITE))
    scene.add_object(rtk.Sphere().with_material(
        rtk.Material.default().with_color(rtk.rgb(0.8, 1.0, 0.6)).with_diffuse(0.7).with_specular(0.2)))
    scene.add_object(rtk.Sphere().with_transform(rtk.Matrix4x4.scale(0.5, 0.5, 0.5)))
    return scene


def approx(a, b, eps=1e-5):
    return all(abs(x - y) < eps for x, y in zip(a, b))


def test_trace_known_answers(rtk, dev, test_scene):
    h = dev.upload(test_scene)
    # src/scene.rs:495-502
    c = dev.trace(h, [(0, 0, -5, 1)], [(0, 0, 1, 0)])[0]
    assert approx(c, (0.38012764, 0.47515953, 0.28509575))
    # src/scene.rs:483-492: a miss returns the ambient colour
    test_scene.ambient_color = rtk.RED
    h2 = dev.upload(test_scene)
    assert approx(dev.trace(h2, [(0, 0, -5, 1)], [(0, 1, 0, 0)])[0], rtk.RED)


def test_trace_inside_and_shadow(rtk, dev, oracle, test_scene):
    # src/scene.rs:466-480: light inside, ray from inside the inner sphere
    test_scene.lights[0] = rtk.PointLight(rtk.point(0, 0.25, 0), rtk.WHITE)
    h = dev.upload(test_scene)
    o, d = (0, 0, 0, 1), (0, 0, 1, 0)
    assert approx(dev.trace(h, [o], [d])[0], oracle.trace(test_scene, o, d), 1e-12)
    # src/scene.rs:537-552
    scene = rtk.Scene()
    scene.add_light(rtk.PointLight(rtk.point(0, 0, -10), rtk.WHITE))
    scene.add_object(rtk.Sphere())
    scene.add_object(rtk.Sphere().with_transform(rtk.Matrix4x4.translate(0, 0, 10)))
    h = dev.upload(scene)
    o, d = (0, 0, 5, 1), (0, 0, 1, 0)
    got = dev.trace(h, [o], [d])[0]
    assert approx(got, (0.1, 0.1, 0.1)) and approx(got, oracle.trace(scene, o, d), 1e-12)


def test_trace_reflection_known_answer(rtk, dev, oracle, test_scene):
    # src/scene.rs:571-592: the reflected part of this ray is rgb(0.19007981, 0.23759975, 0.14255986)
    test_scene.add_object(rtk.Plane(rtk.vec3(0, 1, 0)).with_material(rtk.Material.default().with_reflective(0.5))
                          .with_transform(rtk.Matrix4x4.translate(0, -1, 0)))
    h = dev.upload(test_scene)
    s2 = math.sqrt(2.0)
    o, d = (0, 0, -3, 1), (0, -s2 / 2, s2 / 2, 0)
    got = dev.trace(h, [o], [d])[0]
    want = oracle.trace(test_scene, o, d)
    assert approx(got, want, 1e-12)
    data = oracle.lighting_data(test_scene, o, d, [(2, s2)])
    assert approx(oracle.reflected_color(test_scene, data), (0.19007981, 0.23759975, 0.14255986))


def test_random_rays_through_glass(rtk, dev, oracle):
    """Overlapping glass spheres with distinct refractive indices: exercises the n1/n2 rule
    (src/scene/lighting.rs:81-116), total internal reflection and unnormalised secondary rays."""
    rng = np.random.default_rng(11)
    scene = rtk.Scene()
    scene.add_light(rtk.PointLight(rtk.vec3(-4, 6, -8), rtk.WHITE))
    scene.add_light(rtk.PointLight(rtk.point(5, 3, -2), rtk.rgb(0.3, 0.4, 0.5)))
    scene.add_object(rtk.Plane().with_material(
        rtk.Material.default().with_pattern(rtk.Texture.checker(rtk.WHITE, rtk.BLACK)).with_reflective(0.3))
        .with_transform(rtk.Matrix4x4.translate(0, -1.5, 0)))
    for i in range(12):
        c = rng.normal(size=3) * 0.9
        r = rng.uniform(0.5, 1.1)
        m = (rtk.Material.default().with_color(tuple(rng.uniform(0.2, 1, 3))).with_diffuse(0.3)
             .with_transparency(rng.choice([0.0, 0.6, 0.9])).with_reflective(rng.choice([0.0, 0.4, 0.9]))
             .with_refractivity(rng.choice([1.0, 1.3, 1.5, 2.4])))
        scene.add_object(rtk.Sphere().with_material(m).with_transform(
            rtk.Matrix4x4.translate(*c) * rtk.Matrix4x4.scale(r, r, r)))
    n = 4096
    o = np.concatenate([rng.normal(size=(n, 3)) * 2.0, np.ones((n, 1))], axis=1)  # many origins lie inside spheres
    d = rng.normal(size=(n, 3))
    d = np.concatenate([d / np.linalg.norm(d, axis=1, keepdims=True), np.zeros((n, 1))], axis=1)
    h = dev.upload(scene)
    got = dev.trace(h, o, d)
    want = np.array([oracle.trace(scene, tuple(o[i]), tuple(d[i])) for i in range(n)])
    err = np.abs(got - want) / np.maximum(np.abs(want), 1e-3)
    assert (err.max(axis=1) > 1e-9).sum() == 0, f"max {err.max():.3e}"


def test_refractive_index_by_the_letter_in_a_small_scene(rtk, dev, oracle):
    """The same letter-of-the-law n1 (src/scene/lighting.rs:81-116) on the path scenes of at most 8 objects
    take: every object intersected exactly, no filter.  One-pixel cameras whose single ray (the pixel
    centre of a 1x1 frame lies on the camera axis) grazes a small glass sphere within 1e-11 of tangency,
    inside a large glass sphere, every refractive index different."""
    rng = np.random.default_rng(17)
    M = rtk.Matrix4x4
    scene = rtk.Scene()
    scene.add_light(rtk.PointLight(rtk.vec3(-6, 9, -7), rtk.WHITE))

    def glass(refr, color):
        return (rtk.Material.default().with_color(color).with_diffuse(0.4).with_transparency(0.8).with_reflective(0.3)
                .with_refractivity(refr))

    scene.add_object(rtk.Sphere().with_material(glass(1.5, (0.9, 0.9, 1.0))).with_transform(M.scale(30, 30, 30)))
    small = []
    for i in range(6):
        c = rng.normal(size=3) * 6.0
        r = float(rng.uniform(0.6, 1.4))
        small.append((c, r))
        scene.add_object(rtk.Sphere().with_material(glass(1.1 + 0.17 * i, tuple(rng.uniform(0.2, 1, 3))))
                         .with_transform(M.translate(*c) * M.scale(r, r, r)))
    assert len(scene.nodes) <= 8
    grazing = different_n1 = 0
    for k in range(48):
        c, r = small[k % 6]
        d = rng.normal(size=3)
        d /= np.linalg.norm(d)
        perp = np.cross(d, rng.normal(size=3))
        perp /= np.linalg.norm(perp)
        foot = c + perp * r * (1.0 - float(rng.choice([2e-12, 5e-12, 1e-11])))
        o = foot - d * float(rng.uniform(-2.5e-6, 2.5e-6)) if k % 3 == 0 else foot + d * float(rng.uniform(0.5, 4.0))
        camera = rtk.Camera(1, 1, 0.3)
        up = np.cross(d, rng.normal(size=3))
        camera.transform = M.look_at(rtk.point(*o), rtk.point(*(o + d)), rtk.vec3(*up))
        canvas = dev.render(scene, camera)
        rgb, rgba, ost = oracle.render(scene, camera, threads=1)
        assert counts(canvas.stats) == counts(ost), k
        err = np.abs(canvas.rgb.astype(np.float64) - rgb) / np.maximum(np.abs(rgb), 1e-3)
        assert err.max() < 1e-6 and np.array_equal(canvas.rgba8, rgba), (k, float(err.max()))
        ro, rd = oracle.camera_ray_for_pixel(camera, 0, 0)
        by_obj = {}
        for obj, t in oracle.scene_intersect(scene, ro, rd):
            by_obj.setdefault(obj, []).append(t)
        grazing += any(len(ts) == 2 and abs(ts[0] - ts[1]) < 1e-5 for ts in by_obj.values())
    assert grazing >= 30, grazing  # the construction does what it says


# ---------------------------------------------------------------- shapes, textures, lights
def test_general_affine_objects_and_all_textures(rtk, dev, oracle):
    """Non-uniform scale, rotation and shear on spheres; rotated planes keep their untransformed
    normal (src/scene/shapes/planes.rs:30-32); every texture kind; w = 0 and w = 1 lights."""
    M = rtk.Matrix4x4
    scene = rtk.Scene()
    scene.add_light(rtk.PointLight(rtk.vec3(-10, 10, -10), rtk.rgb(0.7, 0.7, 0.7)))
    scene.add_light(rtk.PointLight(rtk.point(6, 8, -6), rtk.rgb(0.5, 0.4, 0.3)))
    mat = rtk.Material.default()
    scene.add_object(rtk.Plane().with_material(mat.with_pattern(rtk.Texture.ring(rtk.WHITE, rtk.rgb(0.1, 0.3, 0.5)))))
    scene.add_object(rtk.Plane().with_material(mat.with_pattern(rtk.Texture.striped(rtk.rgb(1, 0.5, 0), rtk.BLACK)))
                     .with_transform(M.translate(0, 0, 12) * M.rotate_x(math.pi / 2)))
    scene.add_object(rtk.Sphere().with_material(mat.with_pattern(rtk.Texture.gradient(rtk.RED, rtk.rgb(0, 0, 1))))
                     .with_transform(M.translate(-2, 1, 1) * M.rotate_z(0.6) * M.scale(1.6, 0.5, 1.0)))
    scene.add_object(rtk.Sphere().with_material(mat.with_pattern(rtk.Texture.checker(rtk.WHITE, rtk.rgb(0.2, 0.6, 0.2))))
                     .with_transform(M.translate(2, 1.2, 0.5) * M.rotate_y(0.4) * M.rotate_x(-0.7) * M.scale(0.6, 1.2, 0.9)))
    scene.add_object(rtk.Sphere().with_material(mat.with_color(rtk.rgb(0.9, 0.8, 0.2)).with_reflective(0.4))
                     .with_transform(M.translate(0, 0.8, -1.0) * M.shear(0.4, 0, 0, 0.3, 0, 0) * M.scale(0.8, 0.8, 0.8)))
    scene.add_object(rtk.Sphere().with_material(mat.with_color(rtk.rgb(0.3, 0.3, 0.9)))
                     .with_transform(M.scale(1, 0, 1)))  # singular: identity inverse (src/scene.rs:55-62)
    camera = rtk.Camera(640, 360, rtk.PI / 3.0)
    check_against_oracle(dev, oracle, scene, camera, "affine+textures")


def test_exotic_light_w_still_matches(rtk, dev, oracle):
    """A light whose w is neither 0 nor 1 disables the filter's cull for its rays (every sphere is
    evaluated in f64); results must still equal the oracle."""
    scene, camera = rtk.synthetic.config("C2", 240, 135)
    scene.lights[1] = rtk.PointLight(rtk.vec4(5, 30, -10, 2.5), rtk.rgb(0.5, 0.5, 0.5))
    scene.lights[0] = rtk.PointLight(rtk.vec4(-8, 25, -4, 0.4), rtk.rgb(0.5, 0.5, 0.5))
    check_against_oracle(dev, oracle, scene, camera, "exotic light w")


def test_camera_inside_a_sphere_and_depth_limits(rtk, dev, oracle, golden_dir):
    scene = rtk.PackedScene.from_yaml_file(os.path.join(golden_dir, "test01.yaml")).build()
    camera = rtk.Camera(320, 180, rtk.PI / 2.0)
    camera.transform = rtk.Matrix4x4.look_at(rtk.point(-0.5, 1.0, 0.4), rtk.point(1.5, 0.5, -0.5), rtk.vec3(0, 1, 0))
    for depth in (0, 1, 2, 5, 8):
        scene.max_depth = depth
        check_against_oracle(dev, oracle, scene, camera, f"inside sphere, max_depth={depth}")


@pytest.mark.parametrize("n_lights", [0, 1, 3, 5, 8, 11])
def test_light_counts(rtk, dev, oracle, n_lights):
    """Shadow sweeps fuse 1, 2, 4 or 8 lights; counts in between and beyond use padded / repeated batches."""
    scene = rtk.synthetic.make_scene(40, True, n_lights, 1234)
    camera = rtk.synthetic.make_camera(40, 320, 180)
    check_against_oracle(dev, oracle, scene, camera, f"{n_lights} lights")


# ---------------------------------------------------------------- frame shapes
@pytest.mark.parametrize("w,h", [(1, 1), (31, 7), (33, 9), (100, 37), (257, 3)])
def test_ragged_frames(rtk, dev, oracle, w, h):
    scene, _ = rtk.synthetic.config("C2", 8, 8)
    camera = rtk.synthetic.make_camera(64, w, h)
    check_against_oracle(dev, oracle, scene, camera, f"{w}x{h}")


def test_empty_scene_and_no_spheres(rtk, dev, oracle):
    scene = rtk.Scene()
    scene.ambient_color = rtk.rgb(0.25, 0.5, 0.75)
    camera = rtk.Camera(64, 40, 1.0)
    c = check_against_oracle(dev, oracle, scene, camera, "empty scene")
    assert np.allclose(c.rgb, (0.25, 0.5, 0.75)) and counts(c.stats) == (64 * 40, 0, 0)
    scene.add_light(rtk.PointLight(rtk.vec3(0, 5, 0)))
    scene.add_object(rtk.Plane())
    check_against_oracle(dev, oracle, scene, camera, "plane only")


# ---------------------------------------------------------------- cluster / bundle culling under stress
def _random_field(rtk, rng, n, affine, r_lo, r_hi, extent, lights, planes=1):
    """n opaque spheres scattered in a box (overlaps allowed), optionally general affine."""
    M = rtk.Matrix4x4
    scene = rtk.Scene()
    for pos in lights:
        scene.add_light(rtk.PointLight(pos, rtk.rgb(*(1.0 / len(lights),) * 3)))
    mat = rtk.Material.default()
    if planes >= 1:
        scene.add_object(rtk.Plane().with_material(mat.with_pattern(rtk.Texture.checker(rtk.WHITE, rtk.BLACK)).with_specular(0.0)))
    if planes >= 2:
        scene.add_object(rtk.Plane().with_material(mat.with_color(rtk.rgb(0.4, 0.5, 0.6)))
                         .with_transform(M.translate(0, 0, extent * 1.2) * M.rotate_x(math.pi / 2 - 0.2)))
    for _ in range(n):
        r = float(np.exp(rng.uniform(np.log(r_lo), np.log(r_hi))))
        c = (rng.uniform(-extent, extent), rng.uniform(0.0, 0.4 * extent), rng.uniform(0.0, 2 * extent))
        t = M.translate(*c)
        if affine:
            t = t * M.rotate_y(rng.uniform(0, 6.28)) * M.rotate_x(rng.uniform(0, 6.28)) * \
                M.scale(r * rng.uniform(0.3, 1.0), r * rng.uniform(0.3, 1.0), r)
        else:
            t = t * M.scale(r, r, r)
        col = rtk.rgb(*rng.uniform(0.2, 1.0, size=3))
        scene.add_object(rtk.Sphere().with_material(mat.with_color(col)).with_transform(t))
    return scene


def _field_camera(rtk, extent, w, h, eye=None, to=None):
    camera = rtk.Camera(w, h, rtk.PI / 3)
    camera.transform = rtk.Matrix4x4.look_at(eye or rtk.point(0, 0.5 * extent, -0.6 * extent),
                                             to or rtk.point(0, 0, 0.9 * extent), rtk.vec3(0, 1, 0))
    return camera


@pytest.mark.parametrize("glass", [False, True])
def test_shininess_of_every_kind(rtk, dev, oracle, glass):
    """`reflect_dot_eye.powf(shininess)` (src/scene/lighting.rs:143-147): integer exponents from 1 to 4095 go through
    the kernels' squaring-and-multiplying power, everything else (fractions, 0, 4096 and more) through the
    library's pow.  Black and coloured spheres (a black channel keeps even the faintest highlight alive), every
    exponent kind in one scene, opaque (fused kernel) and with mirrors and glass among them (the ray-tree kernels; a few glass pixels
    there exceed the FP32 range on both sides, which the comparison counts as equal)."""
    rng = np.random.default_rng(77 if glass else 76)
    M = rtk.Matrix4x4
    E = 12.0
    scene = rtk.Scene()
    for pos in (rtk.vec3(-0.8 * E, 1.5 * E, -0.5 * E), rtk.vec3(0.9 * E, 1.1 * E, 0.3 * E)):
        scene.add_light(rtk.PointLight(pos, rtk.rgb(0.6, 0.5, 0.7)))
    mat = rtk.Material.default()
    scene.add_object(rtk.Plane().with_material(mat.with_color(rtk.rgb(0.0, 0.3, 0.0)).with_shininess(33.3)))
    kinds = [1.0, 2.0, 3.0, 7.0, 10.0, 200.0, 255.0, 256.0, 1000.0, 4095.0, 4096.0, 10000.0, 0.0, 0.5, 33.3, 199.999]
    for i in range(170):
        r = float(rng.uniform(0.4, 1.6))
        c = (rng.uniform(-E, E), rng.uniform(0.3, 0.4 * E), rng.uniform(0.0, 2 * E))
        col = rtk.rgb(0.0, 0.0, 0.0) if i % 3 == 0 else rtk.rgb(*rng.uniform(0.0, 1.0, size=3))
        m = mat.with_color(col).with_shininess(kinds[i % len(kinds)]).with_specular(float(rng.choice([0.9, 0.3])))
        if glass and i % 5 == 0:
            m = m.with_transparency(0.8).with_reflective(0.6).with_refractivity(1.5) if i % 10 == 0 else m.with_reflective(0.6)
        scene.add_object(rtk.Sphere().with_material(m).with_transform(M.translate(*c) * M.scale(r, r, r)))
    check_against_oracle(dev, oracle, scene, _field_camera(rtk, E, 256, 144), f"shininess kinds, glass={glass}")


@pytest.mark.parametrize("seed,affine", [(1, False), (2, True), (3, True)])
def test_clustered_fields_w0_lights(rtk, dev, oracle, seed, affine):
    """Several hundred overlapping spheres of very different sizes (many clusters, large r_max
    against small spheres), general affine in two of the cases, two planes, w = 0 lights."""
    rng = np.random.default_rng(seed)
    E = 20.0
    lights = [rtk.vec3(-0.8 * E, 1.5 * E, -0.5 * E), rtk.vec3(0.9 * E, 1.1 * E, 0.3 * E), rtk.vec3(0.1 * E, 2.0 * E, 2.5 * E)]
    scene = _random_field(rtk, rng, 700, affine, 0.05, 4.0, E, lights, planes=2)
    check_against_oracle(dev, oracle, scene, _field_camera(rtk, E, 192, 108), f"field seed {seed}")


def test_lights_near_the_origin_and_inside_the_field(rtk, dev, oracle):
    """w = 0 lights close to the world origin make rho = -1/|L| large: the reference's shadow
    rays then bend strongly (the target ball grows quickly along the ray)."""
    rng = np.random.default_rng(11)
    E = 12.0
    lights = [rtk.vec3(0.3, 0.8, 0.2), rtk.vec3(-2.0, 3.0, 6.0), rtk.vec3(0.0, 0.05, 0.0), rtk.vec3(4.0, 1.0, 14.0)]
    scene = _random_field(rtk, rng, 400, False, 0.1, 1.5, E, lights)
    check_against_oracle(dev, oracle, scene, _field_camera(rtk, E, 160, 90), "near lights")


@pytest.mark.parametrize("case", ["close lights", "far field", "tiny spheres", "mirrored and squashed", "inside"])
def test_spheres_behind_their_shadow_rays(rtk, dev, oracle, case):
    """The fused kernel leaves out the exact test of a sphere an FP32 test proves to lie behind the
    shadow ray (mostly the sphere the pixel sits on).  Scenes at the edges of that proof: w = 0 lights a
    radius away from spheres and inside one; a field 20,000 units from the origin (FP32 cannot tell a
    point from its surface there); spheres a thousandth of the field in radius; spheres with a negative
    uniform scale (one radius all the same) among squashed and rotated ones (never admitted); the
    camera inside a large sphere (shading points stand INSIDE the surface)."""
    M = rtk.Matrix4x4
    rng = np.random.default_rng({"close lights": 1, "far field": 2, "tiny spheres": 3, "mirrored and squashed": 4, "inside": 5}[case])
    E = 15.0
    off = np.array([2.0e4, -1.0e4, 3.0e4]) if case == "far field" else np.zeros(3)
    lights = [rtk.vec3(-0.8 * E, 1.5 * E, -0.5 * E), rtk.vec3(0.9 * E, 1.1 * E, 0.3 * E), rtk.vec3(0.1 * E, 2.0 * E, 2.5 * E)]
    if case == "close lights":
        lights = [rtk.vec3(0.0, 2.1, 0.9 * E), rtk.vec3(3.0, 1.0, 0.5 * E), rtk.vec3(-4.0, 0.6, 1.2 * E), rtk.vec3(0.5, 0.5, 0.6 * E)]
    if case == "far field":
        lights = [rtk.vec3(*(off + np.array([-0.8 * E, 1.5 * E, -0.5 * E]))), rtk.vec3(*(off + np.array([0.9 * E, 1.1 * E, 0.3 * E])))]
    scene = rtk.Scene()
    for pos in lights:
        scene.add_light(rtk.PointLight(pos, rtk.rgb(*(1.0 / len(lights),) * 3)))
    mat = rtk.Material.default()
    scene.add_object(rtk.Plane().with_material(mat.with_pattern(rtk.Texture.checker(rtk.WHITE, rtk.BLACK)).with_specular(0.0))
                     .with_transform(M.translate(*off)))
    r_lo, r_hi = (0.01, 0.03) if case == "tiny spheres" else (0.2, 2.5)
    for i in range(500):
        r = float(np.exp(rng.uniform(np.log(r_lo), np.log(r_hi))))
        c = off + np.array([rng.uniform(-E, E), rng.uniform(0.0, 0.4 * E), rng.uniform(0.0, 2 * E)])
        t = M.translate(*c)
        kind = i % 4 if case == "mirrored and squashed" else 0
        if kind == 1:
            t = t * M.scale(-r, -r, -r)
        elif kind == 2:
            t = t * M.scale(r, 0.5 * r, r)
        elif kind == 3:
            t = t * M.rotate_y(rng.uniform(0, 6.28)) * M.scale(r, r, r)
        else:
            t = t * M.scale(r, r, r)
        scene.add_object(rtk.Sphere().with_material(mat.with_color(rtk.rgb(*rng.uniform(0.2, 1.0, size=3)))).with_transform(t))
    if case == "close lights":  # a light inside this one
        scene.add_object(rtk.Sphere().with_material(mat).with_transform(M.translate(0.5, 0.5, 0.6 * E) * M.scale(1.5, 1.5, 1.5)))
    eye, to = off + np.array([0, 0.5 * E, -0.6 * E]), off + np.array([0, 0, 0.9 * E])
    if case == "inside":
        scene.add_object(rtk.Sphere().with_material(mat.with_color(rtk.rgb(0.8, 0.7, 0.6))).with_transform(M.translate(0, 0, 0) * M.scale(60, 60, 60)))
    camera = _field_camera(rtk, E, 160, 90, eye=rtk.point(*eye), to=rtk.point(*to))
    check_against_oracle(dev, oracle, scene, camera, f"behind test: {case}")


def test_clustered_field_w1_and_mixed_lights(rtk, dev, oracle):
    """Lights built from points (w = 1) and a mix of w = 0 / w = 1 take the two-kernel path."""
    rng = np.random.default_rng(5)
    E = 15.0
    scene = _random_field(rtk, rng, 300, True, 0.1, 2.0, E, [rtk.point(-10, 20, -8), rtk.point(12, 9, 20)])
    check_against_oracle(dev, oracle, scene, _field_camera(rtk, E, 160, 90), "w=1 lights")
    scene.lights[0] = rtk.PointLight(rtk.vec3(-10, 20, -8), rtk.rgb(0.5, 0.5, 0.5))
    check_against_oracle(dev, oracle, scene, _field_camera(rtk, E, 160, 90), "mixed lights")


def test_camera_inside_the_field_looking_everywhere(rtk, dev, oracle):
    """The camera sits between the spheres (some patches' cones contain whole clusters, others
    look away from all of them) and inside one of them."""
    rng = np.random.default_rng(8)
    E = 10.0
    scene = _random_field(rtk, rng, 500, False, 0.2, 1.2, E, [rtk.vec3(-6, 18, -4), rtk.vec3(7, 15, 25)])
    for eye, to in ((rtk.point(0.3, 2.0, 9.0), rtk.point(4, 1, 15)), (rtk.point(0.3, 2.0, 9.0), rtk.point(-3, 6, -2)),
                    (rtk.point(1, 30, 10), rtk.point(1, 0, 10.001))):
        check_against_oracle(dev, oracle, scene, _field_camera(rtk, E, 128, 72, eye, to), "inside field")
    big = rtk.Sphere().with_material(rtk.Material.default().with_color(rtk.rgb(0.7, 0.2, 0.2))) \
        .with_transform(rtk.Matrix4x4.translate(0.3, 2.0, 9.0) * rtk.Matrix4x4.scale(3, 3, 3))
    scene.add_object(big)
    check_against_oracle(dev, oracle, scene, _field_camera(rtk, E, 128, 72, rtk.point(0.3, 2.0, 9.0), rtk.point(4, 1, 15)), "inside a sphere")


def test_sphere_counts_around_the_cluster_size(rtk, dev, oracle):
    """63, 64, 65, 128, 129 spheres: partial last cluster, padding records."""
    for n in (1, 7, 63, 64, 65, 128, 129):
        scene = rtk.synthetic.make_scene(n, True, 3, 77 + n)
        camera = rtk.synthetic.make_camera(n, 160, 90)
        check_against_oracle(dev, oracle, scene, camera, f"{n} spheres")


def test_large_coordinates(rtk, dev, oracle):
    """The field far from the world origin (FP32 records are recentred on the field, the
    lights' quirk term scales with the distance from the origin)."""
    rng = np.random.default_rng(21)
    E = 10.0
    scene = _random_field(rtk, rng, 200, False, 0.2, 1.5, E, [rtk.vec3(-6, 18, -4), rtk.vec3(7, 15, 25)], planes=0)
    M = rtk.Matrix4x4
    shift = M.translate(3000.0, -200.0, 1500.0)
    for node in scene.nodes:
        node.transform = shift * node.transform
    scene.lights = [rtk.PointLight(rtk.vec3(2990.0, -170.0, 1490.0), rtk.rgb(0.6, 0.6, 0.6)),
                    rtk.PointLight(rtk.vec3(3010.0, -180.0, 1530.0), rtk.rgb(0.4, 0.4, 0.4))]
    camera = rtk.Camera(160, 90, rtk.PI / 3)
    camera.transform = M.look_at(rtk.point(3000.0, -195.0, 1494.0), rtk.point(3000.0, -200.0, 1509.0), rtk.vec3(0, 1, 0))
    check_against_oracle(dev, oracle, scene, camera, "large coordinates")


def test_transformed_patterns(rtk, dev, oracle):
    """`TransformPattern` around each pattern kind (src/graphics/patterns.rs:7-35), on spheres and
    planes, opaque (shading kernel) and with a reflective object in the scene (single kernel);
    a singular pattern transform samples the untransformed point."""
    M = rtk.Matrix4x4
    mat = rtk.Material.default()
    T = rtk.Texture
    for reflective in (0.0, 0.5):
        scene = rtk.Scene()
        scene.add_light(rtk.PointLight(rtk.vec3(-10, 10, -10), rtk.rgb(0.8, 0.8, 0.8)))
        scene.add_light(rtk.PointLight(rtk.vec3(6, 12, -3), rtk.rgb(0.3, 0.3, 0.3)))
        scene.add_object(rtk.Plane().with_material(mat.with_reflective(reflective).with_pattern(
            T.checker(rtk.WHITE, rtk.rgb(0.1, 0.1, 0.1)).with_transform(M.rotate_y(0.5) * M.scale(0.37, 0.37, 0.37)))))
        scene.add_object(rtk.Sphere().with_transform(M.translate(-1.6, 1, 0.5)).with_material(mat.with_pattern(
            T.striped(rtk.rgb(1, 0.2, 0.2), rtk.rgb(0.2, 0.2, 1)).with_transform(M.rotate_z(0.7) * M.scale(0.2, 0.2, 0.2)))))
        scene.add_object(rtk.Sphere().with_transform(M.translate(0.6, 1, 0) * M.scale(1, 0.6, 1)).with_material(mat.with_pattern(
            T.gradient(rtk.rgb(0, 1, 0), rtk.rgb(1, 0, 1)).with_transform(M.translate(0.3, 0, 0) * M.shear(0.5, 0, 0, 0, 0.2, 0)))))
        scene.add_object(rtk.Sphere().with_transform(M.translate(2.4, 0.7, -0.8) * M.scale(0.7, 0.7, 0.7)).with_material(mat.with_pattern(
            T.ring(rtk.rgb(1, 1, 0), rtk.rgb(0, 0.3, 0.3)).with_transform(M.scale(0.15, 1, 0.15)).with_transform(M.rotate_x(1.0)))))
        scene.add_object(rtk.Sphere().with_transform(M.translate(-0.4, 0.4, -1.6) * M.scale(0.4, 0.4, 0.4)).with_material(mat.with_pattern(
            T.striped(rtk.WHITE, rtk.BLACK).with_transform(M.scale(0, 1, 1)))))
        check_against_oracle(dev, oracle, scene, rtk.Camera(480, 270, rtk.PI / 3), f"transformed patterns, reflective={reflective}")


def test_nested_transformed_patterns(rtk, dev, oracle):
    """TransformPatterns inside TransformPatterns (the type is generic over its pattern, src/graphics/patterns.rs:7-35):
    every wrapper's inverse in turn, outermost first; singular wrappers drop out one by one; a wrapper need not be
    affine (w then travels down the chain).  Opaque (fused kernel) and with a reflective floor (single kernel)."""
    M = rtk.Matrix4x4
    mat = rtk.Material.default()
    T = rtk.Texture
    projective = M(list(M.scale(0.4, 0.4, 0.4).e[:12]) + [0.05, 0.0, 0.1, 1.0])
    for reflective in (0.0, 0.5):
        scene = rtk.Scene()
        scene.add_light(rtk.PointLight(rtk.vec3(-10, 10, -10), rtk.rgb(0.8, 0.8, 0.8)))
        scene.add_light(rtk.PointLight(rtk.vec3(6, 12, -3), rtk.rgb(0.3, 0.3, 0.3)))
        scene.add_object(rtk.Plane().with_material(mat.with_reflective(reflective).with_pattern(
            T.checker(rtk.WHITE, rtk.rgb(0.1, 0.1, 0.1)).with_transform(M.scale(0.37, 0.37, 0.37)).wrap().with_transform(M.rotate_y(0.5))
            .wrap().with_transform(M.translate(0.21, 0, -0.4)))))
        scene.add_object(rtk.Sphere().with_transform(M.translate(-1.6, 1, 0.5)).with_material(mat.with_pattern(
            T.striped(rtk.rgb(1, 0.2, 0.2), rtk.rgb(0.2, 0.2, 1)).with_transform(M.scale(0.2, 0.2, 0.2)).wrap().with_transform(M.rotate_z(0.7)))))
        # a singular wrapper between two regular ones, and one at each end
        scene.add_object(rtk.Sphere().with_transform(M.translate(0.6, 1, 0) * M.scale(1, 0.6, 1)).with_material(mat.with_pattern(
            T.gradient(rtk.rgb(0, 1, 0), rtk.rgb(1, 0, 1)).with_transform(M.shear(0.5, 0, 0, 0, 0.2, 0)).wrap().with_transform(M.scale(1, 0, 1))
            .wrap().with_transform(M.translate(0.3, 0, 0)))))
        scene.add_object(rtk.Sphere().with_transform(M.translate(2.4, 0.7, -0.8) * M.scale(0.7, 0.7, 0.7)).with_material(mat.with_pattern(
            T.ring(rtk.rgb(1, 1, 0), rtk.rgb(0, 0.3, 0.3)).with_transform(M.scale(0, 1, 1)).wrap().with_transform(M.scale(0.15, 1, 0.15))
            .wrap().with_transform(M.scale(1, 1, 0)))))
        # all wrappers singular: the pattern sees the object-space point
        scene.add_object(rtk.Sphere().with_transform(M.translate(-0.4, 0.4, -1.6) * M.scale(0.4, 0.4, 0.4)).with_material(mat.with_pattern(
            T.striped(rtk.WHITE, rtk.BLACK).with_transform(M.scale(0, 1, 1)).wrap().with_transform(M.scale(1, 1, 0)))))
        # a wrapper that is not affine: its inverse changes w, which the next wrapper's translation then scales
        scene.add_object(rtk.Sphere().with_transform(M.translate(1.2, 0.5, -2.0) * M.scale(0.5, 0.5, 0.5)).with_material(mat.with_pattern(
            T.checker(rtk.rgb(0.9, 0.5, 0.1), rtk.rgb(0.1, 0.5, 0.9)).with_transform(M.translate(0.3, 0.2, 0.1) * M.scale(0.5, 0.5, 0.5))
            .wrap().with_transform(projective))))
        check_against_oracle(dev, oracle, scene, rtk.Camera(480, 270, rtk.PI / 3), f"nested patterns, reflective={reflective}")


def test_large_scene_with_secondary_rays(rtk, dev, oracle):
    """400 and 1000 spheres, a quarter of them glass, reflective floor: reflected and refracted
    rays leave a wavefront in all directions, so opened clusters are swept sphere-per-lane."""
    for n, w, h in ((400, 192, 108), (1000, 128, 72)):
        scene = rtk.synthetic.make_scene(n, True, 3, 4242 + n, glass=True)
        camera = rtk.synthetic.make_camera(n, w, h)
        c = check_against_oracle(dev, oracle, scene, camera, f"{n} spheres with glass")
        assert c.stats.rays_secondary > w * h


def _glass_field(rtk, rng, n, r_lo, r_hi, extent, lights, affine=False, indices=(1.0,)):
    """A volume of spheres (not a layer), a third of them glass with refractive indices from `indices`, over a
    reflective floor: reflected and refracted rays cross many cells of the grid they walk, in every direction."""
    M = rtk.Matrix4x4
    scene = rtk.Scene()
    for pos in lights:
        scene.add_light(rtk.PointLight(pos, rtk.rgb(*(1.0 / len(lights),) * 3)))
    mat = rtk.Material.default()
    scene.add_object(rtk.Plane().with_material(mat.with_pattern(rtk.Texture.checker(rtk.WHITE, rtk.BLACK)).with_reflective(0.5)))
    for i in range(n):
        r = float(np.exp(rng.uniform(np.log(r_lo), np.log(r_hi))))
        c = (rng.uniform(-extent, extent), rng.uniform(r, 0.8 * extent), rng.uniform(0.0, 2 * extent))
        t = M.translate(*c)
        if affine and i % 3 == 1:
            t = t * M.rotate_y(rng.uniform(0, 6.28)) * M.rotate_x(rng.uniform(0, 6.28)) * M.scale(r * rng.uniform(0.3, 1.0), r * rng.uniform(0.3, 1.0), r)
        else:
            t = t * M.scale(r, r, r)
        m = mat.with_color(rtk.rgb(*rng.uniform(0.2, 1.0, size=3)))
        if i % 3 == 0:
            m = m.with_transparency(0.9).with_reflective(0.8).with_ambient(0.1).with_diffuse(0.2).with_refractivity(
                float(indices[i // 3 % len(indices)]))
        scene.add_object(rtk.Sphere().with_material(m).with_transform(t))
    return scene


GRID_CASES = ["equal indices", "different indices", "affine spheres", "point lights", "point and w = 0 lights",
              "radii over three decades", "far from the origin", "camera among the spheres", "camera inside a glass sphere"]


@pytest.mark.parametrize("case", GRID_CASES)
def test_secondary_rays_through_the_cell_grid(rtk, dev, oracle, case):
    """Reflected and refracted rays walk the scene's cell grid lane by lane (forward for the closest hit,
    backward for the entry before it when refractive indices differ); shadow rays of their hits towards lights
    built from points walk it too.  Both trace orders against the oracle, pixels and ray counts."""
    rng = np.random.default_rng(11 + GRID_CASES.index(case))
    M = rtk.Matrix4x4
    extent = 12.0
    w0 = [rtk.vec3(-8, 30, -12), rtk.vec3(10, 25, -5), rtk.vec3(2, 40, 6)]
    w1 = [rtk.point(-8, 30, -12), rtk.point(10, 25, -5), rtk.point(0.0, 3.0, 9.0)]  # the last one inside the field
    kw = dict(n=500, r_lo=0.3, r_hi=1.2, extent=extent, lights=w0)
    if case == "different indices":
        kw["indices"] = (1.0, 1.33, 1.5, 2.4)
    elif case == "affine spheres":
        kw.update(affine=True, indices=(1.0, 1.5))
    elif case == "point lights":
        kw.update(lights=w1, indices=(1.0, 1.5))
    elif case == "point and w = 0 lights":
        kw["lights"] = [w1[0], w0[1], w1[2]]
    elif case == "radii over three decades":  # lists too long for small cells: the grid's cells grow, or the scene keeps its sweeps
        kw.update(n=700, r_lo=0.004, r_hi=4.0)
    scene = _glass_field(rtk, rng, **kw)
    camera = _field_camera(rtk, extent, 144, 81)
    if case == "camera among the spheres":  # spheres behind the camera: the entry before a camera ray's hit (n1) lies there
        kw["indices"] = (1.2, 1.5, 2.0)
        scene = _glass_field(rtk, rng, **kw)
        camera = _field_camera(rtk, extent, 144, 81, eye=rtk.point(0.3, 4.0, 11.0), to=rtk.point(2.0, 2.0, 20.0))
    if case == "camera inside a glass sphere":
        kw["indices"] = (1.2, 1.5, 2.0)
        scene = _glass_field(rtk, rng, **kw)
        glass = rtk.Material.default().with_transparency(0.9).with_reflective(0.5).with_diffuse(0.1).with_refractivity(1.7)
        scene.add_object(rtk.Sphere().with_material(glass).with_transform(M.translate(0.0, 5.0, -4.0) * M.scale(2.5, 2.5, 2.5)))
        camera = _field_camera(rtk, extent, 144, 81, eye=rtk.point(0.2, 5.3, -4.4), to=rtk.point(0.0, 3.0, 12.0))
    if case == "far from the origin":
        shift = M.translate(3000.0, -200.0, 7000.0)
        for node in scene.nodes:
            node.transform = shift * node.transform
        camera.transform = camera.transform * M.translate(-3000.0, 200.0, -7000.0)
    for order in (rtk.abi.RTK_TRACE_BY_LEVEL, rtk.abi.RTK_TRACE_DEPTH_FIRST):
        dev.set_trace_order(order, 1 << 22)
        c = check_against_oracle(dev, oracle, scene, camera, f"cell grid, {case}, order {order}")
        assert c.stats.rays_secondary > 144 * 81 // 2
    dev.set_trace_order(rtk.abi.RTK_TRACE_AUTO, 64 << 20)


@pytest.mark.parametrize("case", ["100,000 spheres in a volume", "100,000 spheres, point lights", "40 spheres in one place",
                                  "33 spheres (the smallest scene with a grid)"])
def test_cell_grid_sizes(rtk, dev, oracle, case):
    """The grid at its limits: a volume of 100,000 spheres (cells capped per axis, lists of several entries), all
    spheres in one cell, and the smallest scene that gets a grid; both trace orders."""
    n, extent, seed = {"1": (100000, 60.0, 1), "4": (40, 5.0, 3), "3": (33, 5.0, 4)}[case[0]]
    seed += case.endswith("point lights")
    rng = np.random.default_rng(seed)
    lights = [rtk.vec3(-50, 200, -80), rtk.vec3(80, 150, 20)]
    if case.endswith("point lights"):
        lights = [rtk.point(-50, 200, -80), rtk.point(10, 30, 40)]
    scene = _glass_field(rtk, rng, n, 0.2 if n > 1000 else 0.4, 0.8 if n > 1000 else 1.0, extent, lights, indices=(1.0, 1.5))
    if case.startswith("40"):
        M = rtk.Matrix4x4
        for i, node in enumerate(scene.nodes[1:]):
            r = node.transform.e[0]
            node.transform = M.translate(0.0, 1.5 + 1e-9 * i, 5.0) * M.scale(r, r, r)
    w, h = (48, 27) if n > 1000 else (96, 54)
    camera = _field_camera(rtk, extent, w, h)
    for order in (rtk.abi.RTK_TRACE_BY_LEVEL, rtk.abi.RTK_TRACE_DEPTH_FIRST):
        dev.set_trace_order(order, 1 << 22)
        c = check_against_oracle(dev, oracle, scene, camera, f"{case}, order {order}")
        assert c.stats.rays_secondary > w * h // 2
    dev.set_trace_order(rtk.abi.RTK_TRACE_AUTO, 64 << 20)


def test_secondary_rays_that_start_far_outside_the_grid(rtk, dev, oracle):
    """A mirror floor seen to the horizon and mirror walls 40,000 and 300,000 units away: reflected rays start
    up to hundreds of thousands of cells away from the spheres; those beyond 200,000 cells take the cluster sweep,
    the others enter the grid through its side (the walk starts where the f64 slab test puts it).  (Not further:
    seen from a million scene sizes away, the reference's own discriminants are rounding noise -- their terms
    cancel to less than 2^-53 of their size -- and which spheres then "block" a light is decided by that noise.)"""
    rng = np.random.default_rng(5)
    M = rtk.Matrix4x4
    scene = _glass_field(rtk, rng, 300, 0.3, 1.0, 8.0, [rtk.vec3(-8, 30, -12), rtk.point(5, 20, 0)])
    mirror = rtk.Material.default().with_color(rtk.rgb(0.9, 0.9, 1.0)).with_reflective(0.9).with_diffuse(0.1)
    scene.nodes[0].material = mirror  # the floor
    scene.add_object(rtk.Plane().with_material(mirror).with_transform(M.translate(0, 0, 40000.0) * M.rotate_x(math.pi / 2)))
    scene.add_object(rtk.Plane().with_material(mirror).with_transform(M.translate(-3.0e5, 0, 0) * M.rotate_z(math.pi / 2)))
    camera = rtk.Camera(160, 90, rtk.PI / 2.5)
    camera.transform = M.look_at(rtk.point(0, 1.5, -30), rtk.point(0, 1.0, 10), rtk.vec3(0, 1, 0))
    for order in (rtk.abi.RTK_TRACE_BY_LEVEL, rtk.abi.RTK_TRACE_DEPTH_FIRST):
        dev.set_trace_order(order, 1 << 22)
        check_against_oracle(dev, oracle, scene, camera, f"origins far outside the grid, order {order}")
    dev.set_trace_order(rtk.abi.RTK_TRACE_AUTO, 64 << 20)


def test_random_rays_in_a_large_scene(rtk, dev, oracle):
    """`Scene::trace` for 2048 unrelated rays through 600 spheres (w = 0 and w = 1 lights)."""
    scene = rtk.synthetic.make_scene(600, True, 2, 99, glass=True)
    scene.lights[1] = rtk.PointLight(rtk.point(4.0, 30.0, -6.0), rtk.rgb(0.5, 0.5, 0.5))
    rng = np.random.default_rng(17)
    n = 2048
    o = np.concatenate([rng.uniform(-15, 15, (n, 1)), rng.uniform(0.5, 12, (n, 1)), rng.uniform(-5, 35, (n, 1)), np.ones((n, 1))], axis=1)
    d = np.concatenate([rng.normal(size=(n, 3)), np.zeros((n, 1))], axis=1)
    d[:, 1] -= 0.5
    d[:, :3] /= np.linalg.norm(d[:, :3], axis=1, keepdims=True)
    h = dev.upload(scene)
    got = dev.trace(h, o, d)
    want = np.array([oracle.trace(scene, tuple(o[i]), tuple(d[i])) for i in range(n)])
    err = np.abs(got - want) / np.maximum(np.abs(want), 1e-3)
    assert (err.max(axis=1) > 1e-9).sum() == 0, f"max {err.max():.3e}"


def test_refractive_index_by_the_letter_on_grazing_rays(rtk, dev, oracle):
    """`compute_refractivity` (src/scene/lighting.rs:81-116) compares hits with `Hit::eq` (same
    object, distances within 1e-5), so a sphere whose two roots lie that close together drops
    out of the reference's container list, and a winner whose other root lies just behind the
    origin stops the scan one entry early.  Rays built to do exactly that, inside a large glass
    sphere, with every refractive index different: the colours must still be the oracle's."""
    rng = np.random.default_rng(5)
    M = rtk.Matrix4x4
    scene = rtk.Scene()
    scene.add_light(rtk.PointLight(rtk.vec3(-6, 9, -7), rtk.WHITE))

    def glass(refr, color):
        return (rtk.Material.default().with_color(color).with_diffuse(0.4).with_transparency(0.8).with_reflective(0.3)
                .with_refractivity(refr))

    scene.add_object(rtk.Sphere().with_material(glass(1.5, (0.9, 0.9, 1.0))).with_transform(M.scale(30, 30, 30)))  # around everything
    small = []
    for i in range(8):
        c = rng.normal(size=3) * 6.0
        r = float(rng.uniform(0.6, 1.4))
        small.append((c, r))
        scene.add_object(rtk.Sphere().with_material(glass(1.1 + 0.17 * i, tuple(rng.uniform(0.2, 1, 3))))
                         .with_transform(M.translate(*c) * M.scale(r, r, r)))
    origins, directions = [], []
    for k in range(600):
        c, r = small[k % 8]
        d = rng.normal(size=3)
        d /= np.linalg.norm(d)
        perp = np.cross(d, rng.normal(size=3))
        perp /= np.linalg.norm(perp)
        foot = c + perp * r * (1.0 - float(rng.choice([2e-12, 5e-12, 1e-11])))  # chord of a few 1e-6: both roots within 1e-5
        if k % 3 == 0:
            o = foot - d * float(rng.uniform(-2.5e-6, 2.5e-6))  # the origin inside the sliver: roots straddle zero
        else:
            o = foot + d * float(rng.uniform(0.5, 4.0))         # the grazed sphere right behind the origin
        origins.append((*o, 1.0))
        directions.append((*d, 0.0))
    grazing = 0
    for o, d in zip(origins[:60], directions[:60]):
        hits = sorted(oracle.scene_intersect(scene, o, d), key=lambda h: h[1])
        by_obj = {}
        for obj, t in hits:
            by_obj.setdefault(obj, []).append(t)
        grazing += any(len(ts) == 2 and abs(ts[0] - ts[1]) < 1e-5 for ts in by_obj.values())
    assert grazing >= 40, grazing  # the construction does what it says
    h = dev.upload(scene)
    got = dev.trace(h, np.array(origins), np.array(directions))
    want = np.array([oracle.trace(scene, o, d) for o, d in zip(origins, directions)])
    err = np.abs(got - want) / np.maximum(np.abs(want), 1e-3)
    assert (err.max(axis=1) > 1e-9).sum() == 0, f"{int((err.max(axis=1) > 1e-9).sum())} rays differ, max {err.max():.3e}"
    h.release()


def test_fifty_thousand_spheres(rtk, dev, oracle):
    """782 clusters: several blocks of 64 clusters per wavefront-level sweep."""
    scene = rtk.synthetic.make_scene(50000, True, 4, 555)
    camera = rtk.synthetic.make_camera(50000, 96, 54)
    check_against_oracle(dev, oracle, scene, camera, "50,000 spheres")


def test_point_lights_inside_the_field(rtk, dev, oracle):
    """w = 1 lights between the spheres and right above the floor: for patches next to a light the
    cone of shadow rays opens up to everything (the light lies inside the ball of shading points)."""
    rng = np.random.default_rng(31)
    E = 10.0
    lights = [rtk.point(0.3, 0.8, 9.0), rtk.point(-4.0, 0.02, 5.0), rtk.point(2.0, 6.0, 14.0)]
    scene = _random_field(rtk, rng, 500, True, 0.15, 1.2, E, lights)
    check_against_oracle(dev, oracle, scene, _field_camera(rtk, E, 192, 108), "point lights inside")
    check_against_oracle(dev, oracle, scene, _field_camera(rtk, E, 128, 72, rtk.point(0.3, 2.0, 9.0), rtk.point(4, 0.5, 15)),
                         "point lights inside, camera inside")
    # one light of each kind in one batch, and more than one batch
    scene.lights = [rtk.PointLight(rtk.vec3(-6, 18, -4), rtk.rgb(0.3, 0.3, 0.3)), rtk.PointLight(lights[0], rtk.rgb(0.3, 0.3, 0.3)),
                    rtk.PointLight(rtk.vec3(7, 15, 25), rtk.rgb(0.2, 0.2, 0.2)), rtk.PointLight(lights[2], rtk.rgb(0.2, 0.2, 0.2)),
                    rtk.PointLight(lights[1], rtk.rgb(0.2, 0.2, 0.2)), rtk.PointLight(rtk.vec3(0.5, 9, 3), rtk.rgb(0.2, 0.2, 0.2))]
    check_against_oracle(dev, oracle, scene, _field_camera(rtk, E, 160, 90), "six mixed lights")


def test_randomised_scenes(rtk, dev, oracle):
    """120 scenes drawn from wide ranges (tests/fuzz_parity.py): radii over three decades, fields far
    from the origin, lights of every kind near and far, affine spheres, patterns, glass."""
    from fuzz_parity import random_scene

    for seed in range(2000, 2120):
        scene, camera, what = random_scene(rtk, np.random.default_rng(seed))
        check_against_oracle(dev, oracle, scene, camera, f"seed {seed}: {what}")


def test_randomised_scenes_with_ambient_colour_and_depth_limits(rtk, dev, oracle):
    """The same generator with a random ambient colour and a depth limit from 1 to 8 per scene, left to the
    library's choice of order and forced level by level: children cut off by the depth limit add the ambient
    colour times their weight, which the level-by-level order keeps in records of their own."""
    from fuzz_parity import random_scene, vary_scene_fields

    for order in (rtk.abi.RTK_TRACE_AUTO, rtk.abi.RTK_TRACE_BY_LEVEL):
        dev.set_trace_order(order, 0)
        for seed in range(5000, 5040):
            scene, camera, what = random_scene(rtk, np.random.default_rng(seed))
            what += vary_scene_fields(rtk, scene, seed)
            check_against_oracle(dev, oracle, scene, camera, f"seed {seed}, order {order}: {what}")
    dev.set_trace_order(rtk.abi.RTK_TRACE_AUTO, 0)


# ---------------------------------------------------------------- more than one sample per pixel
# The reference has one sample per pixel (src/scene/cameras.rs:53-54): the averaging step is
# checked against the oracle's restatement of this library's own definition (parity unpinned);
# the samples themselves are pixels of the reference's camera model and are pinned as above.
def check_sampled(dev, oracle, scene, camera, s, what):
    canvas = dev.render(scene, camera, samples_per_axis=s)
    rgb, rgba, ost = oracle.render(scene, camera, threads=16, samples_per_axis=s)
    assert counts(canvas.stats) == counts(ost), what
    assert canvas.stats.rays_primary == camera.width * camera.height * s * s
    max_rel, _, _, not_equal = assert_parity(canvas.rgb, canvas.rgba8, rgb, rgba, what)
    assert not_equal == 0, f"{what}: {not_equal} pixels differ after the 8-bit cast"
    # FP32 samples equal the oracle's, so the f64 mean of them does, and so does its narrowing
    assert (canvas.rgb == rgb.astype(np.float32)).all(), f"{what}: max relative error {max_rel:.3e}"
    assert (canvas.rgba8[..., 3] == 255).all()
    return canvas


@pytest.mark.parametrize("name,w,h,s", [("C2", 320, 180, 2), ("C2", 213, 120, 3), ("C2", 160, 90, 4), ("C3", 192, 108, 2),
                                        ("C2g", 160, 90, 2), ("C2", 37, 19, 8), ("C2", 8, 5, 16)])
def test_sampled_render_matches_the_oracle(rtk, dev, oracle, name, w, h, s):
    scene, camera = rtk.synthetic.config(name, w, h)
    check_sampled(dev, oracle, scene, camera, s, f"{name} {w}x{h} s={s}")


def test_sampled_bundled_scene(rtk, dev, oracle, golden_dir):
    scene = rtk.PackedScene.from_yaml_file(os.path.join(golden_dir, "test01.yaml")).build()
    check_sampled(dev, oracle, scene, rtk.Camera(240, 135, rtk.PI / 3.0), 2, "C1 240x135 s=2")


def test_one_sample_per_axis_is_the_plain_render(rtk, dev):
    scene, camera = rtk.synthetic.config("C2", 640, 360)
    a = dev.render(scene, camera)
    b = dev.render(scene, camera, samples_per_axis=1)
    assert (a.rgb == b.rgb).all() and (a.rgba8 == b.rgba8).all() and counts(a.stats) == counts(b.stats)


def test_sampled_device_render_and_resolve_alone(rtk, dev):
    """Device outputs: the sampled call equals rendering the finer camera and averaging it with
    rtk_resolve; a smaller frame after a larger one reuses the context's sample buffer."""
    import torch

    scene, _ = rtk.synthetic.config("C2", 8, 8)
    h = dev.upload(scene)
    for (w, hh, s) in [(256, 144, 3), (128, 72, 2)]:
        _, camera = rtk.synthetic.config("C2", w, hh)
        rgb, rgba = dev.render_device_sampled(h, camera, s)
        fine, _ = dev.render_device(h, camera.supersampled(s), want_rgba8=False)
        rgb2, rgba2 = dev.resolve(fine, s)
        torch.cuda.synchronize()
        assert torch.equal(rgb, rgb2) and torch.equal(rgba, rgba2)
        f = fine.cpu().numpy().astype(np.float64).reshape(hh, s, w, s, 3)
        acc = np.zeros((hh, w, 3))
        for j in range(s):
            for i in range(s):
                acc = acc + f[:, j, :, i, :]
        mean = acc / float(s * s)
        assert (rgb.cpu().numpy() == mean.astype(np.float32)).all()
        v = mean * 255.0
        expect8 = np.where(v > 0.0, np.minimum(v, 255.0), 0.0).astype(np.uint8)
        assert (rgba.cpu().numpy()[..., :3] == expect8).all()
    h.release()


def test_resolve_of_special_values(rtk, dev):
    """Saturation, negative and NaN samples go through the 8-bit cast like a single colour does."""
    import torch

    s = 2
    samples = torch.zeros((2 * s, 3 * s, 3), dtype=torch.float32, device=dev.tdev)
    samples[0:2, 0:2] = 2.0                     # pixel (0,0): saturates
    samples[0:2, 2:4] = -1.0                    # pixel (1,0): negative -> 0
    samples[0:2, 4:6] = float("nan")            # pixel (2,0): NaN -> 0
    samples[2:4, 0:2, 0] = torch.tensor([[0.25, 0.5], [0.75, 1.0]], device=dev.tdev)  # mean 0.625 -> 159
    rgb, rgba = dev.resolve(samples, s)
    torch.cuda.synchronize()
    rgba = rgba.cpu().numpy()
    assert tuple(rgba[0, 0]) == (255, 255, 255, 255) and tuple(rgba[0, 1]) == (0, 0, 0, 255)
    assert tuple(rgba[0, 2]) == (0, 0, 0, 255) and tuple(rgba[1, 0]) == (159, 0, 0, 255)
    assert float(rgb[1, 0, 0]) == 0.625 and bool(torch.isnan(rgb[0, 2]).all())


def test_largest_configuration_crops(rtk, dev, oracle):
    """C5 at its full 16384 x 16384 (268 M pixels, 8 lights): the frame stays on the device, two
    rectangles of it are checked against the oracle, and the ray count respects its bounds."""
    import torch

    scene, camera = rtk.synthetic.config("C5")
    assert (camera.width, camera.height) == (16384, 16384) and len(scene.lights) == 8
    h = dev.upload(scene)
    rgb, rgba = dev.render_device(h, camera)
    st = dev.stats()
    assert st.rays_primary == 16384 * 16384 and st.rays_secondary == 0
    assert st.rays_primary <= st.rays <= st.rays_primary * 9
    for rect in [(8128, 9000, 8256, 9096), (16256, 16300, 16384, 16384)]:
        x0, y0, x1, y1 = rect
        want_rgb, want_rgba, _ = oracle.render(scene, camera, rect=rect, threads=16)
        g_rgb, g_rgba = rgb[y0:y1, x0:x1].cpu().numpy(), rgba[y0:y1, x0:x1].cpu().numpy()
        max_rel, _, _, not_equal = assert_parity(g_rgb, g_rgba, want_rgb, want_rgba, f"C5 crop {rect}")
        assert not_equal == 0 and max_rel < 1e-6
    del rgb, rgba
    torch.cuda.empty_cache()
    h.release()


def test_many_lights(rtk, dev, oracle):
    """`Scene::add_light` (src/scene.rs:131-133) has no limit: 64 and 300 lights, as scene-file lights
    (w = 0, 16 / 75 batches of four), as lights built from points (w = 1), and with w = 0.5, which sends
    the frame down the three-kernel path whose hit records count a hit's shadow rays in 15 bits."""
    for n_lights, w, size in ((64, 0.0, (160, 90)), (64, 1.0, (160, 90)), (300, 0.0, (96, 54)), (300, 1.0, (96, 54)), (300, 0.5, (96, 54))):
        scene = rtk.synthetic.make_scene(30, True, n_lights, 4321)
        scene.lights = [rtk.PointLight(rtk.vec4(*l.position[:3], w), l.intensity) for l in scene.lights]
        camera = rtk.synthetic.make_camera(30, *size)
        canvas = check_against_oracle(dev, oracle, scene, camera, f"{n_lights} lights, w = {w}")
        assert canvas.stats.rays_shadow <= n_lights * canvas.stats.rays_primary
