"""Randomised parity sweep of the HIP path against the CPU oracle (run by hand on a GPU box):

    python tests/fuzz_parity.py [n_scenes] [seed]          (FUZZ_BIG=1: thousands of spheres per scene;
                                                            FUZZ_GLASS=1: hundreds to thousands of spheres, always
                                                            glass, always w = 0 lights (FUZZ_LIGHTS=w1 / mixed /
                                                            exotic: those instead): the light grids and the
                                                            level-by-level order, at frame sizes either side of
                                                            the library's choice between the two orders;
                                                            FUZZ_FIELDS=1: a random ambient colour and depth limit
                                                            per scene as well)

Scenes are drawn from wide ranges -- sphere radii over three decades, fields far from the origin,
lights of every kind next to and far from the geometry, general affine transforms, every texture,
optional glass -- and rendered at low resolution.  Any pixel that differs after the 8-bit cast, or
any ray count that differs, is reported with the seed that reproduces it.
"""
import importlib
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def random_scene(rtk, rng):
    M, T = rtk.Matrix4x4, rtk.Texture
    scene = rtk.Scene()
    only_glass = bool(os.environ.get("FUZZ_GLASS"))
    n = int(rng.choice([300, 700, 2500, 6000] if only_glass else [2000, 5000, 12000] if os.environ.get("FUZZ_BIG") else [3, 20, 70, 200, 600]))
    extent = float(10.0 ** rng.uniform(0.3, 2.5))
    offset = rng.normal(size=3) * float(rng.choice([0.0, 0.0, 10.0, 3000.0]))
    r_lo, r_hi = extent * 10.0 ** rng.uniform(-3.0, -1.5), extent * 10.0 ** rng.uniform(-1.3, -0.3)
    glass = only_glass or rng.random() < 0.3
    mat = rtk.Material.default()
    for _ in range(int(rng.integers(0, 3))):
        t = M.translate(*(offset + rng.normal(size=3) * extent * 0.1))
        if rng.random() < 0.5:
            t = t * M.rotate_x(rng.uniform(-0.6, 0.6)) * M.rotate_z(rng.uniform(-0.6, 0.6))
        pat = [T.checker, T.ring, T.striped, T.gradient][int(rng.integers(0, 4))](rtk.WHITE, rtk.rgb(*rng.uniform(0, 1, 3)))
        if rng.random() < 0.3:
            pat = pat.with_transform(M.scale(*rng.uniform(0.2, 3.0, 3)) * M.rotate_y(rng.uniform(0, 3)))
        m = mat.with_pattern(pat).with_specular(float(rng.choice([0.0, 0.9])))
        if glass and rng.random() < 0.5:
            m = m.with_reflective(float(rng.uniform(0.1, 0.9)))
        scene.add_object(rtk.Plane().with_material(m).with_transform(t))
    for _ in range(n):
        r = float(np.exp(rng.uniform(np.log(r_lo), np.log(r_hi))))
        c = offset + np.array([rng.uniform(-extent, extent), rng.uniform(0, 0.5 * extent), rng.uniform(0, 2 * extent)])
        t = M.translate(*c)
        if rng.random() < 0.4:
            t = t * M.rotate_y(rng.uniform(0, 6.28)) * M.rotate_x(rng.uniform(0, 6.28)) * M.scale(
                r * rng.uniform(0.2, 1), r * rng.uniform(0.2, 1), r)
        else:
            t = t * M.scale(r, r, r)
        m = mat.with_color(rtk.rgb(*rng.uniform(0.1, 1, 3))).with_shininess(float(rng.choice([200.0, 10.0, 1.0])))
        if glass and rng.random() < 0.25:
            m = m.with_transparency(0.9).with_reflective(0.9).with_ambient(0.2).with_diffuse(0.2).with_refractivity(
                float(rng.choice([1.0, 1.5])))
        scene.add_object(rtk.Sphere().with_material(m).with_transform(t))
    kinds = os.environ.get("FUZZ_LIGHTS") or ("w0" if only_glass else rng.choice(["w0", "w1", "mixed", "exotic"], p=[0.4, 0.25, 0.25, 0.1]))
    for k in range(int(rng.integers(1, 7))):
        near = rng.random() < 0.4
        pos = offset + (rng.normal(size=3) * extent * (0.3 if near else 3.0) + np.array([0, extent * (0.5 if near else 2.0), 0]))
        w = {"w0": 0.0, "w1": 1.0, "mixed": float(k % 2), "exotic": float(rng.choice([0.0, 1.0, 0.5, 2.0]))}[kinds]
        scene.add_light(rtk.PointLight(rtk.vec4(*pos, w), rtk.rgb(*rng.uniform(0.1, 0.6, 3))))
    w, h = (64, 36) if n > 100 else (96, 54)
    camera = rtk.Camera(w, h, float(rng.uniform(0.5, 1.6)))
    eye = offset + np.array([rng.uniform(-0.5, 0.5) * extent, rng.uniform(0.05, 0.8) * extent, rng.uniform(-0.8, 1.0) * extent])
    to = offset + np.array([rng.uniform(-0.3, 0.3) * extent, 0.0, rng.uniform(0.5, 1.5) * extent])
    camera.transform = M.look_at(rtk.point(*eye), rtk.point(*to), rtk.vec3(0, 1, 0))
    return scene, camera, f"n={n} extent={extent:.3g} lights={kinds} glass={glass}"


def vary_scene_fields(rtk, scene, seed):
    """The scene-wide fields too: an ambient colour (some components zero) and a depth limit from 1 to 8.  A
    generator of its own, so that the scene stays the one of the seed."""
    r2 = np.random.default_rng(seed + 7_000_000)
    if r2.random() < 0.7:
        scene.ambient_color = rtk.rgb(*(r2.uniform(0.0, 0.3, 3) * (r2.random(3) < 0.8)))
    scene.max_depth = int(r2.choice([1, 2, 3, 4, 5, 6, 8]))
    return f" ambient={tuple(round(float(v), 3) for v in scene.ambient_color)} depth={scene.max_depth}"


def main():
    n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rtk = importlib.import_module("raytracer-rust_amd")
    device = importlib.import_module("raytracer-rust_amd.device")
    import oracle_binding as oracle

    dev = device.Device(0)
    if os.environ.get("RTK_TRACE_ORDER"):  # 1: depth first, 2: level by level, whatever the library would pick
        dev.set_trace_order(int(os.environ["RTK_TRACE_ORDER"]))
    bad = 0
    for seed in range(seed0, seed0 + n_scenes):
        rng = np.random.default_rng(seed)
        scene, camera, what = random_scene(rtk, rng)
        if os.environ.get("FUZZ_FIELDS"):
            what += vary_scene_fields(rtk, scene, seed)
        canvas = dev.render(scene, camera)
        rgb, rgba, st = oracle.render(scene, camera, threads=16)
        d8 = np.abs(canvas.rgba8.astype(int) - rgba.astype(int)).max(axis=-1)
        with np.errstate(all="ignore"):
            rel = np.abs(canvas.rgb.astype(np.float64) - rgb) / np.maximum(np.abs(rgb), 1e-3)
            # a colour beyond the FP32 range (the reference's unclamped Schlick term can blow up) narrows to
            # infinity on both sides: equal, not infinitely far apart
            rel[np.isinf(canvas.rgb) & (canvas.rgb == rgb.astype(np.float32))] = 0.0
        got = (canvas.stats.rays_primary, canvas.stats.rays_shadow, canvas.stats.rays_secondary)
        want = (st.rays_primary, st.rays_shadow, st.rays_secondary)
        ok = int((d8 > 0).sum()) == 0 and got == want and float(rel.max()) < 1e-6
        bad += not ok
        print(f"seed {seed:4d} {'ok  ' if ok else 'FAIL'} {what}: differing px {int((d8 > 0).sum())}, max rel {rel.max():.2e}, "
              f"rays {got} vs {want}", flush=True)
    dev.close()
    print(f"{n_scenes - bad} of {n_scenes} scenes identical")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
