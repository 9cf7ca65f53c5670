"""Pins the CPU oracle to the reference's rendered image.

output.png was rendered by the reference itself (src/main.rs:17-20: test01.yaml,
1920x1080, FOV pi/3).  The fixture holds every 4th pixel of it plus the sha256 of the
full decoded frame, so the full-frame check needs no access to the reference tree.
"""
import hashlib
import json
import os

import numpy as np
import pytest


@pytest.fixture(scope="module")
def bundled_render(rtk, oracle, golden_dir):
    scene = rtk.PackedScene.from_yaml_file(os.path.join(golden_dir, "test01.yaml")).build()
    camera = rtk.Camera(1920, 1080, rtk.PI / 3.0)
    rgb, rgba, stats = oracle.render(scene, camera, threads=min(8, os.cpu_count() or 1))
    return rgb, rgba, stats


def test_full_frame_matches_reference_image_bit_for_bit(bundled_render, golden_dir):
    _, rgba, _ = bundled_render
    meta = json.load(open(os.path.join(golden_dir, "output_meta.json")))
    assert rgba.shape == (meta["height"], meta["width"], 4)
    assert hashlib.sha256(rgba.tobytes()).hexdigest() == meta["decoded_rgba_sha256"]
    for spot in meta["spots"]:
        assert list(rgba[spot["y"], spot["x"], :3]) == spot["rgb"]


def test_stride4_subsample_matches(bundled_render, golden_dir):
    _, rgba, _ = bundled_render
    golden = np.load(os.path.join(golden_dir, "output_stride4.npy"))
    assert np.array_equal(rgba[::4, ::4, :3], golden)


def test_linear_values_and_ray_counts(bundled_render):
    rgb, _, stats = bundled_render
    assert np.allclose(rgb[540, 960], (0.652815, 0.081615, 0.816015), atol=1e-6)
    assert np.allclose(rgb[435, 690], (0.996176, 0.299790, 1.195143), atol=1e-6)
    assert np.allclose(rgb[1000, 960], (1.020326, 0.690838, 1.114450), atol=1e-6)
    assert stats.rays_primary == 1920 * 1080
    # one `Scene::intersect` call per ray: mean 10.58 per pixel on this scene
    assert stats.rays == 21939338


def test_threaded_render_equals_single_threaded(rtk, oracle, golden_dir):
    scene = rtk.PackedScene.from_yaml_file(os.path.join(golden_dir, "test01.yaml")).build()
    camera = rtk.Camera(192, 108, rtk.PI / 3.0)
    a = oracle.render(scene, camera, threads=1)
    b = oracle.render(scene, camera, threads=3)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert a[2].rays == b[2].rays
    # a rectangle of the frame equals the same window of the full render
    c = oracle.render(scene, camera, rect=(40, 50, 120, 100), threads=2)
    assert np.array_equal(c[0], a[0][50:100, 40:120])
