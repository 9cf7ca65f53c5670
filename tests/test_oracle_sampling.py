"""The oracle's multi-sample render.  The reference has one sample per pixel
(src/scene/cameras.rs:53-54), so nothing of the reference pins the averaging step: parity
unpinned for samples_per_axis > 1.  What is pinned: every sample is a pixel of the reference's
camera model at a higher resolution, and one sample per axis is the reference's render."""
import os

import numpy as np


def _scene(rtk, golden_dir):
    return rtk.PackedScene.from_yaml_file(os.path.join(golden_dir, "test01.yaml")).build()


def test_one_sample_per_axis_is_the_plain_render(rtk, oracle, golden_dir):
    import ctypes as C

    scene, camera = _scene(rtk, golden_dir), rtk.Camera(96, 54, rtk.PI / 3.0)
    rgb, rgba, st = oracle.render(scene, camera)
    s, c = scene.to_abi(), camera.to_abi()
    rgb1 = np.zeros_like(rgb)
    rgba1 = np.zeros_like(rgba)
    st1 = rtk.abi.rtk_stats()
    rc = oracle.lib().oracle_render_sampled(C.byref(s), C.byref(c), 1, 0, 0, 96, 54, rgb1.ctypes.data_as(C.c_void_p),
                                            rgba1.ctypes.data_as(C.c_void_p), C.byref(st1), 2)
    assert rc == 0
    # the mean of one FP32 sample is that sample
    assert (rgb1 == rgb.astype(np.float32).astype(np.float64)).all()
    assert (st1.rays_primary, st1.rays_shadow, st1.rays_secondary) == (st.rays_primary, st.rays_shadow, st.rays_secondary)
    # the 8-bit cast of the FP32-narrowed colour may differ from the cast of the f64 colour only at a boundary
    assert np.abs(rgba1.astype(int) - rgba.astype(int)).max() <= 1


def test_samples_are_pixels_of_the_finer_camera(rtk, oracle, golden_dir):
    scene, camera = _scene(rtk, golden_dir), rtk.Camera(48, 27, rtk.PI / 3.0)
    for s in (2, 3):
        fine, _, fst = oracle.render(scene, camera.supersampled(s), want_rgba8=False, threads=4)
        rgb, rgba, st = oracle.render(scene, camera, samples_per_axis=s, threads=4)
        assert st.rays_primary == 48 * 27 * s * s == fst.rays_primary
        assert (st.rays_shadow, st.rays_secondary) == (fst.rays_shadow, fst.rays_secondary)
        f32 = fine.astype(np.float32).astype(np.float64).reshape(27, s, 48, s, 3)
        acc = np.zeros((27, 48, 3))
        for j in range(s):  # row-major order of the samples, as the definition states
            for i in range(s):
                acc = acc + f32[:, j, :, i, :]
        mean = acc / float(s * s)
        assert (rgb == mean).all()
        v = mean * 255.0
        expect8 = np.where(v > 0.0, np.minimum(v, 255.0), 0.0).astype(np.uint8)  # truncation, saturation
        assert (rgba[..., :3] == expect8).all() and (rgba[..., 3] == 255).all()


def test_crop_of_a_sampled_render(rtk, oracle, golden_dir):
    scene, camera = _scene(rtk, golden_dir), rtk.Camera(64, 36, rtk.PI / 3.0)
    full, full8, _ = oracle.render(scene, camera, samples_per_axis=2)
    part, part8, _ = oracle.render(scene, camera, rect=(10, 20, 40, 36), samples_per_axis=2)
    assert (part == full[20:36, 10:40]).all() and (part8 == full8[20:36, 10:40]).all()
