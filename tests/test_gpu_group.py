"""The multi-GPU frame from one process (rtk_group_*), on one GPU: a device named several times
makes members that exchange through RCCL's send-to-self, so bands, sends, receives in place and the
two-frames-in-flight buffers all run; the frame must equal the single-context render bit for bit."""
import importlib
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# RCCL in a process of its own: a communicator that failed to come up must not take the other tests with it
SCRIPT = textwrap.dedent(r"""
    import importlib, sys
    sys.path.insert(0, sys.argv[1])
    import torch
    rtk = importlib.import_module("raytracer-rust_amd")
    device = importlib.import_module("raytracer-rust_amd.device")
    group = importlib.import_module("raytracer-rust_amd.group")
    dev = device.Device(0)

    def check(name, w, h, members, frames=5, other_camera=None, bands_per_member=0):
        scene, camera = rtk.synthetic.config(name, w, h)
        handle = dev.upload(scene)
        want_rgb, want_rgba = dev.render_device(handle, camera)
        want = dev.stats()
        torch.cuda.synchronize()
        g = group.DeviceGroup([0] * members)
        assert len(g) == members
        assert g.exchange() == ("none" if members == 1 else sys.argv[2])
        g.upload(scene)
        g.set_bands_per_member(bands_per_member)
        stream = torch.cuda.Stream()
        with torch.cuda.stream(stream):
            outs = [(torch.zeros_like(want_rgb), torch.zeros_like(want_rgba)) for _ in range(2)]
            for k in range(frames):  # queued back to back, two output buffers in turn
                o = outs[k % 2]
                if k >= 2:
                    o[0].zero_(); o[1].zero_()
                g.render_device(camera, out_rgb=o[0], out_rgba8=o[1])
                if k >= 1:  # the frame before this one is complete on the stream by now
                    pass
            stream.synchronize()
            for o in outs[: min(frames, 2)]:
                assert torch.equal(o[0], want_rgb) and torch.equal(o[1], want_rgba), (name, members)
            st = g.stats()
            assert (st.rays_primary, st.rays_shadow, st.rays_secondary, st.exact_tests) == \
                   (want.rays_primary, want.rays_shadow, want.rays_secondary, want.exact_tests), (name, members)
            assert st.kernel_ms > 0.0 and (members == 1 or st.gather_ms > 0.0)
            # one output only, and another view on the same group
            only = g.render_device(camera, want_rgb=False)
            stream.synchronize()
            assert only[0] is None and torch.equal(only[1], want_rgba)
            if other_camera is not None:
                w_rgb, w_rgba = dev.render_device(handle, other_camera)
                got = g.render_device(other_camera)
                stream.synchronize()
                torch.cuda.synchronize()
                assert torch.equal(got[0], w_rgb) and torch.equal(got[1], w_rgba)
        g.close()
        handle.release()
        print("ok", name, w, h, members, "gather_ms %.3f" % st.gather_ms, flush=True)

    check("C2", 480, 270, 1)
    check("C2", 480, 270, 2)
    check("C3", 641, 363, 3)
    check("C2g", 250, 131, 8)
    _, cam2 = rtk.synthetic.config("C3", 320, 200)
    cam2.transform = rtk.Matrix4x4.look_at(rtk.point(20, 9, -30), rtk.point(-5, 0, 25), rtk.vec3(0, 1, 0))
    check("C3", 1920, 1080, 4, other_camera=cam2)
    check("C3", 1920, 1080, 4, bands_per_member=32)
    check("C2", 480, 270, 3, bands_per_member=1000)  # bands of one tile row
    check("C2", 40, 9, 4)      # fewer bands than members: some members own nothing
    # a scene replaced on a live group
    g = group.DeviceGroup([0, 0])
    for name in ("C2", "C3"):
        scene, camera = rtk.synthetic.config(name, 200, 120)
        handle = dev.upload(scene)
        want = dev.render_device(handle, camera)
        g.upload(scene)
        got = g.render_device(camera)
        torch.cuda.synchronize()
        assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
    g.close()
    print("GROUP_OK")
""")


@pytest.mark.parametrize("setting,exchange", [("", "rccl"), ("rccl", "rccl"), ("peer", "peer copies")])
def test_group_frames_on_one_gpu(tmp_path, setting, exchange):
    """Through RCCL (the default) and through the peer-to-peer copies a group falls back to without it."""
    script = tmp_path / "group_frames.py"
    script.write_text(SCRIPT)
    env = dict(os.environ, RTK_GROUP_EXCHANGE=setting)
    r = subprocess.run([sys.executable, str(script), ROOT, exchange], capture_output=True, text=True, timeout=280, env=env)
    assert r.returncode == 0 and "GROUP_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-4000:]


def test_group_argument_errors(rtk):
    import ctypes as C

    device = importlib.import_module("raytracer-rust_amd.device")
    lib = device.library()
    g = C.c_void_p()
    assert lib.rtk_group_create(None, 1, C.byref(g)) == rtk.abi.RTK_ERR_INVALID
    ids = (C.c_int * 1)(0)
    assert lib.rtk_group_create(ids, 0, C.byref(g)) == rtk.abi.RTK_ERR_INVALID
    bad = (C.c_int * 2)(0, 99)
    assert lib.rtk_group_create(bad, 2, C.byref(g)) == rtk.abi.RTK_ERR_NO_DEVICE and not g.value
    assert lib.rtk_group_create(ids, 1, C.byref(g)) == 0
    cam = rtk.Camera(64, 36, 1.0).to_abi()
    assert lib.rtk_group_render_device(g, C.byref(cam), None, None, None) == rtk.abi.RTK_ERR_INVALID  # no scene yet
    assert b"rtk_group_upload" in lib.rtk_last_error()
    assert lib.rtk_group_size(g) == 1 and lib.rtk_group_size(None) == 0
    assert lib.rtk_group_exchange(g) == b"none" and lib.rtk_group_exchange(None) == b"none"
    assert lib.rtk_group_band_tile_rows(g, 2160) == 67 and lib.rtk_group_band_tile_rows(g, 8) == 1
    assert lib.rtk_group_set_bands_per_member(g, 32) == 0 and lib.rtk_group_band_tile_rows(g, 2160) == 8
    assert lib.rtk_group_set_bands_per_member(g, 1 << 20) == rtk.abi.RTK_ERR_INVALID
    assert lib.rtk_group_set_bands_per_member(g, 0) == 0 and lib.rtk_group_band_tile_rows(g, 2160) == 67
    assert lib.rtk_group_set_bands_per_member(None, 4) == rtk.abi.RTK_ERR_INVALID
    st = rtk.abi.rtk_stats()
    assert lib.rtk_group_get_stats(g, C.byref(st)) == 0 and st.rays_primary == 0
    lib.rtk_group_destroy(g)
    lib.rtk_group_destroy(None)
    os.environ["RTK_GROUP_EXCHANGE"] = "carrier pigeon"
    try:
        two = (C.c_int * 2)(0, 0)
        assert lib.rtk_group_create(two, 2, C.byref(g)) == rtk.abi.RTK_ERR_INVALID and b"RTK_GROUP_EXCHANGE" in lib.rtk_last_error()
    finally:
        del os.environ["RTK_GROUP_EXCHANGE"]
