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

    def expected_exchange(members, w, h, per_member, root_share, rows):
        # (messages per output, pixels that travel) of a frame, from the partition rules of include/rtk.h
        tiles_y = (h + 7) // 8
        if rows is not None:
            period, shares = max(1, tiles_y), [(rows[r], rows[r + 1] - rows[r]) for r in range(members)]
        else:
            per = per_member or (2 if members >= 8 else 3 if members >= 4 else 4)
            vw = members - 1 + root_share
            m = max(1, tiles_y // (vw * per))
            period = vw * m
            shares = [(0, root_share * m)] + [((root_share + r - 1) * m, m) for r in range(1, members)]
        messages = pixels = 0
        for off, width in shares[1:]:
            t0 = off
            while width and t0 < tiles_y:
                t1 = min(tiles_y, t0 + width)
                messages += 1
                pixels += (min(h, t1 * 8) - t0 * 8) * w
                t0 += period
        return messages, pixels

    def check(name, w, h, members, frames=5, other_camera=None, bands_per_member=0, root_share=1, rows=None, balance=False):
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
        g.set_root_share(root_share)
        if rows is not None:
            g.set_rows(h, rows)
        if balance:  # boundaries by measurement: whatever they come out as, the frame is the same
            g.balance(camera, rounds=2)
            i0 = g.info()
            assert i0["partition"] == ("rows" if members > 1 else "bands")
        stream = torch.cuda.Stream()
        with torch.cuda.stream(stream):
            outs = [(torch.zeros_like(want_rgb), torch.zeros_like(want_rgba)) for _ in range(2)]
            for k in range(frames):  # queued back to back, two output buffers in turn
                o = outs[k % 2]
                if k >= 2:
                    o[0].zero_(); o[1].zero_()
                g.render_device(camera, out_rgb=o[0], out_rgba8=o[1])
            stream.synchronize()
            for o in outs[: min(frames, 2)]:
                assert torch.equal(o[0], want_rgb) and torch.equal(o[1], want_rgba), (name, members)
            st = g.stats()
            assert (st.rays_primary, st.rays_shadow, st.rays_secondary, st.exact_tests) == \
                   (want.rays_primary, want.rays_shadow, want.rays_secondary, want.exact_tests), (name, members)
            assert st.kernel_ms > 0.0 and (members == 1 or st.gather_ms > 0.0)
            info = g.info()
            assert info["members"] == members and info["exchange"] == g.exchange() and info["root_share"] == root_share
            assert sum(info["share_tile_rows"]) == (h + 7) // 8 or members > 16
            assert sum(info["share_rays"]) == want.rays or members > 16
            if not balance and members > 1:
                msgs, pixels = expected_exchange(members, w, h, bands_per_member, root_share, rows)
                assert (info["messages"], info["bytes_into_root"]) == (2 * msgs, 16 * pixels), (info, msgs, pixels)
            # the 8-bit frame only: 4 bytes per pixel travel, one message per run of rows
            only = g.render_device(camera, want_rgb=False)
            stream.synchronize()
            assert only[0] is None and torch.equal(only[1], want_rgba)
            info8 = g.info()
            if members > 1:
                assert info8["bytes_into_root"] * 4 == info["bytes_into_root"] and info8["messages"] * 2 == info["messages"], (info8, info)
            if other_camera is not None:
                w_rgb, w_rgba = dev.render_device(handle, other_camera)
                got = g.render_device(other_camera)
                stream.synchronize()
                torch.cuda.synchronize()
                assert torch.equal(got[0], w_rgb) and torch.equal(got[1], w_rgba)
        assert torch.cuda.current_device() == 0  # the group's calls leave the caller's device alone
        g.close()
        handle.release()
        print("ok", name, w, h, members, "gather_ms %.3f" % st.gather_ms, info["share_tile_rows"], info["messages"], flush=True)

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
    # the root holds several bands of every round
    check("C3", 641, 363, 2, root_share=3)
    check("C3", 1920, 1080, 4, root_share=2, bands_per_member=2)
    check("C2", 480, 270, 8, root_share=5)
    check("C2", 40, 9, 4, root_share=2)
    # one run of rows per member: explicit boundaries (an empty run included), then boundaries by measurement
    check("C3", 641, 363, 2, rows=[0, 30, 46])
    check("C3", 1920, 1080, 4, rows=[0, 70, 70, 100, 135])
    check("C2", 480, 270, 8, rows=[0, 10, 12, 14, 18, 22, 26, 30, 34])
    check("C3", 1920, 1080, 2, balance=True)
    check("C3", 1920, 1080, 4, balance=True)
    check("C4", 1280, 720, 8, balance=True)
    # the 8-bit frame alone never allocates FP32 band buffers
    scene, camera = rtk.synthetic.config("C3", 1920, 1080)
    handle = dev.upload(scene)
    want = dev.render_device(handle, camera)
    g = group.DeviceGroup([0] * 4)
    g.upload(scene)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    got = g.render_device(camera, want_rgb=False)
    torch.cuda.synchronize()
    used = free0 - torch.cuda.mem_get_info()[0]
    assert torch.equal(got[1], want[1])
    # the frame itself (8 MB) and the three peers' 8-bit rows, two buffers each: far below the 56 MB of FP32 rows
    assert used < 40 << 20, used
    g.close()
    handle.release()
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
    assert lib.rtk_group_set_root_share(g, 3) == 0 and lib.rtk_group_set_root_share(g, 0) == 0
    assert lib.rtk_group_set_root_share(g, 5000) == rtk.abi.RTK_ERR_INVALID and lib.rtk_group_set_root_share(None, 1) == rtk.abi.RTK_ERR_INVALID
    rows = (C.c_uint32 * 2)(0, 270)
    assert lib.rtk_group_set_rows(g, 2160, rows) == 0 and lib.rtk_group_band_tile_rows(g, 2160) == 270
    assert lib.rtk_group_band_tile_rows(g, 1080) == 33  # boundaries hold for the height they were made for
    assert lib.rtk_group_set_rows(g, 2160, (C.c_uint32 * 2)(0, 269)) == rtk.abi.RTK_ERR_INVALID
    assert lib.rtk_group_set_rows(g, 2160, (C.c_uint32 * 2)(1, 270)) == rtk.abi.RTK_ERR_INVALID
    assert lib.rtk_group_set_rows(g, 0, rows) == rtk.abi.RTK_ERR_INVALID
    assert lib.rtk_group_set_rows(g, 2160, None) == 0 and lib.rtk_group_band_tile_rows(g, 2160) == 67
    info = rtk.abi.rtk_group_info()
    assert lib.rtk_group_get_info(g, C.byref(info)) == 0 and info.n_members == 1 and info.exchange == 0 and info.messages == 0
    assert lib.rtk_group_get_info(g, None) == rtk.abi.RTK_ERR_INVALID
    assert lib.rtk_group_balance(g, C.byref(cam), 1, 1, 0, 0.0) == rtk.abi.RTK_ERR_INVALID  # no scene yet
    assert lib.rtk_group_balance(g, C.byref(cam), 0, 0, 0, 0.0) == rtk.abi.RTK_ERR_INVALID
    assert lib.rtk_abi_version() == rtk.abi.RTK_ABI_VERSION == 3
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
