"""The C-ABI library loads without a GPU and exports every symbol include/rtk.h declares."""
import ctypes as C
import importlib
import math
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib(rtk):
    device = importlib.import_module("raytracer-rust_amd.device")
    if not os.path.exists(device.LIB_PATH):
        device.build_library()
    return device.library()


def test_header_and_binding_agree_on_the_symbol_list(rtk, lib):
    header = open(os.path.join(ROOT, "include", "rtk.h")).read()
    body = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(rtk_[a-z0-9_]+)\s*\(", body))
    assert declared == set(rtk.abi.SYMBOLS), (declared ^ set(rtk.abi.SYMBOLS))
    for name in declared:
        assert hasattr(lib, name), f"librtk.so does not export {name}"


def test_struct_sizes_match_the_header_layout(rtk):
    a = rtk.abi
    assert C.sizeof(a.rtk_material) == 8 + 6 * 8 + 7 * 8 + 16 * 8 + 8  # the last 8: inner_pattern_transforms
    assert C.sizeof(a.rtk_object) == 8 + 16 * 8 + 4 * 8 + C.sizeof(a.rtk_material)
    assert C.sizeof(a.rtk_light) == 7 * 8
    assert C.sizeof(a.rtk_scene) == 8 + 8 + 8 + 8 + 24 + 8
    assert C.sizeof(a.rtk_camera) == 8 + 4 * 8 + 16 * 8
    assert C.sizeof(a.rtk_stats) == 7 * 8
    assert C.sizeof(a.rtk_tiling) == 8
    assert C.sizeof(a.rtk_banding) == 24


def test_camera_init_equals_host_camera(rtk, lib):  # Camera::new, src/scene/cameras.rs:19-49
    for w, h, fov in [(200, 125, math.pi / 2), (125, 200, math.pi / 2), (1920, 1080, math.pi / 3), (7, 3, 1.0)]:
        c = rtk.abi.rtk_camera()
        assert lib.rtk_camera_init(C.byref(c), w, h, fov) == 0
        host = rtk.Camera(w, h, fov).to_abi()
        for f in ("width", "height", "half_width", "half_height", "field_of_view", "pixel_size"):
            assert getattr(c, f) == getattr(host, f), f
        assert list(c.transform) == list(host.transform)
    assert lib.rtk_camera_init(None, 1, 1, 1.0) == rtk.abi.RTK_ERR_INVALID
    assert lib.rtk_camera_init(C.byref(rtk.abi.rtk_camera()), 0, 1, 1.0) == rtk.abi.RTK_ERR_INVALID
    assert b"rtk_camera_init" in lib.rtk_last_error()


def test_tile_partition_counts(rtk, lib):
    dist = importlib.import_module("raytracer-rust_amd.dist")
    for w, h in [(1920, 1080), (3840, 2160), (33, 9), (1, 1), (100, 37)]:
        for world in (1, 2, 3, 4, 8):
            total = 0
            for rank in range(world):
                n = lib.rtk_tiles_for_rank(w, h, rank, world)
                assert n == dist.tiles_for_rank(w, h, rank, world)
                assert lib.rtk_compact_pixels(w, h, rank, world) == n * 256
                total += n
            tx, ty = dist.tile_grid(w, h)
            assert total == tx * ty
    assert lib.rtk_tiles_for_rank(64, 64, 2, 2) == 0 and lib.rtk_tiles_for_rank(64, 64, 0, 0) == 0


def test_band_partition_counts(rtk, lib):
    """Whole tile rows in bands of m, dealt round-robin: every tile row of the range has one owner."""
    a = rtk.abi
    for h in (1080, 2160, 9, 1, 37, 16384):
        tiles_y = (h + 7) // 8
        for world in (1, 2, 3, 8):
            for m in (1, 2, 5, 64):
                for first, n in ((0, 0), (min(3, tiles_y), 0), (0, min(4, tiles_y))):
                    rows = n if n else tiles_y - first
                    total = 0
                    for rank in range(world):
                        bd = a.rtk_banding(rank, world, m, first, n, 1)
                        got = lib.rtk_band_tile_rows(h, C.byref(bd))
                        want = sum(min(m, rows - b * m) for b in range(rank, (rows + m - 1) // m, world))
                        assert got == want, (h, world, m, first, n, rank)
                        assert lib.rtk_band_pixel_rows(h, C.byref(bd)) == 8 * got
                        total += got
                    assert total == rows
    bad = a.rtk_banding(2, 2, 1, 0, 0, 0)
    assert lib.rtk_band_tile_rows(64, C.byref(bad)) == 0
    assert lib.rtk_band_tile_rows(64, C.byref(a.rtk_banding(0, 1, 0, 0, 0, 0))) == 0   # zero band height
    assert lib.rtk_band_tile_rows(64, C.byref(a.rtk_banding(0, 1, 1, 9, 0, 0))) == 0   # starts below the frame
    assert lib.rtk_band_tile_rows(64, C.byref(a.rtk_banding(0, 1, 1, 4, 5, 0))) == 0   # more rows than the frame has
    assert lib.rtk_band_tile_rows(64, None) == 0


def test_no_gpu_means_an_error_not_a_fallback(rtk, lib):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    ctx = C.c_void_p()
    assert lib.rtk_create(0, C.byref(ctx)) == rtk.abi.RTK_ERR_NO_DEVICE
    assert not ctx.value
    device = importlib.import_module("raytracer-rust_amd.device")
    with pytest.raises(RuntimeError):
        device.Device(0)


def test_product_package_never_touches_the_oracle():
    """The CPU oracle is test infrastructure: nothing of the product, its C++ host, the tools or the
    examples mentions it (tests/, the smoke check and the bench's CPU baseline are its only users)."""
    for sub in ("raytracer-rust_amd", "host", "include", "tools", "examples"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, sub)):
            for f in files:
                if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp", ".sh")) or f == "Makefile":
                    text = open(os.path.join(dirpath, f)).read()
                    assert "oracle" not in text.lower(), os.path.join(dirpath, f)


def test_camera_supersample(rtk, lib):
    """The sample camera: same view, s times the resolution, pixel_size / s (no GPU needed)."""
    cam = rtk.Camera(200, 125, math.pi / 2)
    cam.transform = rtk.Matrix4x4.look_at(rtk.point(1, 2, -3), rtk.point(0, 0, 0), rtk.vec3(0, 1, 0))
    src = cam.to_abi()
    for s in (1, 2, 3, 8, 16):
        out = rtk.abi.rtk_camera()
        assert lib.rtk_camera_supersample(C.byref(src), s, C.byref(out)) == 0
        host = cam.supersampled(s).to_abi()
        for f in ("width", "height", "half_width", "half_height", "field_of_view", "pixel_size"):
            assert getattr(out, f) == getattr(host, f), (s, f)
        assert (out.width, out.height, out.pixel_size) == (200 * s, 125 * s, src.pixel_size / s)
        assert list(out.transform) == list(src.transform)
    out = rtk.abi.rtk_camera()
    for bad in (0, 17):
        assert lib.rtk_camera_supersample(C.byref(src), bad, C.byref(out)) == rtk.abi.RTK_ERR_INVALID
        with pytest.raises(ValueError):
            cam.supersampled(bad)
    assert lib.rtk_camera_supersample(None, 2, C.byref(out)) == rtk.abi.RTK_ERR_INVALID
    big = rtk.Camera(1 << 30, 8, 1.0).to_abi()
    assert lib.rtk_camera_supersample(C.byref(big), 4, C.byref(out)) == rtk.abi.RTK_ERR_UNSUPPORTED
