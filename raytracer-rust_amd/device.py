"""Binding of the HIP device library (csrc/librtk.so) and the device-side render calls.

There is no fallback: if the library is missing or no gfx950 device is present, the
calls raise.  torch is used only to own device memory and the stream.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from . import abi
from .canvas import Canvas

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "librtk.so")


class RtkError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"rtk error {code}: {message}")
        self.code = code


def build_library(force=False):
    """Compiles csrc/librtk.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "librtk.so"]
    if force:
        cmd.insert(1, "-B")
    subprocess.run(cmd, check=True, capture_output=True)
    return LIB_PATH


_lib = None

class _OnStream:
    """Queues library calls in order with torch's current stream.  A torch stream created by the
    caller is handed over as it is.  torch reports its DEFAULT stream as handle 0, which the C ABI reads
    as "the context's own stream" -- work queued there would not be ordered with what torch runs before
    and after it -- so calls made under the default stream go to a side stream that first waits for the
    default stream and that the default stream then waits for."""

    def __init__(self, device):
        self.device, self.current, self.side = device, None, None

    def __enter__(self):
        torch = self.device.torch
        self.current = torch.cuda.current_stream(self.device.tdev)
        if self.current.cuda_stream:
            return C.c_void_p(self.current.cuda_stream)
        if self.device.side_stream is None:
            self.device.side_stream = torch.cuda.Stream(self.device.tdev)
        self.side = self.device.side_stream
        self.side.wait_stream(self.current)
        return C.c_void_p(self.side.cuda_stream)

    def __exit__(self, *exc):
        if self.side is not None:
            self.current.wait_stream(self.side)
        return False


def library():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(the renderer has no CPU fallback)")
        # PyTorch ships a HIP runtime of its own.  Loaded after librtk.so has brought in the system's,
        # it finds no device (torch.cuda.is_available() turns False); loaded first, its runtime serves
        # both.  A host without PyTorch (the C ABI's real caller) has one runtime and no such order.
        import torch  # noqa: F401
        _lib = abi.bind(C.CDLL(LIB_PATH))
    return _lib


def _check(rc):
    if rc != abi.RTK_OK:
        raise RtkError(rc, library().rtk_last_error().decode())


def camera_init(width, height, field_of_view):
    cam = abi.rtk_camera()
    _check(library().rtk_camera_init(C.byref(cam), width, height, field_of_view))
    return cam


def tiles_for_rank(width, height, rank, world):
    return int(library().rtk_tiles_for_rank(width, height, rank, world))


def compact_pixels(width, height, rank, world):
    return int(library().rtk_compact_pixels(width, height, rank, world))


def compact_pixels_sampled(width, height, samples_per_axis, rank, world):
    return int(library().rtk_compact_pixels_sampled(width, height, samples_per_axis, rank, world))


class SceneHandle:
    def __init__(self, device, handle, scene_abi):
        self.device, self.handle, self._keep = device, handle, scene_abi

    def release(self):
        if self.handle:
            library().rtk_scene_release(self.device.ctx, self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


class Device:
    """One rtk_context bound to one HIP device."""

    def __init__(self, index=0):
        import torch

        if not torch.cuda.is_available():
            raise RuntimeError("no HIP device visible: the renderer needs an MI355X (there is no CPU fallback)")
        self.torch = torch
        self.index = index
        self.tdev = torch.device("cuda", index)
        self.side_stream = None  # see _OnStream
        ctx = C.c_void_p()
        _check(library().rtk_create(index, C.byref(ctx)))
        self.ctx = ctx

    def close(self):
        if self.ctx:
            library().rtk_destroy(self.ctx)
            self.ctx = None

    # ---- scenes -----------------------------------------------------------------
    def upload(self, scene):
        s = scene.to_abi() if hasattr(scene, "to_abi") else scene
        h = C.c_void_p()
        _check(library().rtk_scene_upload(self.ctx, C.byref(s), C.byref(h)))
        return SceneHandle(self, h, s)

    # ---- rendering ----------------------------------------------------------------
    def render_device(self, handle, camera, rank=0, world=1, want_rgb=True, want_rgba8=True, out_rgb=None,
                      out_rgba8=None):
        """Asynchronous render into torch tensors on this device (current torch stream).
        world == 1: [H, W, 3] float32 / [H, W, 4] uint8; world > 1: compact [pixels, 3] / [pixels, 4]."""
        torch = self.torch
        cam = camera.to_abi() if hasattr(camera, "to_abi") else camera
        if world == 1:
            shape3, shape4 = (cam.height, cam.width, 3), (cam.height, cam.width, 4)
        else:
            n = compact_pixels(cam.width, cam.height, rank, world)
            shape3, shape4 = (n, 3), (n, 4)
        rgb = rgba = None
        if want_rgb:
            rgb = out_rgb if out_rgb is not None else torch.empty(shape3, dtype=torch.float32, device=self.tdev)
        if want_rgba8:
            rgba = out_rgba8 if out_rgba8 is not None else torch.empty(shape4, dtype=torch.uint8, device=self.tdev)
        tiling = abi.rtk_tiling(rank, world)
        with _OnStream(self) as stream:
            _check(library().rtk_render_device(
                self.ctx, handle.handle, C.byref(cam), C.byref(tiling),
                C.c_void_p(rgb.data_ptr()) if rgb is not None else None,
                C.c_void_p(rgba.data_ptr()) if rgba is not None else None,
                stream))
        return rgb, rgba

    def set_trace_order(self, order, max_queue_nodes=0):
        """How scenes with secondary rays are traced: abi.RTK_TRACE_AUTO / _DEPTH_FIRST / _BY_LEVEL, and
        (when not 0) the capacity of the node queue of the level-by-level path."""
        _check(library().rtk_set_trace_order(self.ctx, int(order), int(max_queue_nodes)))

    def render_device_bands(self, handle, camera, rank=0, world=1, band_tile_rows=1, first_tile_row=0, n_tile_rows=0,
                            compact=False, want_rgb=True, want_rgba8=True, out_rgb=None, out_rgba8=None):
        """One rank of a band partition (whole tile rows dealt round-robin in bands of `band_tile_rows`).
        compact=False: [H, W, 3] / [H, W, 4] frames of which only this rank's rows are written (pass the
        frame every rank writes into as out_rgb / out_rgba8); compact=True: the rank's own rows only,
        [rtk_band_pixel_rows, W, 3] / [.., W, 4]."""
        torch = self.torch
        cam = camera.to_abi() if hasattr(camera, "to_abi") else camera
        bd = abi.rtk_banding(rank, world, band_tile_rows, first_tile_row, n_tile_rows, 1 if compact else 0)
        rows = int(library().rtk_band_pixel_rows(cam.height, C.byref(bd))) if compact else cam.height
        rgb = rgba = None
        if want_rgb:
            rgb = out_rgb if out_rgb is not None else torch.zeros((rows, cam.width, 3), dtype=torch.float32, device=self.tdev)
        if want_rgba8:
            rgba = out_rgba8 if out_rgba8 is not None else torch.zeros((rows, cam.width, 4), dtype=torch.uint8, device=self.tdev)
        with _OnStream(self) as stream:
            _check(library().rtk_render_device_bands(
                self.ctx, handle.handle, C.byref(cam), C.byref(bd),
                C.c_void_p(rgb.data_ptr()) if rgb is not None else None,
                C.c_void_p(rgba.data_ptr()) if rgba is not None else None,
                stream))
        return rgb, rgba

    def render_device_sampled(self, handle, camera, samples_per_axis, rank=0, world=1, want_rgb=True, want_rgba8=True,
                              out_rgb=None, out_rgba8=None):
        """render_device with samples_per_axis squared samples per pixel (the reference has one
        sample per pixel, src/scene/cameras.rs:53-54).  world > 1: compact buffers of
        compact_pixels_sampled() pixels, samples_per_axis 1, 2, 4 or 8."""
        torch = self.torch
        cam = camera.to_abi() if hasattr(camera, "to_abi") else camera
        s = int(samples_per_axis)
        if world == 1:
            shape3, shape4 = (cam.height, cam.width, 3), (cam.height, cam.width, 4)
        else:
            n = compact_pixels_sampled(cam.width, cam.height, s, rank, world)
            shape3, shape4 = (n, 3), (n, 4)
        rgb = rgba = None
        if want_rgb:
            rgb = out_rgb if out_rgb is not None else torch.empty(shape3, dtype=torch.float32, device=self.tdev)
        if want_rgba8:
            rgba = out_rgba8 if out_rgba8 is not None else torch.empty(shape4, dtype=torch.uint8, device=self.tdev)
        tiling = abi.rtk_tiling(rank, world)
        with _OnStream(self) as stream:
            _check(library().rtk_render_device_sampled(
                self.ctx, handle.handle, C.byref(cam), s, C.byref(tiling),
                C.c_void_p(rgb.data_ptr()) if rgb is not None else None,
                C.c_void_p(rgba.data_ptr()) if rgba is not None else None,
                stream))
        return rgb, rgba

    def resolve(self, samples, samples_per_axis):
        """Averages a [s*H, s*W, 3] float32 sample frame on this device into ([H, W, 3] float32, [H, W, 4] uint8)."""
        torch = self.torch
        s = int(samples_per_axis)
        if samples.dim() != 3 or samples.shape[2] != 3 or samples.dtype != torch.float32 or not samples.is_contiguous():
            raise ValueError("samples must be a contiguous [s*H, s*W, 3] float32 tensor")
        if s < 1 or samples.shape[0] % s or samples.shape[1] % s:
            raise ValueError("sample frame is not a multiple of samples_per_axis")
        h, w = samples.shape[0] // s, samples.shape[1] // s
        rgb = torch.empty((h, w, 3), dtype=torch.float32, device=self.tdev)
        rgba = torch.empty((h, w, 4), dtype=torch.uint8, device=self.tdev)
        with _OnStream(self) as stream:
            _check(library().rtk_resolve(self.ctx, w, h, s, C.c_void_p(samples.data_ptr()), C.c_void_p(rgb.data_ptr()),
                                         C.c_void_p(rgba.data_ptr()), stream))
        return rgb, rgba

    def stats(self):
        st = abi.rtk_stats()
        _check(library().rtk_get_stats(self.ctx, C.byref(st)))
        return st

    def timing_history(self, n=abi.RTK_TIMING_HISTORY):
        """(kernel_ms, rays_ms) lists of the last `n` device renders (at most RTK_TIMING_HISTORY), oldest first."""
        n = max(1, min(int(n), abi.RTK_TIMING_HISTORY))
        k, r = (C.c_double * n)(), (C.c_double * n)()
        got = library().rtk_get_timing_history(self.ctx, n, k, r)
        if got < 0:
            _check(got)
        return list(k[:got]), list(r[:got])

    def render(self, scene, camera, samples_per_axis=1):
        """`Camera::render` through the host-buffer entry point; returns a Canvas."""
        s = scene.to_abi() if hasattr(scene, "to_abi") else scene
        cam = camera.to_abi() if hasattr(camera, "to_abi") else camera
        # page-locked host buffers: the copy back runs at PCIe speed and the pages are not faulted in per frame
        torch = self.torch
        rgb = torch.empty((cam.height, cam.width, 3), dtype=torch.float32, pin_memory=True).numpy()
        rgba = torch.empty((cam.height, cam.width, 4), dtype=torch.uint8, pin_memory=True).numpy()
        st = abi.rtk_stats()
        _check(library().rtk_render_sampled(self.ctx, C.byref(s), C.byref(cam), int(samples_per_axis),
                                            rgb.ctypes.data_as(C.c_void_p), rgba.ctypes.data_as(C.c_void_p), C.byref(st)))
        return Canvas(cam.width, cam.height, rgb, rgba, st)

    def trace(self, handle, origins, directions):
        """`Scene::trace` for rays given as [n, 4] float64 arrays; returns [n, 3] float64."""
        o = np.ascontiguousarray(origins, dtype=np.float64).reshape(-1, 4)
        d = np.ascontiguousarray(directions, dtype=np.float64).reshape(-1, 4)
        out = np.empty((o.shape[0], 3), dtype=np.float64)
        _check(library().rtk_trace_rays(self.ctx, handle.handle, o.ctypes.data_as(C.c_void_p),
                                        d.ctypes.data_as(C.c_void_p), o.shape[0], out.ctypes.data_as(C.c_void_p)))
        return out

    def untile(self, width, height, world, slot_pixels, rgb_compact=None, rgba8_compact=None, out_rgb=None,
               out_rgba8=None, rank_stride_bytes=None, samples_per_axis=1):
        """Compact per-rank tile buffers -> row-major frames (device tensors).  By default the ranks'
        buffers lie end to end, `slot_pixels` each; with `rank_stride_bytes` every rank's FP32 and
        RGBA8 buffers start that many bytes after the previous rank's (one gathered message per rank)."""
        torch = self.torch
        rgb = rgba = None
        if rgb_compact is not None:
            rgb = out_rgb if out_rgb is not None else torch.empty((height, width, 3), dtype=torch.float32, device=self.tdev)
        if rgba8_compact is not None:
            rgba = out_rgba8 if out_rgba8 is not None else torch.empty((height, width, 4), dtype=torch.uint8, device=self.tdev)
        ptrs = (C.c_void_p(rgb_compact.data_ptr()) if rgb_compact is not None else None,
                C.c_void_p(rgb.data_ptr()) if rgb is not None else None,
                C.c_void_p(rgba8_compact.data_ptr()) if rgba8_compact is not None else None,
                C.c_void_p(rgba.data_ptr()) if rgba is not None else None)
        with _OnStream(self) as stream:
            if samples_per_axis != 1:  # buffers of sampled, tiled renders: (32/s) x (8/s) tiles
                rgb_stride = rank_stride_bytes if rank_stride_bytes is not None else slot_pixels * 12
                px_stride = rank_stride_bytes if rank_stride_bytes is not None else slot_pixels * 4
                _check(library().rtk_untile_sampled(self.ctx, width, height, int(samples_per_axis), world, rgb_stride,
                                                    px_stride, *ptrs, stream))
            elif rank_stride_bytes is None:
                _check(library().rtk_untile(self.ctx, width, height, world, slot_pixels, *ptrs, stream))
            else:
                _check(library().rtk_untile_strided(self.ctx, width, height, world, rank_stride_bytes, rank_stride_bytes,
                                                    *ptrs, stream))
        return rgb, rgba


_default = None


def default_device():
    global _default
    if _default is None:
        _default = Device(int(os.environ.get("LOCAL_RANK", "0")))
    return _default
