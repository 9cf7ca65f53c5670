"""Wall time of the host-buffer entry point (rtk_render: flatten + upload, render, copy back) per frame.

For each scene: page-locked and pageable output buffers; the same scene passed again (the library keeps
the last scene of a context on the device) and a scene that changed since the last call (flattened and
uploaded again)."""
import ctypes as C
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rtk = importlib.import_module("raytracer-rust_amd")
device = importlib.import_module("raytracer-rust_amd.device")
abi = rtk.abi
import torch

dev = device.Device(0)
lib = device.library()


def call(s, cam, rgb, rgba):
    st = abi.rtk_stats()
    t0 = time.perf_counter()
    device._check(lib.rtk_render(dev.ctx, C.byref(s), C.byref(cam), rgb.ctypes.data_as(C.c_void_p),
                                 rgba.ctypes.data_as(C.c_void_p), C.byref(st)))
    return time.perf_counter() - t0, st


for name in sys.argv[1:] or ["C1", "C4"]:
    if name == "C1":
        scene = rtk.PackedScene.from_yaml_file(os.path.join(ROOT, "tests/golden/test01.yaml")).build()
        camera = rtk.Camera(1920, 1080, rtk.PI / 3)
    else:
        scene, camera = rtk.synthetic.config(name)
    s, cam = scene.to_abi(), camera.to_abi()
    h, w = cam.height, cam.width
    buffers = {
        "page-locked": (torch.empty((h, w, 3), dtype=torch.float32, pin_memory=True).numpy(),
                        torch.empty((h, w, 4), dtype=torch.uint8, pin_memory=True).numpy()),
        "pageable": (np.zeros((h, w, 3), np.float32), np.zeros((h, w, 4), np.uint8)),
    }
    reference = None
    for kind, (rgb, rgba) in buffers.items():
        call(s, cam, rgb, rgba)
        same = [call(s, cam, rgb, rgba) for _ in range(6)]
        changed = []
        for k in range(6):
            s.objects[0].material.color_a[0] = 0.5 + 0.01 * k  # a different scene every call
            changed.append(call(s, cam, rgb, rgba))
        st = same[-1][1]
        rays = st.rays_primary + st.rays_shadow + st.rays_secondary
        t_same, t_changed = min(t for t, _ in same), min(t for t, _ in changed)
        print(f"{name} {w}x{h} {kind:11s}: same scene {t_same * 1e3:6.2f} ms ({rays / t_same / 1e6:7.0f} Mrays/s), "
              f"changed scene {t_changed * 1e3:6.2f} ms ({rays / t_changed / 1e6:7.0f} Mrays/s), kernels {st.kernel_ms:.3f} ms",
              flush=True)
