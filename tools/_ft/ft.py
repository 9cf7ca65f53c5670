"""Temporary: where the time of a scene upload goes (instrumented library variant)."""
import ctypes as C, importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
rtk = importlib.import_module("raytracer-rust_amd")
device = importlib.import_module("raytracer-rust_amd.device")
dev = device.Device(0)
lib = device.library()
scene, camera = rtk.synthetic.config("C4")
s = scene.to_abi()
for k in range(4):
    if k == 3: os.environ["FT"] = "1"
    h = C.c_void_p()
    t0 = time.perf_counter()
    device._check(lib.rtk_scene_upload(dev.ctx, C.byref(s), C.byref(h)))
    t1 = time.perf_counter()
    print(f"rtk_scene_upload {1e3 * (t1 - t0):.3f} ms", flush=True)
    lib.rtk_scene_release(dev.ctx, h)
