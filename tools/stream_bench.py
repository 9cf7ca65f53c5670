"""Wall time per frame of frames queued back to back on one stream (what bench.py's value is made of), for the
library in RTK_LIB (default: the tree's):  python tools/stream_bench.py [config] [frames]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
rtk = importlib.import_module("raytracer-rust_amd")
device = importlib.import_module("raytracer-rust_amd.device")
if os.environ.get("RTK_LIB"):
    device.LIB_PATH = os.path.abspath(os.environ["RTK_LIB"])
name = sys.argv[1] if len(sys.argv) > 1 else "C4"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dev = device.Device(0)
scene, camera = rtk.synthetic.config(name)
handle = dev.upload(scene)
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    out = dev.render_device(handle, camera)
    for _ in range(20):
        out = dev.render_device(handle, camera, out_rgb=out[0], out_rgba8=out[1])
    best = None
    for rep in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(frames):
            dev.render_device(handle, camera, out_rgb=out[0], out_rgba8=out[1])
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / frames
        best = ms if best is None else min(best, ms)
kernel_ms, _ = dev.timing_history(50)
print(f"{name}: {best:.4f} ms per frame queued back to back; kernel {sum(kernel_ms) / len(kernel_ms):.4f} ms")
