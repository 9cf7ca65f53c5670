"""Kernel time of every frame of a back-to-back run (HIP events of the library), to see how long the GPU takes to settle."""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

rtk = importlib.import_module("raytracer-rust_amd")
device = importlib.import_module("raytracer-rust_amd.device")
rdist = importlib.import_module("raytracer-rust_amd.dist")

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
torch.cuda.set_device(0)
torch.cuda.set_stream(torch.cuda.Stream())
scene, camera = rtk.synthetic.config("C4")
dev = device.Device(0)
handle = dev.upload(scene)
r = rdist.gpu_renderer(dev, handle, camera, group=None, want_rgb=True)
done = 0
while done < n:  # the library keeps the times of the last 64 frames: read them chunk by chunk (a sync costs microseconds)
    m = min(50, n - done)
    for _ in r.stream(m):
        pass
    torch.cuda.synchronize()
    kernel_ms, rays_ms = dev.timing_history(m)
    for k in range(0, m, 10):
        chunk = rays_ms[k:k + 10]
        print(f"frames {done + k:4d}-{done + k + len(chunk) - 1:4d}: {sum(chunk) / len(chunk):.4f} ms  (min {min(chunk):.4f}, max {max(chunk):.4f})")
    done += m
