"""Wall time of the host-buffer entry point (rtk_render: upload, render, copy back) per frame."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rtk = importlib.import_module("raytracer-rust_amd")
device = importlib.import_module("raytracer-rust_amd.device")
dev = device.Device(0)
for name in sys.argv[1:] or ["C1", "C4"]:
    if name == "C1":
        scene = rtk.PackedScene.from_yaml_file(os.path.join(ROOT, "tests/golden/test01.yaml")).build()
        camera = rtk.Camera(1920, 1080, rtk.PI / 3)
    else:
        scene, camera = rtk.synthetic.config(name)
    s = scene.to_abi()
    times = []
    for _ in range(6):
        t0 = time.perf_counter()
        canvas = dev.render(s, camera)
        times.append(time.perf_counter() - t0)
    rays = canvas.stats.rays
    best = min(times[1:])
    print(f"{name}: {camera.width}x{camera.height} host call {best * 1e3:.2f} ms (first {times[0] * 1e3:.1f} ms), kernels {canvas.stats.kernel_ms:.3f} ms, "
          f"{rays / best / 1e6:.0f} Mrays/s through host buffers", flush=True)
