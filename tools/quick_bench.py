"""Kernel-time-only micro bench: best-of-N HIP-event time of the render kernel per config."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rtk = importlib.import_module("raytracer-rust_amd")
device = importlib.import_module("raytracer-rust_amd.device")
if os.environ.get("RTK_LIB"):  # A/B runs: another build of the library
    device.LIB_PATH = os.path.abspath(os.environ["RTK_LIB"])
dev = device.Device(0)
if os.environ.get("RTK_TRACE_ORDER"):  # 0 auto, 1 depth first, 2 by level
    dev.set_trace_order(int(os.environ["RTK_TRACE_ORDER"]), int(os.environ.get("RTK_QUEUE_NODES", "0")))
names = sys.argv[1:] or ["C2", "C3", "C4"]
for name in names:
    size = (1920, 1080)
    if "@" in name:  # e.g. C4g@3840x2160
        name, res = name.split("@")
        size = tuple(int(v) for v in res.split("x"))
    if name == "C1":
        scene = rtk.PackedScene.from_yaml_file(os.path.join(ROOT, "tests/golden/test01.yaml")).build()
        camera = rtk.Camera(1920, 1080, rtk.PI / 3)
    elif name in ("C3g", "C4g", "C3gw1", "C4gw1", "C3gt", "C4gt"):  # the glass variant of a big configuration at 1080p: secondary rays
        n = 1000 if name.startswith("C3g") else 10000
        scene = rtk.synthetic.make_scene(n, True, 4, 0xC0FFEE00 + n, glass=True)
        if name.endswith("t"):  # ... with refractive indices that differ: the kernels keep the entry before the winner
            for i, node in enumerate(scene.nodes):
                if node.material.transparency > 0:
                    node.material.refractivity = 1.5 if i % 2 else 1.33
        if name.endswith("w1"):  # ... with lights built from points
            scene.lights = [rtk.PointLight(rtk.point(*l.position[:3]), l.intensity) for l in scene.lights]
        camera = rtk.synthetic.make_camera(n, *size)
    elif name == "C3w1":           # C3 with w = 1 lights: the two-kernel opaque path
        scene, camera = rtk.synthetic.config("C3")
        scene.lights = [rtk.PointLight(rtk.point(*l.position[:3]), l.intensity) for l in scene.lights]
    elif name in ("C4s2", "C4s4"):  # C4 at 4K / 1080p with 2x2 / 4x4 samples per pixel: the averaging kernel
        sp = int(name[-1])
        scene, camera = rtk.synthetic.config("C4", 3840 * 2 // sp, 2160 * 2 // sp)
        h = dev.upload(scene)
        best = None
        for _ in range(6):
            dev.render_device_sampled(h, camera, sp)
            st = dev.stats()
            if best is None or st.kernel_ms < best.kernel_ms:
                best = st
        px = camera.width * camera.height
        ms = best.kernel_ms - best.rays_ms
        print(f"{name}: {best.kernel_ms:8.3f} ms ({camera.width}x{camera.height}, {sp * sp} samples per pixel), of which averaging "
              f"{ms:.3f} ms = {px * (12 * sp * sp + 16) / (ms * 1e-3) / 1e9:.0f} GB/s", flush=True)
        h.release()
        continue
    elif name[0] == "g" and name[1:].isdigit():  # a small glass scene: that many spheres + plane, 2 lights, 1080p
        n = int(name[1:])
        scene = rtk.synthetic.make_scene(n, True, 2, 0xC0FFEE00 + n, glass=True)
        camera = rtk.synthetic.make_camera(n, *size)
    elif name == "C5s":
        scene, camera = rtk.synthetic.config("C5", 4096, 4096)
    else:
        scene, camera = rtk.synthetic.config(name)
    h = dev.upload(scene)
    best = None
    for _ in range(6):
        dev.render_device(h, camera)
        st = dev.stats()
        if best is None or st.kernel_ms < best.kernel_ms:
            best = st
    nsph = sum(1 for n in scene.nodes if n.kind == 0)
    fl = 26 * nsph + 8 * (len(scene.nodes) - nsph)
    print(f"{name}: {best.kernel_ms:8.3f} ms  {best.rays / best.kernel_ms / 1e3:9.1f} Mrays/s  "
          f"{100 * best.rays * fl / (best.kernel_ms * 1e-3) / 157.3e12:6.1f}% of FP32 peak (algorithmic)  exact/ray-test {best.exact_tests / (best.rays * max(nsph,1)):.2e}  rays {best.rays_primary}+{best.rays_secondary}+{best.rays_shadow}  exact_tests {best.exact_tests}", flush=True)
    h.release()
