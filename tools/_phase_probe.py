import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rtk = importlib.import_module("raytracer-rust_amd")
device = importlib.import_module("raytracer-rust_amd.device")
device.LIB_PATH = os.path.abspath(os.environ["RTK_LIB"])
dev = device.Device(0)
for name, size in (("C4", (3840, 2160)), ("C3", (3840, 2160)), ("C2", (1920, 1080))):
    scene, camera = rtk.synthetic.config(name, *size)
    h = dev.upload(scene)
    for _ in range(3):
        dev.render_device(h, camera)
        st = dev.stats()
    waves = (camera.width + 31) // 32 * ((camera.height + 7) // 8) * 4
    geo, store = st.exact_tests >> 20, st.exact_tests & ((1 << 20) - 1)
    print(f"{name}: {st.kernel_ms:.3f} ms; per wave cycles: primary {st.rays_primary / waves:.0f} geometry {geo / waves:.0f} shadow {st.rays_shadow / waves:.0f} "
          f"shade {st.rays_secondary / waves:.0f} store(low bits only) {store}", flush=True)
    h.release()
