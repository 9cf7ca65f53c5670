import ctypes as C, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rtk = importlib.import_module("raytracer-rust_amd")
device = importlib.import_module("raytracer-rust_amd.device")
device.LIB_PATH = os.path.abspath(os.environ["RTK_LIB"])
dev = device.Device(0)
lib = device.library()
names = ["shadow exact bodies", "bodies where every lane is behind", "bodies where no lane is blocked", "lanes behind AND blocked (must be 0)",
         "lane tests", "lane tests behind", "lane tests blocked", "-"]
for name, size in (("C4", (3840, 2160)), ("C3", (3840, 2160)), ("C2", (1920, 1080))):
    scene, camera = rtk.synthetic.config(name, *size)
    h = dev.upload(scene)
    ev = (C.c_ulonglong * 8)()
    dev.render_device(h, camera)
    import torch; torch.cuda.synchronize()
    lib.rtk_debug_events(ev)
    dev.render_device(h, camera)
    torch.cuda.synchronize()
    st = dev.stats()
    lib.rtk_debug_events(ev)
    waves = (camera.width + 31) // 32 * ((camera.height + 7) // 8) * 4
    print(name, f"{st.kernel_ms:.3f} ms, waves {waves}, exact tests per wave (lanes) {st.exact_tests / waves:.1f}")
    for i in range(7):
        print(f"   {names[i]:40s} {ev[i] / waves:8.2f} per wave")
    h.release()
