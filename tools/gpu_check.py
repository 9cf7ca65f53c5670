"""Quick GPU sanity run: bundled scene vs the golden image, synthetic scenes vs the oracle."""
import importlib, os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import numpy as np
rtk = importlib.import_module("raytracer-rust_amd")
device = importlib.import_module("raytracer-rust_amd.device")
import oracle_binding as ob

dev = device.Device(0)

def compare(name, scene, camera, rect=None):
    t = time.time(); canvas = dev.render(scene, camera); tg = time.time() - t
    st = canvas.stats
    t = time.time(); rgb, rgba, ost = ob.render(scene, camera, rect=rect, threads=16); tc = time.time() - t
    g_rgb, g_rgba = canvas.rgb, canvas.rgba8
    if rect:
        x0, y0, x1, y1 = rect
        g_rgb, g_rgba = g_rgb[y0:y1, x0:x1], g_rgba[y0:y1, x0:x1]
    d8 = np.abs(g_rgba.astype(int) - rgba.astype(int)).max(axis=2)
    rel = np.abs(g_rgb.astype(np.float64) - rgb) / np.maximum(np.abs(rgb), 1e-3)
    print(f"{name}: gpu {tg*1e3:.1f} ms (kernel {st.kernel_ms:.2f} ms) cpu {tc:.1f}s | 8bit exact-mismatch {int((d8>0).sum())} "
          f">1LSB {int((d8>1).sum())} of {d8.size} | max rel {rel.max():.2e} px rel>1e-4: {int((rel.max(axis=2)>1e-4).sum())}")
    print(f"   rays gpu p/s/2 {st.rays_primary}/{st.rays_shadow}/{st.rays_secondary} exact {st.exact_tests} | oracle {ost.rays_primary}/{ost.rays_shadow}/{ost.rays_secondary}"
          + ("" if rect else f" | equal: {(st.rays_primary, st.rays_shadow, st.rays_secondary) == (ost.rays_primary, ost.rays_shadow, ost.rays_secondary)}"))
    rays = st.rays
    print(f"   {rays / st.kernel_ms * 1e-3:.1f} Mrays/s")
    return canvas

scene = rtk.PackedScene.from_yaml_file(os.path.join(ROOT, "tests/golden/test01.yaml")).build()
cam = rtk.Camera(1920, 1080, rtk.PI / 3)
c = compare("C1 bundled 1080p", scene, cam)
gold = np.load(os.path.join(ROOT, "tests/golden/output_stride4.npy"))
print("   vs output.png stride4 mismatches:", int((np.abs(c.rgba8[::4, ::4, :3].astype(int) - gold.astype(int)).max(axis=2) > 0).sum()))
for name, w, h in [("C2", 480, 270), ("C2g", 480, 270), ("C3", 384, 216)]:
    s, cm = rtk.synthetic.config(name, w, h)
    compare(f"{name} {w}x{h}", s, cm)
if "--big" in sys.argv:
    s, cm = rtk.synthetic.config("C3")
    compare("C3 4K crop", s, cm, rect=(1792, 1200, 2048, 1344))
    s, cm = rtk.synthetic.config("C4")
    compare("C4 4K crop", s, cm, rect=(1792, 1200, 2048, 1344))
