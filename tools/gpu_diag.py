import importlib, os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import numpy as np
rtk = importlib.import_module("raytracer-rust_amd")
device = importlib.import_module("raytracer-rust_amd.device")
import oracle_binding as ob
dev = device.Device(0)
M = rtk.Matrix4x4
mat = rtk.Material.default()
def build(skip=(), lights=(0,1)):
    scene = rtk.Scene()
    L = [rtk.PointLight(rtk.vec3(-10, 10, -10), rtk.rgb(0.7, 0.7, 0.7)), rtk.PointLight(rtk.point(6, 8, -6), rtk.rgb(0.5, 0.4, 0.3))]
    for i in lights: scene.add_light(L[i])
    objs = [
      rtk.Plane().with_material(mat.with_pattern(rtk.Texture.ring(rtk.WHITE, rtk.rgb(0.1, 0.3, 0.5)))),
      rtk.Plane().with_material(mat.with_pattern(rtk.Texture.striped(rtk.rgb(1, 0.5, 0), rtk.BLACK))).with_transform(M.translate(0, 0, 12) * M.rotate_x(math.pi / 2)),
      rtk.Sphere().with_material(mat.with_pattern(rtk.Texture.gradient(rtk.RED, rtk.rgb(0, 0, 1)))).with_transform(M.translate(-2, 1, 1) * M.rotate_z(0.6) * M.scale(1.6, 0.5, 1.0)),
      rtk.Sphere().with_material(mat.with_pattern(rtk.Texture.checker(rtk.WHITE, rtk.rgb(0.2, 0.6, 0.2)))).with_transform(M.translate(2, 1.2, 0.5) * M.rotate_y(0.4) * M.rotate_x(-0.7) * M.scale(0.6, 1.2, 0.9)),
      rtk.Sphere().with_material(mat.with_color(rtk.rgb(0.9, 0.8, 0.2)).with_reflective(0.4)).with_transform(M.translate(0, 0.8, -1.0) * M.shear(0.4, 0, 0, 0.3, 0, 0) * M.scale(0.8, 0.8, 0.8)),
      rtk.Sphere().with_material(mat.with_color(rtk.rgb(0.3, 0.3, 0.9))).with_transform(M.scale(1, 0, 1)),
    ]
    for i, o in enumerate(objs):
        if i not in skip: scene.add_object(o)
    return scene
camera = rtk.Camera(640, 360, rtk.PI / 3.0)
def run(tag, scene):
    c = dev.render(scene, camera)
    rgb, rgba, ost = ob.render(scene, camera, threads=16)
    d8 = np.abs(c.rgba8.astype(int) - rgba.astype(int)).max(axis=2)
    ys, xs = np.nonzero(d8)
    print(f"{tag}: shadow rays gpu {c.stats.rays_shadow} oracle {ost.rays_shadow} | secondary {c.stats.rays_secondary}/{ost.rays_secondary} | differing pixels {len(xs)}",
          (f"bbox x[{xs.min()},{xs.max()}] y[{ys.min()},{ys.max()}]" if len(xs) else ""))
    return c, rgb
run("full", build())
for skip in range(6):
    run(f"without object {skip}", build(skip=(skip,)))
run("light0 only", build(lights=(0,)))
run("light1 only", build(lights=(1,)))
