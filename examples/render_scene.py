"""The reference's `main` (src/main.rs:12-24) with the GPU renderer: load a scene file, render it
with Camera::new(1920, 1080, PI / 3), save the PNG.

    python examples/render_scene.py [scene.yaml] [output.png] [samples_per_axis]
"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rtk = importlib.import_module("raytracer-rust_amd")


def main():
    scene_file = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "test01.yaml")
    output = sys.argv[2] if len(sys.argv) > 2 else "./output.png"
    samples = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    scene = rtk.PackedScene.from_yaml_file(scene_file).build()    # loader.rs:179-199
    camera = rtk.Camera(1920, 1080, rtk.PI / 3.0)                 # main.rs:18
    canvas = camera.render(scene, samples_per_axis=samples)       # cameras.rs:72-85, on the GPU
    canvas.save_to_png(output)                                    # graphics.rs:77-81
    st = canvas.stats
    print(f"{scene_file}: {st.rays} rays ({st.rays_primary} camera, {st.rays_shadow} shadow, {st.rays_secondary} secondary) "
          f"in {st.kernel_ms:.3f} ms of kernels -> {output}")


if __name__ == "__main__":
    main()
