"""How many bands per member?  On ONE GPU: the slowest member's share of the C4 frame for several band
heights (what a member's kernels take on a GPU of its own), and the group's frame through RCCL's
send-to-self (every member on device 0) with its exchange time, which shows what a message costs."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
rtk = importlib.import_module("raytracer-rust_amd")
device = importlib.import_module("raytracer-rust_amd.device")
group = importlib.import_module("raytracer-rust_amd.group")

name = sys.argv[1] if len(sys.argv) > 1 else "C4"
scene, camera = rtk.synthetic.config(name)
w, h = camera.width, camera.height
tiles_y = (h + 7) // 8
dev = device.Device(0)
stream = torch.cuda.Stream()
handle = dev.upload(scene)
with torch.cuda.stream(stream):
    f_rgb = torch.empty((h, w, 3), dtype=torch.float32, device=dev.tdev)
    f_rgba = torch.empty((h, w, 4), dtype=torch.uint8, device=dev.tdev)


def timed(fn, reps=4):
    best = 1e9
    with torch.cuda.stream(stream):
        fn()
        stream.synchronize()
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            fn()
            b.record(stream)
            stream.synchronize()
            best = min(best, a.elapsed_time(b))
    return best


for world in (2, 4, 8):
    g = group.DeviceGroup([0] * world)
    g.upload(scene)
    for per_member in (2, 4, 8, 16, 32):
        m = max(1, tiles_y // (world * per_member))
        shares = [timed(lambda: dev.render_device_bands(handle, camera, r, world, m, out_rgb=f_rgb, out_rgba8=f_rgba)) for r in range(world)]
        g.set_bands_per_member(per_member)
        assert g.band_tile_rows(h) == m
        with torch.cuda.stream(stream):
            for _ in range(3):
                g.render_device(camera, out_rgb=f_rgb, out_rgba8=f_rgba)
            stream.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                g.render_device(camera, out_rgb=f_rgb, out_rgba8=f_rgba)
            stream.synchronize()
            step = (time.perf_counter() - t0) / 10 * 1e3
        st = g.stats()
        n_bands = (tiles_y + m - 1) // m
        print(f"{name} world {world} bands/member {per_member:2d} ({m} tile rows, {n_bands} bands): slowest share {max(shares):.3f} ms, "
              f"mean {sum(shares) / world:.3f}; all members on one GPU {step:.3f} ms/frame, exchange {st.gather_ms:.3f} ms", flush=True)
    g.close()
