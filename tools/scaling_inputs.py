"""Single-GPU measurements behind the multi-GPU scaling estimate: what one member's share of the frame
costs (bands and tiles), and what the root spends putting gathered tiles in place (the de-interleave
kernel the band partition does without)."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
rtk = importlib.import_module("raytracer-rust_amd")
device = importlib.import_module("raytracer-rust_amd.device")
dist = importlib.import_module("raytracer-rust_amd.dist")
dev = device.Device(0)
stream = torch.cuda.Stream()


def timed(fn, reps=8):
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


for name, worlds in (("C4", (1, 2, 4, 8)), ("C5", (1, 8))):
    scene, camera = rtk.synthetic.config(name)
    w, h = camera.width, camera.height
    want_rgb = name != "C5" or True
    handle = dev.upload(scene)
    with torch.cuda.stream(stream):
        f_rgb = torch.empty((h, w, 3), dtype=torch.float32, device=dev.tdev)
        f_rgba = torch.empty((h, w, 4), dtype=torch.uint8, device=dev.tdev)
    for world in worlds:
        tiles_y = (h + 7) // 8
        m = max(1, tiles_y // (world * 4))
        worst_band = worst_tile = 0.0
        for rank in range(world):
            tb = timed(lambda: dev.render_device_bands(handle, camera, rank, world, m, out_rgb=f_rgb, out_rgba8=f_rgba), 4)
            worst_band = max(worst_band, tb)
            if world > 1:
                n = dist.tiles_for_rank(w, h, rank, world) * 256
                with torch.cuda.stream(stream):
                    c_rgb = torch.empty((n, 3), dtype=torch.float32, device=dev.tdev)
                    c_rgba = torch.empty((n, 4), dtype=torch.uint8, device=dev.tdev)
                tt = timed(lambda: dev.render_device(handle, camera, rank=rank, world=world, out_rgb=c_rgb, out_rgba8=c_rgba), 4)
                worst_tile = max(worst_tile, tt)
                del c_rgb, c_rgba
        line = f"{name} {w}x{h} world {world}: slowest member's share by bands {worst_band:.3f} ms"
        if world > 1:
            slot = dist.slot_pixels(w, h, world)
            with torch.cuda.stream(stream):
                g_rgb = torch.zeros((world * slot, 3), dtype=torch.float32, device=dev.tdev)
                g_rgba = torch.zeros((world * slot, 4), dtype=torch.uint8, device=dev.tdev)
            t_both = timed(lambda: dev.untile(w, h, world, slot, g_rgb, g_rgba, out_rgb=f_rgb, out_rgba8=f_rgba))
            t_px = timed(lambda: dev.untile(w, h, world, slot, None, g_rgba, out_rgba8=f_rgba))
            line += f", by tiles {worst_tile:.3f} ms; de-interleave on the root (tiles only) FP32+RGBA8 {t_both:.3f} ms, RGBA8 only {t_px:.3f} ms"
            del g_rgb, g_rgba
        print(line, flush=True)
    handle.release()
