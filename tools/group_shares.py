"""What the in-library group's partition by measurement gives, member by member, on ONE GPU.

For C4 (3840x2160) over 2, 4 and 8 members and C5 (16384x16384) over 8, for both sets of outputs and two
assumed link rates: the group (all members on device 0, RCCL send-to-self) lays its boundaries with
rtk_group_balance; one frame then tells which tile rows each member holds, and each member's run of rows is
rendered ALONE (scene resident, best of several) to get the kernel time it would take on a GPU of its own.
Printed per case: rows and time per member, bytes each peer sends, and the frame time the model
    max(slowest member's kernel, largest peer message / link rate)
predicts with two frames in flight, against the single-GPU frame.  Not a scaling measurement: the links are
assumed, only the shares are measured.

  python tools/group_shares.py [C4|C5 ...]
"""
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

rtk = importlib.import_module("raytracer-rust_amd")
device = importlib.import_module("raytracer-rust_amd.device")
group = importlib.import_module("raytracer-rust_amd.group")
dev = device.Device(0)


def share_alone(handle, camera, first, n, want_rgb, reps=5):
    best = None
    out = (None, None)
    for _ in range(reps):
        out = dev.render_device_bands(handle, camera, 0, 1, max(1, n), first, n, compact=True, want_rgb=want_rgb,
                                      out_rgb=out[0], out_rgba8=out[1])
        ms = dev.stats().kernel_ms
        best = ms if best is None else min(best, ms)
    return best


names = sys.argv[1:] or ["C4", "C5"]
results = []
for name in names:
    scene, camera = rtk.synthetic.config(name)
    w, h = camera.width, camera.height
    handle = dev.upload(scene)
    single = None
    out = (None, None)
    for _ in range(5):
        out = dev.render_device(handle, camera, out_rgb=out[0], out_rgba8=out[1])
        ms = dev.stats().kernel_ms
        single = ms if single is None else min(single, ms)
    del out
    print(f"{name} {w}x{h}: single GPU {single:.3f} ms", flush=True)
    for members in ((2, 4, 8) if name != "C5" else (8,)):
        g = group.DeviceGroup([0] * members)
        g.upload(scene)
        for want_rgb, bpp in ((True, 16), (False, 4)):
            for gbps in (153.0, 100.0):
                g.balance(camera, want_rgb=want_rgb, want_rgba8=True, rounds=5, link_gbps=gbps)
                g.render_device(camera, want_rgb=want_rgb)
                info = g.info()
                torch.cuda.synchronize()
                rows = info["share_tile_rows"]
                first, shares, link = 0, [], []
                for r, n in enumerate(rows):
                    shares.append(share_alone(handle, camera, first, n, want_rgb) if n else 0.0)
                    y0, y1 = first * 8, min(h, (first + n) * 8)
                    link.append(0.0 if r == 0 else (y1 - y0) * w * bpp / (gbps * 1e6))
                    first += n
                frame = max(max(shares), max(link))
                res = {"config": name, "members": members, "outputs": "FP32+RGBA8" if want_rgb else "RGBA8", "link_GBps": gbps,
                       "tile_rows": rows, "share_ms": [round(s, 4) for s in shares], "link_ms": [round(l, 4) for l in link],
                       "messages": info["messages"], "bytes_into_root": info["bytes_into_root"], "frame_ms": round(frame, 4),
                       "single_ms": round(single, 4), "speedup": round(single / frame, 2)}
                results.append(res)
                print(json.dumps(res), flush=True)
        g.close()
    handle.release()
