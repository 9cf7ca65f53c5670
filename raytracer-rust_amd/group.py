"""One frame over several GPUs from ONE process: binding of the rtk_group_* entry points.

Member r renders the bands of whole tile rows dealt to r; members other than the first send their
bands with RCCL and the first (the root) receives each band straight into its final rows of the
frame (csrc/rtk_group.hip).  Pixels are independent (src/scene/cameras.rs:75-82), so the frame is
bit-identical to the single-GPU frame.  A device may be named more than once: such members exchange
through RCCL's send-to-self, which runs every step of the multi-GPU frame on a one-GPU machine.
"""
import ctypes as C

from . import abi
from . import device as _device


class DeviceGroup:
    def __init__(self, device_ids):
        import torch

        self.torch = torch
        ids = [int(d) for d in device_ids]
        if not ids:
            raise ValueError("a group needs at least one device")
        self.device_ids = ids
        self.tdev = torch.device("cuda", ids[0])  # the root: frames land here
        self.side_stream = None                   # see device._OnStream
        arr = (C.c_int * len(ids))(*ids)
        g = C.c_void_p()
        _device._check(_device.library().rtk_group_create(arr, len(ids), C.byref(g)))
        self.group = g
        self._scene = None

    def close(self):
        if self.group:
            _device.library().rtk_group_destroy(self.group)
            self.group = None

    def __len__(self):
        return int(_device.library().rtk_group_size(self.group))

    def upload(self, scene):
        s = scene.to_abi() if hasattr(scene, "to_abi") else scene
        _device._check(_device.library().rtk_group_upload(self.group, C.byref(s)))
        self._scene = s

    def band_tile_rows(self, height):
        return int(_device.library().rtk_group_band_tile_rows(self.group, height))

    def exchange(self):
        """"rccl", "peer copies" or "none"."""
        return _device.library().rtk_group_exchange(self.group).decode()

    def set_bands_per_member(self, bands):
        """About how many bands of a frame each member renders; 0 for the library's choice."""
        _device._check(_device.library().rtk_group_set_bands_per_member(self.group, int(bands)))

    def set_root_share(self, root_share):
        """The root holds `root_share` of every `members - 1 + root_share` bands (its rows do not travel)."""
        _device._check(_device.library().rtk_group_set_root_share(self.group, int(root_share)))

    def set_rows(self, height, first_tile_rows):
        """One run of tile rows per member: member r renders [first_tile_rows[r], first_tile_rows[r + 1]);
        None goes back to bands."""
        if first_tile_rows is None:
            _device._check(_device.library().rtk_group_set_rows(self.group, int(height), None))
            return
        rows = [int(v) for v in first_tile_rows]
        if len(rows) != len(self) + 1:
            raise ValueError("set_rows needs members + 1 boundaries")
        arr = (C.c_uint32 * len(rows))(*rows)
        _device._check(_device.library().rtk_group_set_rows(self.group, int(height), arr))

    def balance(self, camera, want_rgb=True, want_rgba8=True, rounds=0, link_gbps=0.0):
        """Lays one run of rows per member by measurement (rtk_group_balance); synchronises."""
        cam = camera.to_abi() if hasattr(camera, "to_abi") else camera
        _device._check(_device.library().rtk_group_balance(self.group, C.byref(cam), int(bool(want_rgb)), int(bool(want_rgba8)),
                                                           int(rounds), float(link_gbps)))

    def info(self):
        """The last frame in numbers: messages, bytes into the root, each member's rows, kernel time and rays."""
        info = abi.rtk_group_info()
        _device._check(_device.library().rtk_group_get_info(self.group, C.byref(info)))
        n = min(info.n_members, abi.RTK_GROUP_INFO_MEMBERS)
        return {
            "members": int(info.n_members),
            "exchange": ("none", "rccl", "peer copies")[info.exchange],
            "root_share": int(info.root_share),
            "partition": "rows" if info.rows_mode else "bands",
            "messages": int(info.messages),
            "bytes_into_root": int(info.bytes_into_root),
            "gather_ms": float(info.gather_ms),
            "share_tile_rows": [int(info.share_tile_rows[r]) for r in range(n)],
            "share_ms": [float(info.share_ms[r]) for r in range(n)],
            "share_rays": [int(info.share_rays[r]) for r in range(n)],
        }

    def render_device(self, camera, want_rgb=True, want_rgba8=True, out_rgb=None, out_rgba8=None):
        """Queues one frame; returns ([H, W, 3] float32, [H, W, 4] uint8) tensors on the root's device.
        Asynchronous on torch's current stream of that device."""
        torch = self.torch
        cam = camera.to_abi() if hasattr(camera, "to_abi") else camera
        rgb = rgba = None
        if want_rgb:
            rgb = out_rgb if out_rgb is not None else torch.empty((cam.height, cam.width, 3), dtype=torch.float32, device=self.tdev)
        if want_rgba8:
            rgba = out_rgba8 if out_rgba8 is not None else torch.empty((cam.height, cam.width, 4), dtype=torch.uint8, device=self.tdev)
        with _device._OnStream(self) as stream:
            _device._check(_device.library().rtk_group_render_device(
                self.group, C.byref(cam),
                C.c_void_p(rgb.data_ptr()) if rgb is not None else None,
                C.c_void_p(rgba.data_ptr()) if rgba is not None else None,
                stream))
        return rgb, rgba

    def stats(self):
        st = abi.rtk_stats()
        _device._check(_device.library().rtk_group_get_stats(self.group, C.byref(st)))
        return st
