"""Multi-GPU frames: tile partition, gather to rank 0, de-interleave.

Pixels are independent (the reference's loop has no cross-iteration state,
src/scene/cameras.rs:75-82), so the frame shards with no data-path collective: rank r of
`world` renders the 32x8 tiles t with t % world == r (round-robin keeps sky and ground
tiles evenly mixed).  The only exchange is ONE gather per frame of each rank's compact tile
buffers (FP32 and RGBA8 in one message) to rank 0 -- over xGMI through RCCL when the process
group's backend is "nccl" -- followed by one de-interleave kernel on rank 0.

One process per GPU; the process group is the caller's (`torch.distributed`).
The render and the de-interleave step are injectable so that the partition and gather
logic can be exercised on CPU with the `gloo` backend.
"""
import numpy as np

from . import abi

TILE_W, TILE_H = abi.RTK_TILE_W, abi.RTK_TILE_H
TILE_PIXELS = TILE_W * TILE_H


def tile_grid(width, height):
    return (width + TILE_W - 1) // TILE_W, (height + TILE_H - 1) // TILE_H


def tiles_for_rank(width, height, rank, world):
    tx, ty = tile_grid(width, height)
    n = tx * ty
    return n // world + (1 if (n % world) > rank else 0)


def slot_pixels(width, height, world):
    """Pixels per gather slot: the largest compact buffer (rank 0 owns the most tiles)."""
    return tiles_for_rank(width, height, 0, world) * TILE_PIXELS


def slot_pixels_sampled(width, height, samples_per_axis, world):
    """The same for sampled renders: the ranks tile the SAMPLE frame, and a 32x8 tile of samples
    becomes a (32/s) x (8/s) tile of pixels (s = 1, 2, 4, 8)."""
    s = int(samples_per_axis)
    if s not in (1, 2, 4, 8):
        raise ValueError("tiled sampled renders need samples_per_axis 1, 2, 4 or 8")
    return tiles_for_rank(width * s, height * s, 0, world) * (TILE_PIXELS // (s * s))


def tile_rect(width, height, tile):
    """Pixel rectangle (x0, y0, x1, y1) of a global tile index, clipped to the frame."""
    tx, _ = tile_grid(width, height)
    x0, y0 = (tile % tx) * TILE_W, (tile // tx) * TILE_H
    return x0, y0, min(x0 + TILE_W, width), min(y0 + TILE_H, height)


def untile_host(width, height, world, slot, compact, channels):
    """Reference de-interleave on the host: `compact` is [world * slot, channels]."""
    compact = np.asarray(compact).reshape(world * slot, channels)
    frame = np.zeros((height, width, channels), dtype=compact.dtype)
    tx, ty = tile_grid(width, height)
    for tile in range(tx * ty):
        rank, k = tile % world, tile // world
        x0, y0, x1, y1 = tile_rect(width, height, tile)
        block = compact[rank * slot + k * TILE_PIXELS: rank * slot + (k + 1) * TILE_PIXELS].reshape(TILE_H, TILE_W, channels)
        frame[y0:y1, x0:x1] = block[: y1 - y0, : x1 - x0]
    return frame


def untile_host_sampled(width, height, samples_per_axis, world, slot, compact, channels):
    """The same for the (32/s) x (8/s) pixel tiles of a sampled, tiled render."""
    s = int(samples_per_axis)
    tw, th = TILE_W // s, TILE_H // s
    per_tile = tw * th
    compact = np.asarray(compact).reshape(world * slot, channels)
    frame = np.zeros((height, width, channels), dtype=compact.dtype)
    tx, ty = tile_grid(width * s, height * s)
    for tile in range(tx * ty):
        rank, k = tile % world, tile // world
        x0, y0 = (tile % tx) * tw, (tile // tx) * th
        x1, y1 = min(x0 + tw, width), min(y0 + th, height)
        if x0 >= width or y0 >= height:
            continue
        block = compact[rank * slot + k * per_tile: rank * slot + (k + 1) * per_tile].reshape(th, tw, channels)
        frame[y0:y1, x0:x1] = block[: y1 - y0, : x1 - x0]
    return frame


class DistributedRenderer:
    """Renders a frame over the ranks of a torch.distributed process group.

    render_compact(rank, world, out_rgb, out_rgba8): fills this rank's buffers (torch
        tensors [slot, 3] float32 and [slot, 4] uint8; either may be None) -- compact
        tile-major layout when world > 1, the row-major frame itself (first W*H pixels)
        when world == 1, exactly as rtk_render_device lays them out.
    untile(message, rgb_view, rgba8_view, stride_bytes): rank-0 only, world > 1; `message` is the
        gathered [world, stride_bytes] uint8 tensor, the views are rank 0's parts of it
        (None for an output that was not asked for) -> (rgb frame, rgba8 frame).

    A rank's FP32 and RGBA8 tiles travel in ONE message (FP32 part first), so a frame costs one
    gather whatever outputs are wanted.

    `frame()` renders, gathers and de-interleaves one frame.  For a stream of frames use
    `submit()` / `collect()`: up to two frames may be in flight, so the gather of frame k (on the
    communication stream of the process group) overlaps rank 0's de-interleave of frame k-1 and
    every rank's render of frame k+1.  Two message buffers alternate; nothing is allocated per frame.
    """

    DEPTH = 2

    def __init__(self, width, height, group=None, want_rgb=True, want_rgba8=True, tensor_device=None,
                 render_compact=None, untile=None, always_gather=False, samples_per_axis=1, root_share=1):
        import collections

        import torch
        import torch.distributed as dist

        self.torch, self.dist = torch, dist
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.width, self.height = width, height
        self.samples_per_axis = int(samples_per_axis)
        self.want_rgb, self.want_rgba8 = want_rgb, want_rgba8
        self.render_compact, self.untile = render_compact, untile
        self.tensor_device = tensor_device if tensor_device is not None else torch.device("cpu")
        # always_gather: a single rank still sends its frame through the process group (to itself),
        # which exercises the collective path on one GPU
        self.always_gather = bool(always_gather) and dist.is_initialized()
        self.in_flight = collections.deque()  # (buffer index, pending communication), oldest first
        self.submitted = 0
        self.set_root_share(root_share)

    def set_root_share(self, root_share):
        """Rank 0 receives every other rank's tiles and sends none, so on short frames the exchange,
        not the render, sets the pace.  root_share = a > 1 deals the tiles to `world - 1 + a` virtual
        ranks and gives a of them to rank 0: it renders a / (world - 1 + a) of the frame, the other
        ranks send correspondingly less.  a = 1 is the plain round-robin with one gather per frame;
        a > 1 exchanges with one send per rank and matching receives on rank 0 (message sizes of
        rank 0's and the other ranks' shares differ, which a gather cannot express)."""
        if self.in_flight:
            raise RuntimeError("DistributedRenderer: collect() the frames in flight before changing the partition")
        a = int(root_share)
        if a < 1:
            raise ValueError("root_share must be at least 1")
        if self.world == 1:
            a = 1
        torch, dev = self.torch, self.tensor_device
        self.root_share = a
        self.virtual_world = self.world - 1 + a
        vw = self.virtual_world
        self.slot = (slot_pixels(self.width, self.height, vw) if self.samples_per_axis == 1 or vw == 1
                     else slot_pixels_sampled(self.width, self.height, self.samples_per_axis, vw))
        self.rgb_bytes = self.slot * 12 if self.want_rgb else 0
        self.message_bytes = self.rgb_bytes + (self.slot * 4 if self.want_rgba8 else 0)
        self.messages = self.gathered = None
        if a == 1 or self.rank != 0:  # with a > 1 rank 0 renders straight into its rows of the gathered buffer
            self.messages = [torch.zeros(self.message_bytes, dtype=torch.uint8, device=dev) for _ in range(self.DEPTH)]
        if self.rank == 0 and vw > 1:
            self.gathered = [torch.zeros((vw, self.message_bytes), dtype=torch.uint8, device=dev) for _ in range(self.DEPTH)]
        if self.always_gather and self.gathered is None:
            self.gathered = [torch.zeros((1, self.message_bytes), dtype=torch.uint8, device=dev) for _ in range(self.DEPTH)]

    def _views(self, message):
        """([slot, 3] float32, [slot, 4] uint8) views of one rank's message."""
        rgb = message[: self.rgb_bytes].view(self.torch.float32).view(self.slot, 3) if self.want_rgb else None
        rgba8 = message[self.rgb_bytes:].view(self.slot, 4) if self.want_rgba8 else None
        return rgb, rgba8

    def submit(self):
        """Starts the next frame: this rank's tiles, then (world > 1) the exchange, without waiting."""
        if len(self.in_flight) >= self.DEPTH:
            raise RuntimeError("DistributedRenderer: collect() a frame before submitting a third")
        i = self.submitted % self.DEPTH
        self.submitted += 1
        dist, a, vw = self.dist, self.root_share, self.virtual_world
        work = None
        if a == 1:
            rgb, rgba8 = self._views(self.messages[i])
            self.render_compact(self.rank, self.world, rgb, rgba8)
            if self.world > 1 or self.always_gather:
                targets = list(self.gathered[i].unbind(0)) if self.rank == 0 else None
                work = [dist.gather(self.messages[i], gather_list=targets, dst=0, group=self.group, async_op=True)]
        elif self.rank == 0:
            for v in range(a):  # rank 0 is virtual ranks 0 .. a-1
                rgb, rgba8 = self._views(self.gathered[i][v])
                self.render_compact(v, vw, rgb, rgba8)
            ops = [dist.P2POp(dist.irecv, self.gathered[i][a + r - 1], self._peer(r), self.group) for r in range(1, self.world)]
            work = dist.batch_isend_irecv(ops)
        else:
            rgb, rgba8 = self._views(self.messages[i])
            self.render_compact(a + self.rank - 1, vw, rgb, rgba8)
            work = dist.batch_isend_irecv([dist.P2POp(dist.isend, self.messages[i], self._peer(0), self.group)])
        self.in_flight.append((i, work))

    def _peer(self, group_rank):
        """Global rank of a rank of this renderer's group (what point-to-point calls address)."""
        return group_rank if self.group is None else self.dist.get_global_rank(self.group, group_rank)

    def collect(self):
        """Finishes the oldest frame in flight.  Returns (rgb, rgba8) frames on rank 0, (None, None) elsewhere."""
        i, work = self.in_flight.popleft()
        for w in work or ():
            w.wait()  # device tensors: the current stream waits for the exchange; host tensors: the host does
        if self.world == 1:
            n = self.width * self.height
            rgb, rgba8 = self._views(self.gathered[i][0] if work else self.messages[i])
            rgb = rgb[:n].view(self.height, self.width, 3) if rgb is not None else None
            rgba8 = rgba8[:n].view(self.height, self.width, 4) if rgba8 is not None else None
            return rgb, rgba8
        if self.rank != 0:
            return None, None
        rgb0, rgba0 = self._views(self.gathered[i][0])
        return self.untile(self.gathered[i], rgb0, rgba0, self.message_bytes)

    def autotune(self, candidates=(1, 2, 3, 4, 6), frames=6, sync=None):
        """Picks root_share by timing a short stream of frames per candidate (every rank calls this;
        rank 0's clock decides and the choice is broadcast).  `sync` waits for the device, if any."""
        import time

        if self.world == 1:
            return 1
        torch, dist = self.torch, self.dist
        best, best_t = 1, None
        for a in candidates:
            self.set_root_share(a)
            for _ in self.stream(2):  # connections and buffers of this partition
                pass
            if sync:
                sync()
            dist.barrier(group=self.group)
            t0 = time.perf_counter()
            for _ in self.stream(frames):
                pass
            if sync:
                sync()
            dist.barrier(group=self.group)
            t = time.perf_counter() - t0
            if best_t is None or t < best_t:
                best, best_t = a, t
        choice = torch.tensor([best], dtype=torch.int64, device=self.tensor_device)
        dist.broadcast(choice, src=self._peer(0), group=self.group)
        self.set_root_share(int(choice.item()))
        return self.root_share

    def frame(self):
        """One frame, start to finish."""
        self.submit()
        return self.collect()

    def stream(self, n):
        """n frames with two in flight; yields each frame's (rgb, rgba8) in order."""
        for k in range(n):
            self.submit()
            if k > 0:
                yield self.collect()
        if n > 0:
            yield self.collect()


def gpu_renderer(device, handle, camera, group=None, want_rgb=True, want_rgba8=True, always_gather=False,
                 samples_per_axis=1, root_share=1):
    """DistributedRenderer wired to the HIP path: compact tile render + device untile.
    samples_per_axis > 1 (2, 4 or 8 when there is more than one rank): that many squared samples per pixel."""
    cam = camera.to_abi() if hasattr(camera, "to_abi") else camera
    w, h = cam.width, cam.height
    spa = int(samples_per_axis)

    def render_compact(rank, world, out_rgb, out_rgba8):
        if spa == 1:
            _render_compact(device, handle, cam, rank, world, out_rgb, out_rgba8)
        else:
            device.render_device_sampled(handle, cam, spa, rank, max(world, 1), out_rgb is not None, out_rgba8 is not None,
                                         out_rgb, out_rgba8)

    frames = {}

    def untile(message, rgb0, rgba0, stride_bytes):
        world = message.shape[0]
        torch = device.torch
        if rgb0 is not None and "rgb" not in frames:
            frames["rgb"] = torch.empty((h, w, 3), dtype=torch.float32, device=device.tdev)
        if rgba0 is not None and "rgba8" not in frames:
            frames["rgba8"] = torch.empty((h, w, 4), dtype=torch.uint8, device=device.tdev)
        return device.untile(w, h, world, slot_pixels(w, h, world), rgb0, rgba0, frames.get("rgb"), frames.get("rgba8"),
                             rank_stride_bytes=stride_bytes, samples_per_axis=spa)

    return DistributedRenderer(w, h, group, want_rgb, want_rgba8, device.tdev, render_compact, untile, always_gather, spa,
                               root_share)


def _render_compact(device, handle, cam, rank, world, out_rgb, out_rgba8):
    import ctypes as C

    from . import device as _device

    lib = _device.library()
    tiling = abi.rtk_tiling(rank, max(world, 1))
    with _device._OnStream(device) as stream:
        _device._check(lib.rtk_render_device(
            device.ctx, handle.handle, C.byref(cam), C.byref(tiling),
            C.c_void_p(out_rgb.data_ptr()) if out_rgb is not None else None,
            C.c_void_p(out_rgba8.data_ptr()) if out_rgba8 is not None else None,
            stream))
