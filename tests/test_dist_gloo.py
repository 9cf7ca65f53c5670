"""The multi-rank frame path on CPU: two gloo ranks each produce their tiles (with the CPU
oracle standing in for the device render), gather to rank 0 and de-interleave.  The frame
must equal the single-rank frame bit for bit -- tiles are independent
(src/scene/cameras.rs:75-82 has no cross-iteration state)."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as tdist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, w, h, out_path, spa=1, root_share=1, autotune=False):
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    rtk = importlib.import_module("raytracer-rust_amd")
    rdist = importlib.import_module("raytracer-rust_amd.dist")
    import oracle_binding as ob

    scene, camera = rtk.synthetic.config("C2g", w, h)

    def render_compact(r, wd, out_rgb, out_rgba8):
        # what rtk_render_device(_sampled) writes for tiling {r, wd}: local tile k = global tile k*wd + r
        # of the (sample) frame, a (32/s) x (8/s) block of pixels
        tw, th = rdist.TILE_W // spa, rdist.TILE_H // spa
        per_tile = tw * th
        tx, ty = rdist.tile_grid(w * spa, h * spa)
        out_rgb.zero_()
        out_rgba8.zero_()
        k = 0
        for tile in range(r, tx * ty, wd):
            x0, y0 = (tile % tx) * tw, (tile // tx) * th
            x1, y1 = min(x0 + tw, w), min(y0 + th, h)
            block3 = np.zeros((th, tw, 3), np.float32)
            block4 = np.zeros((th, tw, 4), np.uint8)
            if x0 < w and y0 < h:
                rgb, rgba, _ = ob.render(scene, camera, rect=(x0, y0, x1, y1), samples_per_axis=spa)
                block3[: y1 - y0, : x1 - x0] = rgb.astype(np.float32)
                block4[: y1 - y0, : x1 - x0] = rgba
            out_rgb[k * per_tile:(k + 1) * per_tile] = torch.from_numpy(block3.reshape(per_tile, 3))
            out_rgba8[k * per_tile:(k + 1) * per_tile] = torch.from_numpy(block4.reshape(per_tile, 4))
            k += 1
        assert k == rdist.tiles_for_rank(w * spa, h * spa, r, wd)

    def untile(message, rgb0, rgba0, stride_bytes):
        # one message per (virtual) rank: its FP32 tiles, then its RGBA8 tiles
        vw = message.shape[0]
        slot = rdist.slot_pixels_sampled(w, h, spa, vw)
        assert slot == rdist.slot_pixels(w, h, vw) or spa > 1
        assert message.shape == (vw, stride_bytes) and stride_bytes == slot * 16
        g_rgb = np.concatenate([message[r, : slot * 12].numpy().view(np.float32).reshape(slot, 3) for r in range(vw)])
        g_rgba8 = np.concatenate([message[r, slot * 12:].numpy().reshape(slot, 4) for r in range(vw)])
        if spa == 1:
            return (rdist.untile_host(w, h, vw, slot, g_rgb, 3), rdist.untile_host(w, h, vw, slot, g_rgba8, 4))
        return (rdist.untile_host_sampled(w, h, spa, vw, slot, g_rgb, 3),
                rdist.untile_host_sampled(w, h, spa, vw, slot, g_rgba8, 4))

    renderer = rdist.DistributedRenderer(w, h, render_compact=render_compact, untile=untile, samples_per_axis=spa,
                                         root_share=root_share)
    assert renderer.world == world and renderer.rank == rank
    assert renderer.virtual_world == world - 1 + root_share
    if autotune:
        chosen = renderer.autotune(candidates=(1, 2, 3), frames=2)
        assert chosen in (1, 2, 3) and renderer.root_share == chosen and not renderer.in_flight
        agreed = torch.tensor([chosen])
        tdist.all_reduce(agreed, op=tdist.ReduceOp.MAX)
        assert int(agreed.item()) == chosen  # every rank ended on the same partition
    rgb, rgba = renderer.frame()
    if rank == 0:
        np.savez(out_path, rgb=rgb, rgba=rgba)
    else:
        assert rgb is None and rgba is None
    # a stream of frames with two in flight delivers the same frame every time, in order
    n = 0
    for f_rgb, f_rgba in renderer.stream(3):
        n += 1
        if rank == 0:
            assert np.array_equal(f_rgb, rgb) and np.array_equal(f_rgba, rgba)
        else:
            assert f_rgb is None and f_rgba is None
    assert n == 3 and not renderer.in_flight
    renderer.submit()
    renderer.submit()
    try:
        renderer.submit()
        raise AssertionError("a third frame in flight was accepted")
    except RuntimeError:
        pass
    renderer.collect()
    renderer.collect()
    tdist.barrier()
    tdist.destroy_process_group()


@pytest.mark.parametrize("world,w,h", [(2, 96, 40), (2, 70, 19), (3, 100, 30)])
def test_multi_rank_frame_equals_single_rank(rtk, oracle, tmp_path, world, w, h):
    out = str(tmp_path / "frame.npz")
    mp.spawn(_worker, args=(world, _free_port(), w, h, out), nprocs=world, join=True)
    got = np.load(out)
    scene, camera = rtk.synthetic.config("C2g", w, h)
    rgb, rgba, _ = oracle.render(scene, camera)
    assert np.array_equal(got["rgba"], rgba)
    assert np.array_equal(got["rgb"], rgb.astype(np.float32))


@pytest.mark.parametrize("world,w,h,spa", [(2, 50, 21, 2), (3, 23, 9, 4)])
def test_multi_rank_sampled_frame_equals_single_rank(rtk, oracle, tmp_path, world, w, h, spa):
    out = str(tmp_path / "frame.npz")
    mp.spawn(_worker, args=(world, _free_port(), w, h, out, spa), nprocs=world, join=True)
    got = np.load(out)
    scene, camera = rtk.synthetic.config("C2g", w, h)
    rgb, rgba, _ = oracle.render(scene, camera, samples_per_axis=spa)
    assert np.array_equal(got["rgba"], rgba)
    assert np.array_equal(got["rgb"], rgb.astype(np.float32))


@pytest.mark.parametrize("world,w,h,spa,share", [(2, 96, 40, 1, 2), (3, 70, 19, 1, 3), (2, 50, 21, 2, 2), (4, 64, 24, 1, 6)])
def test_weighted_partition_equals_single_rank(rtk, oracle, tmp_path, world, w, h, spa, share):
    """Rank 0 takes `share` of the `world - 1 + share` virtual ranks and receives the others' tiles
    with point-to-point transfers; the frame is the single-rank frame."""
    out = str(tmp_path / "frame.npz")
    mp.spawn(_worker, args=(world, _free_port(), w, h, out, spa, share), nprocs=world, join=True)
    got = np.load(out)
    scene, camera = rtk.synthetic.config("C2g", w, h)
    rgb, rgba, _ = oracle.render(scene, camera, samples_per_axis=spa)
    assert np.array_equal(got["rgba"], rgba)
    assert np.array_equal(got["rgb"], rgb.astype(np.float32))


def test_partition_autotune_agrees_across_ranks(rtk, oracle, tmp_path):
    out = str(tmp_path / "frame.npz")
    mp.spawn(_worker, args=(2, _free_port(), 64, 24, out, 1, 1, True), nprocs=2, join=True)
    got = np.load(out)
    scene, camera = rtk.synthetic.config("C2g", 64, 24)
    _, rgba, _ = oracle.render(scene, camera)
    assert np.array_equal(got["rgba"], rgba)


def test_untile_host_round_trip(rtk):
    rdist = importlib.import_module("raytracer-rust_amd.dist")
    rng = np.random.default_rng(0)
    for w, h, world in [(33, 9, 3), (64, 16, 2), (100, 37, 8), (5, 5, 4)]:
        frame = rng.integers(0, 255, size=(h, w, 4), dtype=np.uint8)
        slot = rdist.slot_pixels(w, h, world)
        compact = np.zeros((world * slot, 4), np.uint8)
        tx, ty = rdist.tile_grid(w, h)
        for tile in range(tx * ty):
            r, k = tile % world, tile // world
            x0, y0, x1, y1 = rdist.tile_rect(w, h, tile)
            block = np.zeros((rdist.TILE_H, rdist.TILE_W, 4), np.uint8)
            block[: y1 - y0, : x1 - x0] = frame[y0:y1, x0:x1]
            compact[r * slot + k * 256: r * slot + (k + 1) * 256] = block.reshape(256, 4)
        assert np.array_equal(rdist.untile_host(w, h, world, slot, compact, 4), frame)
