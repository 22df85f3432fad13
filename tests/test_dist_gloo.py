"""N > 1 path on CPU: world_size-2 (and 3, uneven) gloo runs of the pixel-band sharding and
the single framebuffer gather.  The render callable is injected; here it is the oracle (the
product default is the HIP path, which refuses to run without a GPU)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _oracle_render_fn(params, spheres, pixel_begin, pixel_count, fb=None, fb_u8=None):
    from oracle import oracle
    op = oracle.make_params(params.width, params.height, params.samples, depth=params.depth,
                            num_spheres=params.num_spheres, mode=params.mode, seed=params.seed)
    f, u, _, _ = oracle.render_frame(op, spheres.numpy(), pixel_begin=pixel_begin, pixel_count=pixel_count)
    fb.copy_(torch.from_numpy(f))
    fb_u8.copy_(torch.from_numpy(u))


def _worker(rank, world, port, w, h, s, out_path, stripes=1):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import ascendpathtracing_amd as apt
    from ascendpathtracing_amd import dist as apt_dist, gen_data
    p = apt.make_params(w, h, s, depth=4, seed=5)
    sph = torch.from_numpy(gen_data.gen_spheres())
    fb, u8 = apt_dist.render_frame_sharded(p, sph, render_fn=_oracle_render_fn, device="cpu", stripes=stripes)
    if rank == 0:
        np.savez(out_path, fb=fb.numpy(), u8=u8.numpy())
    else:
        assert fb is None and u8 is None
    dist.barrier()
    dist.destroy_process_group()


def _worker_pipelined(rank, world, port, out_path, w=8, h=6, stripes=1, nslots=2):
    """Three frames (different seeds) through the double-buffered asynchronous gather bench.py uses;
    w*h need not divide by world*stripes (unequal shards travel in equally padded buffers).  nslots=1: every render writes the buffer
    whose gather is still pending -- begin() must finish that gather first."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import ascendpathtracing_amd as apt
    from ascendpathtracing_amd import dist as apt_dist, gen_data
    sph = torch.from_numpy(gen_data.gen_spheres())
    shard = apt_dist.FrameShard(apt.make_params(w, h, 1, depth=3), rank, world, device="cpu", slots=nslots, stripes=stripes)
    slots = shard.alloc_slots()
    frames = []
    full = None
    for k in range(3):
        p = apt.make_params(w, h, 1, depth=3, seed=k)
        shard.render(slots[k % nslots], sph, _oracle_render_fn, params=p, slot=k % nslots)
        if nslots == 1:
            assert shard._pending is None      # begin(slot) has finished the gather that was reading this buffer
        full = shard.alloc_full() if rank == 0 else (None, None)
        shard.gather_async(k % nslots, *full)      # also completes frame k-1
        frames.append(full)
    shard.finish()
    if rank == 0:
        np.savez(out_path, **{f"fb{k}": f[0].numpy() for k, f in enumerate(frames)},
                 **{f"u8{k}": f[1].numpy() for k, f in enumerate(frames)})
    dist.barrier()
    dist.destroy_process_group()


def test_pipelined_async_gather(tmp_path, oracle):
    out = str(tmp_path / "frames.npz")
    mp.spawn(_worker_pipelined, args=(2, _free_port(), out), nprocs=2, join=True)
    got = np.load(out)
    sph = oracle.gen_spheres()
    for k in range(3):
        fb, u8, _, _ = oracle.render_frame(oracle.make_params(8, 6, 1, depth=3, seed=k), sph)
        assert np.array_equal(got[f"fb{k}"].view(np.uint32), fb.view(np.uint32)), k
        assert np.array_equal(got[f"u8{k}"], u8)


def test_a_render_into_the_slot_of_a_pending_gather_finishes_it_first(tmp_path, oracle):
    """with slots=1 (or any render into the slot whose gather_async has not been finished) the render would overwrite a
    buffer the collective is still sending from.  FrameShard.begin() now completes that gather before the slot is written."""
    out = str(tmp_path / "frames.npz")
    mp.spawn(_worker_pipelined, args=(2, _free_port(), out, 8, 6, 1, 1), nprocs=2, join=True)
    got = np.load(out)
    sph = oracle.gen_spheres()
    for k in range(3):
        fb, u8, _, _ = oracle.render_frame(oracle.make_params(8, 6, 1, depth=3, seed=k), sph)
        assert np.array_equal(got[f"fb{k}"].view(np.uint32), fb.view(np.uint32)), k
        assert np.array_equal(got[f"u8{k}"], u8)


@pytest.mark.parametrize("world,stripes,w,h", [(3, 1, 7, 5), (2, 3, 9, 7)])
def test_pipelined_async_gather_with_unequal_shards(tmp_path, oracle, world, stripes, w, h):
    """the pipelined gather refused unequal shards.  35 pixels over 3 ranks, 63 over 2 x 3 stripes."""
    out = str(tmp_path / "frames.npz")
    mp.spawn(_worker_pipelined, args=(world, _free_port(), out, w, h, stripes), nprocs=world, join=True)
    got = np.load(out)
    sph = oracle.gen_spheres()
    for k in range(3):
        fb, u8, _, _ = oracle.render_frame(oracle.make_params(w, h, 1, depth=3, seed=k), sph)
        assert np.array_equal(got[f"fb{k}"].view(np.uint32), fb.view(np.uint32)), k
        assert np.array_equal(got[f"u8{k}"], u8)


def test_world_8_with_config_c3_shard_geometry(tmp_path, oracle):
    """BASELINE configs[2] is 4096x4096 over 8 ranks: 8 ranks here, and the split the 8-GPU run will use checked
    at C3's real size (each rank one band of 2^21 pixels = 512 whole image columns, no remainder)."""
    from ascendpathtracing_amd.dist import split_range, stripe_ranges
    npix = 4096 * 4096
    for r in range(8):
        b, c = split_range(npix, r, 8)
        assert (b, c) == (r * 2 ** 21, 2 ** 21) and b % 4096 == 0 and stripe_ranges(npix, r, 8, 1) == [(b, c)]
        assert c * 4 * 256 * 8 == 17179869184          # segments per rank per frame
    out = str(tmp_path / "full.npz")
    mp.spawn(_worker, args=(8, _free_port(), 16, 16, 1, out), nprocs=8, join=True)   # 256 pixels: 32 per rank
    got = np.load(out)
    fb, u8, _, _ = oracle.render_frame(oracle.make_params(16, 16, 1, depth=4, seed=5), oracle.gen_spheres())
    assert np.array_equal(got["fb"].view(np.uint32), fb.view(np.uint32)) and np.array_equal(got["u8"], u8)


@pytest.mark.parametrize("world,w,h,s,stripes", [(2, 12, 8, 2, 1), (3, 7, 5, 1, 1), (3, 11, 5, 1, 4)])
def test_sharded_frame_equals_single_rank(tmp_path, oracle, world, w, h, s, stripes):
    out = str(tmp_path / "full.npz")
    mp.spawn(_worker, args=(world, _free_port(), w, h, s, out, stripes), nprocs=world, join=True)
    got = np.load(out)
    sph = oracle.gen_spheres()
    fb, u8, _, _ = oracle.render_frame(oracle.make_params(w, h, s, depth=4, seed=5), sph)
    assert np.array_equal(got["fb"].view(np.uint32), fb.view(np.uint32))
    assert np.array_equal(got["u8"], u8)


def test_stripes_partition_the_frame():
    from ascendpathtracing_amd.dist import stripe_ranges
    for npix, world, stripes in ((35, 3, 1), (63, 2, 3), (1920 * 1080, 8, 16), (4096 * 4096, 8, 1)):
        seen = sorted(r for k in range(world) for r in stripe_ranges(npix, k, world, stripes))
        pos = 0
        for b, c in seen:
            assert b == pos
            pos += c
        assert pos == npix and len(seen) == world * stripes


def test_split_range_is_a_partition():
    from ascendpathtracing_amd.dist import split_range
    for total in (0, 1, 7, 8, 1920 * 1080, 4096 * 4096):
        for world in (1, 2, 3, 8):
            pos = 0
            for r in range(world):
                b, c = split_range(total, r, world)
                assert b == pos and c in (total // world, total // world + 1)
                pos += c
            assert pos == total
