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


def _worker(rank, world, port, w, h, s, out_path):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import ascendpathtracing_amd as apt
    from ascendpathtracing_amd import dist as apt_dist, gen_data
    p = apt.make_params(w, h, s, depth=4, seed=5)
    sph = torch.from_numpy(gen_data.gen_spheres())
    fb, u8 = apt_dist.render_frame_sharded(p, sph, render_fn=_oracle_render_fn, device="cpu")
    if rank == 0:
        np.savez(out_path, fb=fb.numpy(), u8=u8.numpy())
    else:
        assert fb is None and u8 is None
    dist.barrier()
    dist.destroy_process_group()


def _worker_pipelined(rank, world, port, out_path):
    """Three frames (different seeds) through the double-buffered asynchronous gather bench.py uses."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import ascendpathtracing_amd as apt
    from ascendpathtracing_amd import dist as apt_dist, gen_data
    sph = torch.from_numpy(gen_data.gen_spheres())
    shard = apt_dist.FrameShard(apt.make_params(8, 6, 1, depth=3), rank, world, device="cpu", slots=2)
    slots = shard.alloc_slots()
    frames = []
    full = None
    for k in range(3):
        p = apt.make_params(8, 6, 1, depth=3, seed=k)
        fb, u8 = slots[k % 2]
        _oracle_render_fn(p, sph, shard.pixel_begin, shard.pixel_count, fb=fb, fb_u8=u8)
        full = shard.alloc_full() if rank == 0 else (None, None)
        shard.gather_async(k % 2, *full)      # also completes frame k-1
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


@pytest.mark.parametrize("world,w,h,s", [(2, 12, 8, 2), (3, 7, 5, 1)])
def test_sharded_frame_equals_single_rank(tmp_path, oracle, world, w, h, s):
    out = str(tmp_path / "full.npz")
    mp.spawn(_worker, args=(world, _free_port(), w, h, s, out), nprocs=world, join=True)
    got = np.load(out)
    sph = oracle.gen_spheres()
    fb, u8, _, _ = oracle.render_frame(oracle.make_params(w, h, s, depth=4, seed=5), sph)
    assert np.array_equal(got["fb"].view(np.uint32), fb.view(np.uint32))
    assert np.array_equal(got["u8"], u8)


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
