"""Per-frame cost of the pipelined RCCL gather path on one GPU (single-rank NCCL group): C2 frames with and
without shard.gather_async."""
import os, socket, sys, time
sys.path.insert(0, os.getcwd())
import torch, torch.distributed as dist
import ascendpathtracing_amd as apt
from ascendpathtracing_amd import gen_data, render, dist as apt_dist
with socket.socket() as sk:
    sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
sph = torch.from_numpy(gen_data.gen_spheres()).cuda()
p = apt.make_params(1920, 1080, 64, depth=8)
shard = apt_dist.FrameShard(p, 0, 1, slots=2)
slots = shard.alloc_slots(); full = shard.alloc_full()
def run(gather, n=20):
    for k in range(3):
        shard.render(slots[k % 2], sph, render.render_frame)
        if gather: shard.gather_async(k % 2, *full)
    shard.finish(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(n):
        shard.render(slots[k % 2], sph, render.render_frame)
        if gather: shard.gather_async(k % 2, *full)
    shard.finish(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
print("no gather  ms/frame", round(run(False), 3))
print("with gather ms/frame", round(run(True), 3))
dist.destroy_process_group()
