import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import ascendpathtracing_amd as apt
from ascendpathtracing_amd import gen_data, render, dist as apt_dist
mode = sys.argv[1]
p = apt.make_params(1920, 1080, 64, depth=8, num_spheres=8, seed=0)
sph = torch.from_numpy(gen_data.gen_spheres()).cuda()
n = 1920 * 1080
if mode in ("same", "events", "sync"):
    fb = torch.empty((3, n), dtype=torch.float32, device="cuda"); u8 = torch.empty((n, 3), dtype=torch.uint8, device="cuda")
    for k in range(8):
        if mode == "events":
            a = torch.cuda.Event(enable_timing=True); a.record()
        render.render_frame(p, sph, fb=fb, fb_u8=u8)
        if mode == "events":
            b = torch.cuda.Event(enable_timing=True); b.record()
        if mode == "sync":
            torch.cuda.synchronize()
elif mode == "alt":
    bufs = [(torch.empty((3, n), dtype=torch.float32, device="cuda"), torch.empty((n, 3), dtype=torch.uint8, device="cuda")) for _ in range(2)]
    for k in range(8):
        render.render_frame(p, sph, fb=bufs[k % 2][0], fb_u8=bufs[k % 2][1])
elif mode == "shard":
    shard = apt_dist.FrameShard(p, 0, 1, slots=2)
    slots = shard.alloc_slots()
    for k in range(8):
        shard.render(slots[k % 2], sph, render.render_frame, slot=k % 2)
elif mode == "fresh":
    for k in range(8):
        fb = torch.empty((3, n), dtype=torch.float32, device="cuda"); u8 = torch.empty((n, 3), dtype=torch.uint8, device="cuda")
        render.render_frame(p, sph, fb=fb, fb_u8=u8)
torch.cuda.synchronize()
