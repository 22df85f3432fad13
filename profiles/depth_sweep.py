#!/usr/bin/env python3
"""Frame time vs depth on C2 geometry: depth 0 isolates ray-generate + accumulation."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ascendpathtracing_amd as apt
from ascendpathtracing_amd import gen_data, render
sph = torch.from_numpy(gen_data.gen_spheres()).cuda()
for d in (0, 1, 2, 4, 8, 16):
    p = apt.make_params(1920, 1080, 64, depth=d)
    render.render_frame(p, sph); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        a.record(); render.render_frame(p, sph); b.record(); torch.cuda.synchronize(); best = min(best, a.elapsed_time(b))
    print(f"depth {d:2d}: {best:8.3f} ms")
