#!/usr/bin/env python3
"""Why bench.py's C4 figure and variants_bench.py's differed by 4 %: the same frame timed the way each does it, in one process.
   python profiles/debug/c4_bench_paths.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import ascendpathtracing_amd as apt
from ascendpathtracing_amd import gen_data, render
ns = 10000
scene_h = gen_data.gen_scene(ns, seed=1)
scene = torch.from_numpy(scene_h).cuda()
gd = gen_data.build_grid_device(scene, ns)
gh = torch.from_numpy(gen_data.build_grid(scene_h, ns).view("int32")).cuda()
torch.cuda.synchronize()


def timed(fn, reps, warm=1):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t = [a.elapsed_time(b) for a, b in ev]
    return [round(x, 2) for x in t]


for name, g in (("device-built grid", gd), ("host-built grid", gh), ("device-built grid again", gd)):
    p = apt.make_params(1920, 1080, 64, depth=8, num_spheres=ns, accel=g.data_ptr())
    print(json.dumps({"grid": name, "render.render_frame ms per launch (1 warm-up, 6 launches)": timed(lambda: render.render_frame(p, scene), 6)}), flush=True)
fb = torch.empty((3, 1920 * 1080), dtype=torch.float32, device="cuda")
u8 = torch.empty((1920 * 1080, 3), dtype=torch.uint8, device="cuda")
p = apt.make_params(1920, 1080, 64, depth=8, num_spheres=ns, accel=gd.data_ptr())
print(json.dumps({"grid": "device-built, preallocated outputs", "ms": timed(lambda: render.render_frame(p, scene, fb=fb, fb_u8=u8), 6)}), flush=True)
# bench.py renders C4 after the exact-pipeline extra has allocated and freed 19 GB: does where the allocator then puts the tables matter?
big = torch.empty(19 * 10**9, dtype=torch.uint8, device="cuda"); big.fill_(1); torch.cuda.synchronize()
del big
torch.cuda.empty_cache()
scene2 = torch.from_numpy(scene_h).cuda()
g2 = gen_data.build_grid_device(scene2, ns)
torch.cuda.synchronize()
p2 = apt.make_params(1920, 1080, 64, depth=8, num_spheres=ns, accel=g2.data_ptr())
print(json.dumps({"grid": "scene and grid allocated after a 19 GB allocation was freed (empty_cache)", "ms": timed(lambda: render.render_frame(p2, scene2), 6)}), flush=True)
print(json.dumps({"grid": "the first tables again", "ms": timed(lambda: render.render_frame(p, scene), 4)}), flush=True)
print(json.dumps({"ptrs": {"scene": hex(scene.data_ptr()), "grid": hex(gd.data_ptr()), "scene2": hex(scene2.data_ptr()), "grid2": hex(g2.data_ptr())}}))
