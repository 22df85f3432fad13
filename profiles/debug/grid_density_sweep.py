#!/usr/bin/env python3
"""C4 at 64 spp against the grid's cell size (apt_set_debug("grid_spheres_per_cell", v): sphere centres per cell the builders size the cells for) and
the batch threshold -- re-run whenever the walk's cost structure changes.   python profiles/debug/grid_density_sweep.py"""
import hashlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import ascendpathtracing_amd as apt
from ascendpathtracing_amd import gen_data, render
ns = 10000
scene_h = gen_data.gen_scene(ns, seed=1)
scene = torch.from_numpy(scene_h).cuda()


def timed(p, reps=5):
    render.render_frame(p, scene); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fb, u8 = render.render_frame(p, scene); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return round(best, 3), hashlib.sha256(u8.cpu().numpy().tobytes()).hexdigest()[:16]


for v in (0.0, 1.0, 0.7, 0.5, 0.4, 0.3, 0.25, 0.2, 0.15):
    render.set_debug("grid_spheres_per_cell", v)
    grid = torch.from_numpy(gen_data.build_grid(scene_h, ns).view("int32")).cuda()
    p = apt.make_params(1920, 1080, 16, depth=8, num_spheres=ns, accel=grid.data_ptr())
    ms, sha = timed(p)
    with render.TraceCounter() as tc:
        render.render_frame(p, scene)
    traced, cells, tests = tc.stats
    print(json.dumps({"spheres_per_cell": v, "ms": ms, "retire_ms": timed(p.copy(flags=apt.APT_FLAG_RETIRE))[0], "sha": sha, "grid_MB": round(grid.numel() * 4 / 1e6, 2),
                      "cells_per_segment": round(cells / traced, 2), "tests_per_segment": round(tests / traced, 2)}), flush=True)
render.set_debug("grid_spheres_per_cell", 0.0)
grid = torch.from_numpy(gen_data.build_grid(scene_h, ns).view("int32")).cuda()
p = apt.make_params(1920, 1080, 16, depth=8, num_spheres=ns, accel=grid.data_ptr())
for lanes in (24, 28, 32, 36, 40, 44, 48):
    render.set_refill_lanes(lanes)
    print(json.dumps({"batch_lanes": lanes, "ms": timed(p)[0], "retire_ms": timed(p.copy(flags=apt.APT_FLAG_RETIRE))[0]}), flush=True)
render.set_refill_lanes(32)
