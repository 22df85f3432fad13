import json, os, sys
sys.path.insert(0, os.getcwd())
import torch
import ascendpathtracing_amd as apt
from ascendpathtracing_amd import gen_data, render
ns = 10000
scene_h = gen_data.gen_scene(ns, seed=1)
scene = torch.from_numpy(scene_h).cuda()
grid = torch.from_numpy(gen_data.build_grid(scene_h, ns).view("int32")).cuda()
def timed(p, reps=5):
    render.render_frame(p, scene); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); render.render_frame(p, scene); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return round(best, 3)
for s_ in (16, 64):
    p = apt.make_params(1920, 1080, s_, depth=8, num_spheres=ns, accel=grid.data_ptr())
    for ppw in (0, 4, 6, 8, 10, 12, 16, 20, 24, 32):
        render.set_debug("queue_ppw", ppw)
        print(json.dumps({"S": s_, "ppw": ppw, "ms": timed(p, 4 if s_ == 64 else 6)}), flush=True)
render.set_debug("queue_ppw", 0)
