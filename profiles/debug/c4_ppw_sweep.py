import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import ascendpathtracing_amd as apt
from ascendpathtracing_amd import gen_data, render
ns = 10000
scene = torch.from_numpy(gen_data.gen_scene(ns, seed=1)).cuda()
grid = gen_data.build_grid_device(scene, ns)
p = apt.make_params(1920, 1080, 64, depth=8, num_spheres=ns, accel=grid.data_ptr(), flags=gen_data.grid_flags(grid, ns))
def best(p, reps=3):
    render.render_frame(p, scene); torch.cuda.synchronize(); b = 1e9
    for _ in range(reps):
        x, y = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        x.record(); render.render_frame(p, scene); y.record(); torch.cuda.synchronize(); b = min(b, x.elapsed_time(y))
    return round(b, 2)
for ppw in (0, 4, 6, 8, 12, 16, 24, 32):
    with render.debug_knob("queue_ppw", ppw):
        print(json.dumps({"queue_ppw": ppw, "c4_ms": best(p), "c4_retire_ms": best(p.copy(flags=p.flags | apt.APT_FLAG_RETIRE))}), flush=True)
