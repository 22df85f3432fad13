import os, sys, torch, json
sys.path.insert(0, os.getcwd())
import ascendpathtracing_amd as apt
from ascendpathtracing_amd import gen_data, render
host = gen_data.gen_scene(10000, seed=1)
scene = torch.from_numpy(host).cuda()
grid = gen_data.build_grid_device(scene, 10000)
out = {"lib": os.environ.get("APT_LIB_PATH", "default")}
for s, name in ((16, "c4_grid_64spp"), (64, "c4_grid_256spp")):
    p = apt.make_params(1920, 1080, s, depth=8, num_spheres=10000, accel=grid.data_ptr())
    render.render_frame(p, scene); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); render.render_frame(p, scene); b.record(); torch.cuda.synchronize(); best = min(best, a.elapsed_time(b))
    out[name] = round(best, 2)
print(json.dumps(out))
