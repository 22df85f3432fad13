import os, sys, torch, json
sys.path.insert(0, os.getcwd())
import ascendpathtracing_amd as apt
from ascendpathtracing_amd import gen_data, render
sph = torch.from_numpy(gen_data.gen_spheres()).cuda()
out = {"lib": os.environ.get("APT_LIB_PATH", "default")}
for d, flags, name in ((8, 1, "c2_retire"), (32, 1, "c5_retire"), (32, 3, "c5_rr_retire")):
    p = apt.make_params(1920, 1080, 64, depth=d, flags=flags)
    render.render_frame(p, sph); torch.cuda.synchronize()
    best = 1e9
    for _ in range(4):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); render.render_frame(p, sph); b.record(); torch.cuda.synchronize(); best = min(best, a.elapsed_time(b))
    out[name] = round(best, 3)
print(json.dumps(out))
