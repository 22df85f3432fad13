import os, sys, torch, json
sys.path.insert(0, os.getcwd())
import ascendpathtracing_amd as apt
from ascendpathtracing_amd import gen_data, render
scene = torch.from_numpy(gen_data.gen_scene(10000, seed=1)).cuda()
out = {"lib": os.environ.get("APT_LIB_PATH", "default")}
for d, flags, name in ((8, 0, "c4_d8"), (8, 1, "c4_d8_retire"), (32, 0, "c4_d32"), (32, 1, "c4_d32_retire"), (32, 2, "c4_d32_rr"), (32, 3, "c4_d32_rr_retire")):
    p = apt.make_params(1920, 1080, 8, depth=d, num_spheres=10000, flags=flags)
    render.render_frame(p, scene); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); render.render_frame(p, scene); b.record(); torch.cuda.synchronize()
    out[name] = round(a.elapsed_time(b), 1)
print(json.dumps(out))
