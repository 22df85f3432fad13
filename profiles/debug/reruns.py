"""exact re-runs (waves that left the fast loop) per configuration for the library in APT_LIB_PATH"""
import os, sys, json, torch
sys.path.insert(0, os.getcwd())
import ascendpathtracing_amd as apt
from ascendpathtracing_amd import gen_data, render
sph = torch.from_numpy(gen_data.gen_spheres()).cuda()
out = {"lib": os.environ.get("APT_LIB_PATH", "default")}
for name, d, flags in (("c2", 8, 0), ("c2_retire", 8, 1), ("c5_retire", 32, 1)):
    p = apt.make_params(1920, 1080, 64, depth=d, flags=flags)
    with render.TraceCounter() as tc:
        render.render_frame(p, sph)
    out[name] = {"exact_reruns": tc.exact_reruns, "stats": tc.stats}
print(json.dumps(out))
