import os, sys, torch, json
sys.path.insert(0, os.getcwd())
import ascendpathtracing_amd as apt
from ascendpathtracing_amd import gen_data, render
sph=torch.from_numpy(gen_data.gen_spheres()).cuda()
for d in (8,32):
    p=apt.make_params(1920,1080,64,depth=d,flags=apt.APT_FLAG_RETIRE)
    render.render_frame(p,sph); torch.cuda.synchronize()
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    best=1e9
    for _ in range(3):
        a.record(); render.render_frame(p,sph); b.record(); torch.cuda.synchronize(); best=min(best,a.elapsed_time(b))
    with render.TraceCounter() as tc:
        render.render_frame(p,sph)
    t,bb,g=tc.stats; n=p.num_paths
    print(os.environ.get("APT_REFILL_LANES"),"D",d,"ms",round(best,2),"bounce-slots/path",round(bb/n,3),"raygen-slots/path",round(g/n,3))
