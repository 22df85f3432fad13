import sys, os, json, torch
sys.path.insert(0, os.getcwd())
import ascendpathtracing_amd as apt
from ascendpathtracing_amd import gen_data, render
import numpy as np
ns=10000
scene_h=gen_data.gen_scene(ns, seed=1)
scene=torch.from_numpy(scene_h).cuda()
gh=torch.from_numpy(gen_data.build_grid(scene_h, ns).view(np.int32)).cuda()
gd=gen_data.build_grid_device(scene, ns); torch.cuda.synchronize()
print("equal", torch.equal(gh, gd), gh.data_ptr()%4096, gd.data_ptr()%4096)
def t(p, reps=3):
    render.render_frame(p, scene); torch.cuda.synchronize()
    out=[]
    for _ in range(reps):
        a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        a.record(); render.render_frame(p, scene); b.record(); torch.cuda.synchronize(); out.append(round(a.elapsed_time(b),2))
    return out
for name,g in (("host",gh),("device",gd),("host",gh)):
    p=apt.make_params(1920,1080,64,depth=8,num_spheres=ns,accel=g.data_ptr())
    print(name, t(p))
fb=torch.empty((3,1920*1080),device="cuda"); 
p=apt.make_params(1920,1080,64,depth=8,num_spheres=ns,accel=gh.data_ptr())
print("after alloc", t(p))
