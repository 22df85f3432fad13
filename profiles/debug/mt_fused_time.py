#!/usr/bin/env python3
"""C2 through the reference's exact pipeline: the fused MT19937 frame kernel against the three-kernel form.
   python profiles/debug/mt_fused_time.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import ascendpathtracing_amd as apt
from ascendpathtracing_amd import gen_data, render

W, H, S, D = 1920, 1080, 64, 8
sph = torch.from_numpy(gen_data.gen_spheres()).cuda()


def timeit(fn, reps=4):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return round(best, 3)


t0 = time.time()
ck, g_lo = render.mt_group_checkpoints(W, H, S, seed=0)
t_ck = time.time() - t0
ck_d = (torch.from_numpy(ck.view(np.int32)).cuda(), g_lo)
out = {"host_group_checkpoints_s": round(t_ck, 2), "checkpoint_table_MB": round(ck.nbytes / 1e6, 1)}
for mode, name in ((apt.APT_MODE_ORACLE, "o_mode"), (apt.APT_MODE_KERNEL, "k_mode")):
    out["fused_" + name + "_ms"] = timeit(lambda: render.render_reference_frame_fused(W, H, S, depth=D, seed=0, spheres=sph, mode=mode, checkpoints=ck_d))
out["fused_depth0_ms"] = timeit(lambda: render.render_reference_frame_fused(W, H, S, depth=0, seed=0, spheres=sph, checkpoints=ck_d))
print(json.dumps(out), flush=True)
