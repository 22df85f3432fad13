#!/usr/bin/env python3
"""The sample-queue kernels against their launch-shape knobs (pixels per wave, colour buffers), round 5's build (chunked XCD
mapping): C2 + retirement, depth 32 + retirement, C5 as named.  python profiles/debug/queue_knob_sweep.py > profiles/r05_queue_knob_sweep.jsonl"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import ascendpathtracing_amd as apt
from ascendpathtracing_amd import gen_data, render
sph = torch.from_numpy(gen_data.gen_spheres()).cuda()
CASES = ((8, 1, "c2_retire"), (32, 1, "c5_retire"), (32, 3, "c5_rr_retire"))


def best_ms(p, reps=5):
    render.render_frame(p, sph); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); render.render_frame(p, sph); b.record(); torch.cuda.synchronize(); best = min(best, a.elapsed_time(b))
    return round(best, 3)


for knob, values in (("queue_ppw", (0, 4, 8, 12, 16, 24, 32, 48, 64)), ("queue_nbuf", (0, 2, 3, 4))):
    for v in values:
        out = {"knob": knob, "value": v}
        with render.debug_knob(knob, v):
            for d, flags, name in CASES:
                out[name] = best_ms(apt.make_params(1920, 1080, 64, depth=d, flags=flags))
        print(json.dumps(out), flush=True)
render.check_device_status()
