#!/usr/bin/env python3
"""Slot utilisation of the sample-queue kernel: traced segments against the lane-slots of its bounce executions (64 per turn of the hot
loop) and of its ray-generate passes, per case and per pixels-per-wave setting.   python profiles/debug/queue_util.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import ascendpathtracing_amd as apt
from ascendpathtracing_amd import gen_data, render
sph = torch.from_numpy(gen_data.gen_spheres()).cuda()
CASES = {"c2_retire": (8, apt.APT_FLAG_RETIRE), "c5_retire": (32, apt.APT_FLAG_RETIRE), "c5_rr_retire": (32, apt.APT_FLAG_RETIRE | apt.APT_FLAG_RR)}
paths = 1920 * 1080 * 256
for ppw in (0, 4, 8, 32, 64):
    render.set_debug("queue_ppw", ppw)
    for case, (d, flags) in CASES.items():
        p = apt.make_params(1920, 1080, 64, depth=d, flags=flags)
        with render.TraceCounter() as tc:
            render.render_frame(p, sph)
        traced, bslots, gslots = tc.stats
        render.render_frame(p, sph); torch.cuda.synchronize()
        x, y = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        x.record(); render.render_frame(p, sph); y.record(); torch.cuda.synchronize()
        print(json.dumps({"ppw": ppw, "case": case, "ms": round(x.elapsed_time(y), 3), "segments_per_path": round(traced / paths, 3),
                          "bounce_slot_utilisation": round(traced / bslots, 4), "generate_slot_utilisation": round(paths / gslots, 4),
                          "bounce_turns_per_wave_pixel": round(bslots / 64 / (1920 * 1080), 2)}), flush=True)
