#!/usr/bin/env python3
"""apt_decode_color_device through the product library on a C2-sized colour buffer (6.4 GB of random floats): HIP-event time per launch
(min / median of 9), per sample count.  python profiles/debug/decode_time.py [S ...]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import ascendpathtracing_amd as apt
from ascendpathtracing_amd import render

for s in [int(x) for x in sys.argv[1:]] or [8, 16, 32, 64, 256]:
    w, h = 1920, 1080
    p = apt.make_params(w, h, s)
    col = torch.rand(3 * p.num_paths, device="cuda") * 1.3
    render.decode_color_device(p, col); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(9)]
    for a, b in ev:
        a.record(); render.decode_color_device(p, col); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)
    gb = col.numel() * 4 / 1e9
    print(json.dumps({"S": s, "ms_min": round(t[0], 3), "ms_median": round(t[4], 3), "TBps_at_min": round(gb / t[0], 2)}), flush=True)
    del col
    torch.cuda.empty_cache()
