#!/usr/bin/env python3
"""C3 (4096 x 4096, 1024 spp, depth 8) on ONE GPU, band by band: the kernel time of the band each rank of an N-rank job would render
(dist.split_range: the reference's own contiguous split, src/render.cpp:9-10,24-27), for N = 1, 2, 4, 8.  NOT a multi-GPU run: it
measures the load balance of the split (what a strong-scaled job's slowest rank would take) and nothing about the gather, which moves
201 MB / N per rank once per frame.  python profiles/debug/c3_band_times.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import ascendpathtracing_amd as apt
from ascendpathtracing_amd import dist as apt_dist, gen_data, render

W, H, S, D = 4096, 4096, 256, 8
p = apt.make_params(W, H, S, depth=D, seed=0)
sph = torch.from_numpy(gen_data.gen_spheres()).cuda()
npix = W * H
seg_total = npix * 4 * S * D


def band_ms(b, c, fb, u8, reps=2):
    render.render_frame(p, sph, b, c, fb=fb, fb_u8=u8); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, e in ev:
        a.record(); render.render_frame(p, sph, b, c, fb=fb, fb_u8=u8); e.record()
    torch.cuda.synchronize()
    return min(a.elapsed_time(e) for a, e in ev)


t1 = None
for n in (1, 2, 4, 8):
    times = []
    for r in range(n):
        b, c = apt_dist.split_range(npix, r, n)
        fb = torch.empty((3, c), dtype=torch.float32, device="cuda")
        u8 = torch.empty((c, 3), dtype=torch.uint8, device="cuda")
        times.append(round(band_ms(b, c, fb, u8), 3))
        del fb, u8
    if n == 1:
        t1 = times[0]
    slowest = max(times)
    print(json.dumps({"ranks": n, "band_kernel_ms": times, "slowest_band_ms": slowest, "imbalance_max_over_mean": round(slowest / (sum(times) / n), 4),
                      "kernel_only_strong_scaling_efficiency": round(t1 / (n * slowest), 4),
                      "Gray_per_s_if_the_gather_is_covered": round(seg_total / slowest / 1e6, 1),
                      "note": "one GPU, one band at a time: load balance of the split only, no collective, no xGMI"}), flush=True)
