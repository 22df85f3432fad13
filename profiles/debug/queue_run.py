#!/usr/bin/env python3
"""One configuration of the sample-queue kernels, a few launches (for rocprofv3 passes) or a timing sweep.
   python profiles/debug/queue_run.py --case c2_retire --reps 3 [--sweep]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import ascendpathtracing_amd as apt
from ascendpathtracing_amd import gen_data, render
ap = argparse.ArgumentParser()
ap.add_argument("--case", default="c2_retire")
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--sweep", action="store_true")
a = ap.parse_args()
sph = torch.from_numpy(gen_data.gen_spheres()).cuda()
CASES = {"c2_retire": (8, apt.APT_FLAG_RETIRE), "c5_retire": (32, apt.APT_FLAG_RETIRE), "c5_rr_retire": (32, apt.APT_FLAG_RETIRE | apt.APT_FLAG_RR),
         "c2_full": (8, 0), "c5_full": (32, 0)}


def timeit(p, reps):
    render.render_frame(p, sph); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        x, y = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        x.record(); render.render_frame(p, sph); y.record(); torch.cuda.synchronize()
        best = min(best, x.elapsed_time(y))
    return round(best, 3)


if not a.sweep:
    d, flags = CASES[a.case]
    print(json.dumps({"case": a.case, "ms": timeit(apt.make_params(1920, 1080, 64, depth=d, flags=flags), a.reps)}))
    sys.exit(0)
# depth slope: every path traced to the depth given (no retirement possible at depth 1; the slope prices a bounce, the intercept ray-generate + sums)
for d in (1, 2, 3, 4, 8):
    out = {"depth": d}
    for name, flags in (("retire", apt.APT_FLAG_RETIRE), ("full", 0)):
        p = apt.make_params(1920, 1080, 64, depth=d, flags=flags)
        with render.TraceCounter() as tc:
            render.render_frame(p, sph)
        out[name] = {"ms": timeit(p, a.reps), "traced": tc.value}
    print(json.dumps(out), flush=True)
# occupancy sensitivity: extra LDS per wave
for pad in (0, 2048, 4096, 8192, 16384):
    render.set_debug("queue_lds_pad", pad)
    out = {"lds_pad": pad}
    for case in ("c2_retire", "c5_retire"):
        d, flags = CASES[case]
        out[case] = timeit(apt.make_params(1920, 1080, 64, depth=d, flags=flags), a.reps)
    print(json.dumps(out), flush=True)
render.set_debug("queue_lds_pad", 0)
for nbuf in (2, 3, 4):
    render.set_debug("queue_nbuf", nbuf)
    out = {"nbuf": nbuf}
    for case in ("c2_retire", "c5_retire"):
        d, flags = CASES[case]
        out[case] = timeit(apt.make_params(1920, 1080, 64, depth=d, flags=flags), a.reps)
    print(json.dumps(out), flush=True)
