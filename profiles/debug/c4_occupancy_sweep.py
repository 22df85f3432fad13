#!/usr/bin/env python3
"""What LDS costs the sample-queue kernel's grid form (C4: 10 000 spheres, 1080p): the same frame with `queue_lds_pad`
bytes of extra dynamic LDS per wave, i.e. at 20 / 18 / 16 / 14 / 12 / 10 waves per CU (gfx950 hands LDS out in granules of
1280 bytes, 128 per CU; the kernel needs 6 of them).  This is the measurement behind the decision NOT to build the
finished / ready context queues: 64 spare walk contexts are 6.9 KB per wave.
    python profiles/debug/c4_occupancy_sweep.py [--s 16] [--reps 3] > profiles/r05_c4_occupancy_sweep.jsonl"""
import argparse, hashlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import ascendpathtracing_amd as apt
from ascendpathtracing_amd import gen_data, render

ap = argparse.ArgumentParser()
ap.add_argument("--s", type=int, default=16)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--ns", type=int, default=10000)
args = ap.parse_args()
scene = torch.from_numpy(gen_data.gen_scene(args.ns, seed=1)).cuda()
grid = gen_data.build_grid_device(scene, args.ns)
torch.cuda.synchronize()
GRANULE = 1280
maxleaf = min(args.s, 128)          # pt_queue.h queue_lds_bytes() for the grid form without roulette, one pairwise leaf (S <= 128)
nbuf = max(2, min(16, (512 + 4 * maxleaf - 1) // (4 * maxleaf)))
BASE = 2 * 32 * 16 + 16 * 16 + 80 + nbuf * (4 * maxleaf * 12 + 16)
assert BASE <= 6 * GRANULE, BASE
for retire in (0, apt.APT_FLAG_RETIRE):
    p = apt.make_params(1920, 1080, args.s, depth=8, num_spheres=args.ns, accel=grid.data_ptr(), flags=retire | gen_data.grid_flags(grid, args.ns))
    ref = None
    for extra_granules in (0, 1, 2, 3, 4, 6, 8):
        pad = 0 if extra_granules == 0 else (6 + extra_granules) * GRANULE - BASE - 8
        with render.debug_knob("queue_lds_pad", pad):
            fb, u8 = render.render_frame(p, scene)
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(args.reps):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); fb, u8 = render.render_frame(p, scene); b.record(); torch.cuda.synchronize()
                best = min(best, a.elapsed_time(b))
        sha = hashlib.sha256(u8.cpu().numpy().tobytes()).hexdigest()[:16]
        ref = ref or sha
        granules = 6 + extra_granules
        print(json.dumps({"retire": bool(retire), "S": args.s, "lds_pad_bytes": pad, "lds_granules_per_wave": granules,
                          "waves_per_cu_by_lds": min(20, 128 // granules), "ms": round(best, 3), "same_frame": sha == ref}), flush=True)
render.check_device_status()
