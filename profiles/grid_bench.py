#!/usr/bin/env python3
"""C4 (10 000-sphere stress scene) through the grid traversal: time, walk statistics, image hash.
    python profiles/grid_bench.py [--s 16] [--reps 3] [--ns 10000] [--stats]
The hash (sha256 of the u8 frame) must not change when the traversal is tuned: the image is
bit-identical to the brute-force one (tests/test_gpu_parity.py checks that at small sizes)."""
import argparse, hashlib, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ascendpathtracing_amd as apt
from ascendpathtracing_amd import gen_data, render

ap = argparse.ArgumentParser()
ap.add_argument("--s", type=int, default=16)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--ns", type=int, default=10000)
ap.add_argument("--depth", type=int, default=8)
ap.add_argument("--stats", action="store_true")
ap.add_argument("--retire", action="store_true")
ap.add_argument("--refill", type=int, default=0, help="apt_set_refill_lanes: waiting lanes that trigger the per-segment block of the queue kernel's grid form")
args = ap.parse_args()
if args.refill:
    render.set_refill_lanes(args.refill)
scene_h = gen_data.gen_scene(args.ns, seed=1)
scene = torch.from_numpy(scene_h).cuda()
t0 = time.time()
grid = torch.from_numpy(gen_data.build_grid(scene_h, args.ns).view("int32")).cuda()
t_grid = time.time() - t0
t0 = time.time()
dgrid = gen_data.build_grid_device(scene, args.ns); torch.cuda.synchronize()
t_first = time.time() - t0
t0 = time.time()
dgrid = gen_data.build_grid_device(scene, args.ns); torch.cuda.synchronize()
t_dev = time.time() - t0
assert torch.equal(dgrid, grid)
p = apt.make_params(1920, 1080, args.s, depth=args.depth, num_spheres=args.ns, accel=grid.data_ptr(),
                    flags=(apt.APT_FLAG_RETIRE if args.retire else 0) | gen_data.grid_flags(grid, args.ns))
fb, u8 = render.render_frame(p, scene)
torch.cuda.synchronize()
best = 1e9
for _ in range(args.reps):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); fb, u8 = render.render_frame(p, scene); b.record(); torch.cuda.synchronize()
    best = min(best, a.elapsed_time(b))
seg = p.num_paths * args.depth
out = {"refill": args.refill, "ns": args.ns, "S": args.s, "depth": args.depth, "ms": round(best, 3), "nominal_gray_per_s": round(seg / best / 1e6, 2),
       "sha256_u8": hashlib.sha256(u8.cpu().numpy().tobytes()).hexdigest()[:16],
       "grid_build_host_s": round(t_grid, 4), "grid_build_device_s": round(t_dev, 4), "grid_build_device_first_call_s": round(t_first, 4),
       "grid_bytes": int(grid.numel() * 4)}
if args.stats:
    with render.TraceCounter() as tc:
        render.render_frame(p, scene)
    traced, cells, tests = tc.stats
    out['brute_force_lanes'] = int(tc.buf[3].item())
    out.update(traced=traced, cells_per_segment=round(cells / max(traced, 1), 2), tests_per_segment=round(tests / max(traced, 1), 2))
print(json.dumps(out))
