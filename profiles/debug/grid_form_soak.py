#!/usr/bin/env python3
"""Soak of the sample-queue kernel's grid form: random scenes inside the reference camera's view (random sphere counts, radius
laws, clusters, coincident copies with different albedo, occasional large members), random frame sizes / sample counts / depths /
modes / flags; the frame and the traced-segment count must equal the brute-force traversal's (no accel) bit for bit.
    python profiles/debug/grid_form_soak.py [--n 200] [--seed 1] > gpurun_out/grid_form_soak.jsonl"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import ascendpathtracing_amd as apt
from ascendpathtracing_amd import gen_data, render

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=200)
ap.add_argument("--seed", type=int, default=1)
args = ap.parse_args()
rng = np.random.RandomState(args.seed)
bad = 0
t0 = time.time()
for it in range(args.n):
    ns = int(rng.choice([9, 20, 60, 200, 700, 2500]))
    scene = gen_data.gen_scene(ns, seed=int(rng.randint(1 << 30))).copy()
    tab = scene[:10 * ns].reshape(10, ns)
    small = np.arange(6, ns - 1)
    kind = rng.randint(5)
    if kind == 1 and len(small) > 8:        # tight clusters
        for _ in range(int(rng.randint(1, 4))):
            idx = rng.choice(small, size=min(len(small), int(rng.randint(4, 200))), replace=False)
            c = np.array([rng.uniform(10, 90), rng.uniform(5, 75), rng.uniform(20, 160)], dtype=np.float32)
            tab[1:4, idx] = c[:, None] + rng.normal(0, rng.choice([0.0, 0.01, 1.0]), size=(3, len(idx))).astype(np.float32)
    elif kind == 2 and len(small) > 8:      # coincident copies (exact ties), different albedo
        for _ in range(int(rng.randint(1, 5))):
            idx = rng.choice(small, size=int(rng.randint(2, 8)), replace=False)
            tab[0:4, idx] = tab[0:4, idx[:1]]
            tab[0, idx] = rng.uniform(1.0, 4.0) ** 2
    elif kind == 3:                         # radius law: many tiny or a few big ones
        tab[0, small] = (10.0 ** rng.uniform(-1.5, 0.7, size=len(small))).astype(np.float32) ** 2
    elif kind == 4 and len(small) > 3:      # symmetric pairs about the camera axis x = 50 (equal roots for axis-parallel rays)
        idx = rng.choice(small, size=2 * (min(len(small), 40) // 2), replace=False)
        a, b = idx[::2], idx[1::2]
        tab[0:4, b] = tab[0:4, a]
        tab[1, b] = 100.0 - tab[1, a]
    d_scene = torch.from_numpy(scene).cuda()
    hgrid = gen_data.build_grid(scene, ns)
    grid = torch.from_numpy(hgrid.view(np.int32)).cuda()
    w, h = int(rng.randint(3, 20)), int(rng.randint(2, 14))
    s_ = int(rng.choice([8, 9, 16, 20, 33, 64, 136]))
    depth = int(rng.choice([1, 2, 5, 8, 12]))
    mode = int(rng.randint(2))
    flags = int(rng.choice([0, apt.APT_FLAG_RETIRE, apt.APT_FLAG_RR, apt.APT_FLAG_RETIRE | apt.APT_FLAG_RR]))
    p = apt.make_params(w, h, s_, depth=depth, num_spheres=ns, seed=int(rng.randint(1 << 30)), mode=mode, flags=flags)
    with render.TraceCounter() as tb:
        fb_b, u8_b = render.render_frame(p, d_scene)
    with render.TraceCounter() as tq:
        # every other scene with APT_FLAG_GRID_SLOTS (round 5: the frame is then ONE launch, the grid form alone)
        vouch = gen_data.grid_flags(hgrid, ns) if it % 2 else 0
        fb_q, u8_q = render.render_frame(p.copy(accel=grid.data_ptr(), flags=flags | vouch), d_scene)
    torch.cuda.synchronize()
    ok = bool(torch.equal(fb_q.view(torch.int32), fb_b.view(torch.int32)) and torch.equal(u8_q, u8_b) and tq.value == tb.value)
    if not ok:
        bad += 1
        print(json.dumps({"FAIL": it, "ns": ns, "kind": int(kind), "w": w, "h": h, "S": s_, "depth": depth, "mode": mode, "flags": flags,
                          "traced": [tq.value, tb.value]}), flush=True)
    if it % 20 == 19:
        print(json.dumps({"done": it + 1, "failed": bad, "seconds": round(time.time() - t0, 1)}), flush=True)
print(json.dumps({"scenes": args.n, "seed": args.seed, "failed": bad, "seconds": round(time.time() - t0, 1)}))
sys.exit(1 if bad else 0)
