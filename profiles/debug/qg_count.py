#!/usr/bin/env python3
"""Traced-segment counter of one frame by four routes: the oracle, the sample-queue kernel's grid form, the nested item walk
(the "grid_walk" knob) and the brute-force traversal over LDS tiles, for sample counts with and without an n % 8 tail.  (Found in
round 3: with a tail the frame kernels counted the re-traced first tail sample of the lanes past the tail; fixed in pt_kernels.h.)
    python profiles/debug/qg_count.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import ascendpathtracing_amd as apt
from ascendpathtracing_amd import gen_data, render
from oracle import oracle
ns = 700
scene = gen_data.gen_scene(ns, seed=3)
d_scene = torch.from_numpy(scene).cuda()
grid = torch.from_numpy(gen_data.build_grid(scene, ns).view(np.int32)).cuda()
for (w, h, s_, depth) in ((12, 8, 8, 6), (9, 7, 20, 5), (9, 7, 16, 5), (9, 7, 24, 5)):
    p = apt.make_params(w, h, s_, depth=depth, num_spheres=ns, seed=11, flags=apt.APT_FLAG_RETIRE)
    op = oracle.make_params(w, h, s_, depth=depth, num_spheres=ns, seed=11, flags=oracle.FLAG_RETIRE)
    _, _, _, tw = oracle.render_frame(op, scene, threads=8)
    with render.TraceCounter() as tq:
        render.render_frame(p.copy(accel=grid.data_ptr()), d_scene)
    with render.debug_knob("grid_walk", 1), render.TraceCounter() as tn:
        render.render_frame(p.copy(accel=grid.data_ptr()), d_scene)
    with render.TraceCounter() as tb:
        render.render_frame(p, d_scene)
    print(w, h, s_, depth, "oracle", int(tw), "queue", tq.value, "nested", tn.value, "tiles", tb.value, flush=True)
