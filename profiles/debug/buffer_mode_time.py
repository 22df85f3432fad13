#!/usr/bin/env python3
"""render_do_ex (buffer mode, the drop-in boundary) on a C2-sized ray buffer: HIP-event time per launch, K- and O-mode, for the whole
range (>= 2^20 paths: two paths per lane) and for a range just below the threshold scaled up (one path per lane).
python profiles/debug/buffer_mode_time.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import ascendpathtracing_amd as apt
from ascendpathtracing_amd import gen_data, render

sph = torch.from_numpy(gen_data.gen_spheres()).cuda()
W, H, S, D = 1920, 1080, 64, 8
for mode, name in ((apt.APT_MODE_KERNEL, "K"), (apt.APT_MODE_ORACLE, "O")):
    p = apt.make_params(W, H, S, depth=D, mode=mode)
    rays = render.gen_rays_device(p).reshape(-1)
    colors = torch.empty(3 * p.num_paths, device="cuda")

    def timeit(fn, reps=3):
        fn(); torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for a, b in ev:
            a.record(); fn(); b.record()
        torch.cuda.synchronize()
        return min(a.elapsed_time(b) for a, b in ev)

    ms_all = timeit(lambda: render.render_do_ex(p, None, rays, sph, colors))
    c = (1 << 20) - 1                                       # below the two-path threshold: the one-path kernel
    chunks = 64
    ms_one = timeit(lambda: [render.render_do_ex(p.copy(path_begin=k * c, path_count=c), None, rays, sph, colors) for k in range(chunks)])
    seg = p.num_paths * D
    print(json.dumps({"mode": name, "whole_buffer_ms": round(ms_all, 3), "ps_per_segment": round(ms_all * 1e9 / seg, 3),
                      "one_path_kernel_ps_per_segment": round(ms_one * 1e9 / (chunks * c * D), 3),
                      "one_path_kernel_scaled_to_the_buffer_ms": round(ms_one / (chunks * c) * p.num_paths, 3)}), flush=True)
    del rays, colors
    torch.cuda.empty_cache()
