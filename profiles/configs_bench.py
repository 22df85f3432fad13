#!/usr/bin/env python3
"""Times the other BASELINE.json configurations on one MI355X (bench.py covers configs[1] = C2):
C1 (256x256, 4 spp, D4, buffer mode incl. the exact reference boundary), C4 (10k-sphere scene,
LDS-staged traversal), C5 (32 bounces, with and without result-preserving retirement).  One
JSON line per case.   python profiles/configs_bench.py [--spp-c4 S]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ascendpathtracing_amd as apt
from ascendpathtracing_amd import gen_data, render

ap = argparse.ArgumentParser()
ap.add_argument("--spp-c4", type=int, default=8, help="S for the 10k-sphere case (64 = full 256 spp, ~8x longer)")
ap.add_argument("--reps", type=int, default=3)
args = ap.parse_args()


def timeit(fn, reps):
    fn(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return min(a.elapsed_time(b) for a, b in ev)


def report(name, ms, segments, ns, extra=None, culled=False):
    """segments = nominal N*D.  The roofline figures (pair tests, flops, fraction of the vector peak) are computed from the
    segments actually TRACED when the case retires paths (extra["segments_traced"]): a fraction above 1 is no roofline figure."""
    out = {"case": name, "kernel_ms": round(ms, 3), "segments_nominal": segments, "Mray_per_s_nominal": round(segments / ms / 1e3, 1)}
    out.update(extra or {})
    traced = out.get("segments_traced", segments)
    out["Mray_per_s_traced"] = round(traced / ms / 1e3, 1)
    if not culled:   # brute force: every ray/sphere pair of a traced segment is evaluated.  Ns = 8: all 20 flops of F(Ns) per pair
        # run (branch-free).  LDS tiles: the 4 sqrt / root flops run only where some lane of the wave can hit, so the
        # EXECUTED work is priced at the 16 discriminant flops per pair (a lower bound).
        flops = traced * ((20 if ns == 8 else 16) * ns + 33)
        out.update({"pair_tests_per_s": round(traced * ns / ms * 1e3, 0), "achieved_TFLOPs": round(flops / ms / 1e9, 3),
                    "frac_of_157.3": round(flops / ms / 1e9 / 157.3, 4)})
        assert out["frac_of_157.3"] <= 1.0
    print(json.dumps(out), flush=True)


sph8 = torch.from_numpy(gen_data.gen_spheres()).cuda()
# C1: buffer mode, the exact render_do boundary (rays from gen_rays, MT19937)
rays = torch.from_numpy(gen_data.gen_rays(256, 256, 1, seed=0).ravel()).cuda()
for mode, name in ((apt.APT_MODE_KERNEL, "K"), (apt.APT_MODE_ORACLE, "O")):
    p = apt.make_params(256, 256, 1, depth=4, mode=mode)
    colors = torch.empty(3 * p.num_paths, device="cuda")
    ms = timeit(lambda: render.render_do_ex(p, None, rays, sph8, colors), args.reps)
    report(f"C1 256x256 4spp D4 buffer mode {name}-mode", ms, p.num_paths * 4, 8)
# C1 with host buffers: pinned rays H2D + render_do + colours D2H (what src/main.cpp:68-77 does around the launch)
p = apt.make_params(256, 256, 1, depth=4)
rays_h = torch.from_numpy(gen_data.gen_rays(256, 256, 1, seed=0).ravel().copy()).pin_memory()
cols_h = torch.empty(3 * p.num_paths, dtype=torch.float32).pin_memory()
rays_d = torch.empty_like(rays_h, device="cuda"); cols_d = torch.empty(3 * p.num_paths, device="cuda")
def roundtrip():
    rays_d.copy_(rays_h, non_blocking=True)
    render.render_do_ex(p, None, rays_d, sph8, cols_d)
    cols_h.copy_(cols_d, non_blocking=True)
ms = timeit(roundtrip, 5)
report("C1 256x256 4spp D4 buffer mode incl. PCIe (6.3 MB H2D + 3.1 MB D2H, pinned)", ms, p.num_paths * 4, 8)
# C2 in O-mode (the NumPy oracle's arithmetic: float64-accumulated dot products in the shading step)
p = apt.make_params(1920, 1080, 64, depth=8, mode=apt.APT_MODE_ORACLE)
ms = timeit(lambda: render.render_frame(p, sph8), args.reps)
report("C2 D8 1080p 256spp O-mode", ms, p.num_paths * 8, 8)
# C2 / C5: frame mode
for d, flags, name in ((8, 0, "C2 D8"), (8, apt.APT_FLAG_RETIRE, "C2 D8 retire"), (32, 0, "C5 D32"),
                       (32, apt.APT_FLAG_RETIRE, "C5 D32 retire"),
                       (32, apt.APT_FLAG_RETIRE | apt.APT_FLAG_RR, "C5 D32 Russian roulette (rr_start 3) + retire")):
    p = apt.make_params(1920, 1080, 64, depth=d, flags=flags)
    with render.TraceCounter() as tc:
        ms = timeit(lambda: render.render_frame(p, sph8), 1)
    traced = tc.value // 2
    ms = timeit(lambda: render.render_frame(p, sph8), args.reps)
    report(f"{name} 1080p 256spp", ms, p.num_paths * d, 8, {"segments_traced": traced})
# C3 (4096x4096, 1024 spp, D8, 8 GPUs): the band one of the 8 ranks renders (2 097 152 pixels, 17.2 G segments)
p3 = apt.make_params(4096, 4096, 256, depth=8)
npix3 = 4096 * 4096 // 8
ms = timeit(lambda: render.render_frame(p3, sph8, 3 * npix3, npix3), 2)
report("C3 4096x4096 1024spp D8, one rank's band of 8", ms, npix3 * 4 * 256 * 8, 8)
# C2 through the reference's own pipeline, entirely on the device and bit-exact with it: MT19937 gen_rays ->
# render_do_ex on a [6][N] ray buffer (12.7 GB) -> colours [3][N] (6.4 GB) -> decode_color.  O-mode = NumPy oracle.
import time
t0 = time.time()
ck = torch.from_numpy(gen_data.mt19937_checkpoints(1920 * 1080 * 4 * 64, seed=0, stride=64).view("int32")).cuda()
t_ck = time.time() - t0
p = apt.make_params(1920, 1080, 64, depth=8, mode=apt.APT_MODE_ORACLE)
rays = gen_data.gen_rays_device(1920, 1080, 64, checkpoints=ck, stride=64).reshape(-1)
colors = torch.empty(3 * p.num_paths, device="cuda")
ms_gen = timeit(lambda: gen_data.gen_rays_device(1920, 1080, 64, checkpoints=ck, stride=64), 2)
ms_ren = timeit(lambda: render.render_do_ex(p, None, rays, sph8, colors), 2)
ms_dec = timeit(lambda: render.decode_color_device(p, colors), 2)
pq = p.copy(flags=apt.APT_FLAG_RETIRE)
ms_ren_q = timeit(lambda: render.render_do_ex(pq, None, rays, sph8, colors), 2)
report("C2 exact reference pipeline on device (MT19937 rays, buffer mode, O-mode)", ms_gen + ms_ren + ms_dec,
       p.num_paths * 8, 8, {"gen_rays_ms": round(ms_gen, 2), "render_ms": round(ms_ren, 2), "decode_ms": round(ms_dec, 2),
                            "render_ms_with_compaction": round(ms_ren_q, 2),
                            "host_checkpoint_seconds_once": round(t_ck, 2), "hbm_GB": round((9 * 4 * p.num_paths) / 1e9, 1)})
del rays, colors, ck
torch.cuda.empty_cache()
# the same pipeline in one kernel (round 3: apt_render_frame_mt)
import numpy as np
t0 = time.time()
ckg, g_lo = render.mt_group_checkpoints(1920, 1080, 64, seed=0)
t_ck = time.time() - t0
ckd = (torch.from_numpy(ckg.view(np.int32)).cuda(), g_lo)
for mode, name in ((apt.APT_MODE_ORACLE, "O-mode"), (apt.APT_MODE_KERNEL, "K-mode")):
    ms = timeit(lambda: render.render_reference_frame_fused(1920, 1080, 64, depth=8, seed=0, spheres=sph8, checkpoints=ckd, mode=mode), 3)
    report(f"C2 exact reference pipeline FUSED into one kernel (MT19937 in LDS, {name})", ms, p.num_paths * 8, 8,
           {"host_checkpoint_seconds_once": round(t_ck, 2), "hbm_GB": round((ckg.nbytes + 15 * 1920 * 1080) / 1e9, 3)})
del ckd
torch.cuda.empty_cache()
# C4: 10k spheres
scene = torch.from_numpy(gen_data.gen_scene(10000, seed=1)).cuda()
for flags, name in ((0, ""), (apt.APT_FLAG_RETIRE, " retire")):
    p = apt.make_params(1920, 1080, args.spp_c4, depth=8, num_spheres=10000, flags=flags)
    ms = timeit(lambda: render.render_frame(p, scene), 1)
    report(f"C4 10k spheres 1080p {4 * args.spp_c4}spp D8{name}", ms, p.num_paths * 8, 10000)
# C4 through the host-built grid (bit-identical image), at the full 256 spp
t0 = time.time()
grid = torch.from_numpy(gen_data.build_grid(scene.cpu().numpy(), 10000).view("int32")).cuda()
t_grid = time.time() - t0
for flags, name in ((0, ""), (apt.APT_FLAG_RETIRE, " retire")):
    p = apt.make_params(1920, 1080, 64, depth=8, num_spheres=10000, flags=flags | gen_data.grid_flags(grid, 10000), accel=grid.data_ptr())
    ms = timeit(lambda: render.render_frame(p, scene), 2)
    report(f"C4 10k spheres 1080p 256spp D8 grid traversal{name}", ms, p.num_paths * 8, 10000,
           {"grid_build_host_seconds": round(t_grid, 3), "grid_bytes": int(grid.numel() * 4)}, culled=True)
