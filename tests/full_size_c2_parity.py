#!/usr/bin/env python3
"""Full-size bitwise check of BASELINE config C2 (1920x1080, 256 spp, 8 bounces) through the
reference's own pipeline, GPU vs CPU restatement, every one of the 530,841,600 paths:
  gen_rays (MT19937, seed 0)  : device (checkpointed stream) vs the library's sequential host generator
  render (O-mode = NumPy oracle arithmetic, and K-mode)       : device vs oracle.render_paths (all host threads)
  decode_color                                               : device vs oracle.decode_color
  fused frame kernel (the benchmarked one, counter-RNG rays)  : device vs oracle.render_frame, every pixel
Checker code (oracle) is used here as the checker only.  ~70 s on 128 host threads, ~40 GB of host memory.
Run directly (python tests/full_size_c2_parity.py) or through pytest with APT_FULL_PARITY=1
(tests/test_gpu_parity.py::test_full_size_c2_parity); the round-1 log is profiles/history/r01_c2_full_parity.log."""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ascendpathtracing_amd as apt
from ascendpathtracing_amd import gen_data, render
from oracle import oracle

W, H, S, D = 1920, 1080, 64, 8
if len(sys.argv) > 1:
    W, H, S = (int(x) for x in sys.argv[1:4])
t0 = time.time()
def log(msg): print(f"[{time.time() - t0:7.1f}s] {msg}", flush=True)
p = apt.make_params(W, H, S, depth=D, mode=apt.APT_MODE_ORACLE)
n = p.num_paths
log(f"config {W}x{H} S={S} D={D}: {n} paths, {n * D} segments")
sph_h = gen_data.gen_spheres(); sph = torch.from_numpy(sph_h).cuda()
rays_d = gen_data.gen_rays_device(W, H, S, seed=0)
torch.cuda.synchronize(); log("device gen_rays done")
rays_h = gen_data.gen_rays(W, H, S, seed=0); log("host gen_rays done (sequential MT19937)")
rays_dh = rays_d.cpu().numpy()
ok_rays = np.array_equal(rays_dh.view(np.uint32), rays_h.view(np.uint32))
log(f"rays bitwise equal: {ok_rays}  sha256={hashlib.sha256(rays_h.tobytes()).hexdigest()[:16]}")
del rays_dh
results = {"rays": ok_rays}
threads = oracle.max_threads()
for mode, omode, name in ((apt.APT_MODE_ORACLE, oracle.MODE_O, "O"), (apt.APT_MODE_KERNEL, oracle.MODE_K, "K")):
    pm = p.copy(mode=mode)
    col_d = render.render_paths(pm, rays_d.reshape(-1), sph); torch.cuda.synchronize()
    col_h, traced = oracle.render_paths(oracle.make_params(W, H, S, depth=D, mode=omode), rays_h, sph_h, threads=threads)
    log(f"{name}-mode: oracle traced {traced} segments on {threads} threads")
    ok = np.array_equal(col_d.cpu().numpy().view(np.uint32), col_h.view(np.uint32))
    log(f"{name}-mode colours bitwise equal: {ok}  sha256={hashlib.sha256(col_h.tobytes()).hexdigest()[:16]}")
    results[f"colors_{name}"] = ok
    if name == "O":
        fb_d, u8_d = render.decode_color_device(pm, col_d); torch.cuda.synchronize()
        pre, fb_h, u8_h = oracle.decode_color(col_h, W, H, S)
        okd = np.array_equal(fb_d.cpu().numpy().view(np.uint32), fb_h.view(np.uint32)) and np.array_equal(u8_d.cpu().numpy(), u8_h)
        log(f"decode_color bitwise equal (float frame + 8-bit): {okd}")
        results["decode"] = okd
        # the fused frame kernel traces counter-RNG rays, not these; compare the two images statistically
        fb_f, _ = render.render_frame(apt.make_params(W, H, S, depth=D, mode=mode, seed=0), sph); torch.cuda.synchronize()
        rms = float(torch.sqrt(((fb_f - fb_d) ** 2).mean()))
        log(f"RMS between this frame and the fused counter-RNG frame (different random jitter): {rms:.4f}")
    del col_d, col_h
# the headline kernel itself: the fused frame (counter-RNG rays generated on the device, 4*S samples accumulated
# on the device) against the CPU restatement's frame, every pixel, with and without result-preserving retirement
del rays_d, rays_h
torch.cuda.empty_cache()
pk = apt.make_params(W, H, S, depth=D, mode=apt.APT_MODE_KERNEL, seed=0)
fb_w, u8_w, _, traced = oracle.render_frame(oracle.make_params(W, H, S, depth=D, mode=oracle.MODE_K, seed=0), sph_h, threads=threads)
log(f"oracle frame done ({traced} segments)")
for flags, name in ((0, "frame_K"), (apt.APT_FLAG_RETIRE, "frame_K_retire")):
    fb_g, u8_g = render.render_frame(pk.copy(flags=flags), sph); torch.cuda.synchronize()
    ok = np.array_equal(fb_g.cpu().numpy().view(np.uint32), fb_w.view(np.uint32)) and np.array_equal(u8_g.cpu().numpy(), u8_w)
    log(f"fused frame ({name}) bitwise equal to the CPU restatement, all {W * H} pixels: {ok}  sha256(u8)={hashlib.sha256(u8_w.tobytes()).hexdigest()[:16]}")
    results[name] = ok
log("RESULT " + ("PASS" if all(results.values()) else "FAIL") + " " + str(results))
sys.exit(0 if all(results.values()) else 1)
