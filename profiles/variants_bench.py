#!/usr/bin/env python3
"""A/B timing of builds of librender_mi355x.so on one MI355X (run through gpurun):

    python profiles/variants_bench.py name=path/to/lib.so [name=path ...] [--reps 7]

Per library, in a child process (one HIP runtime image each), through raw ctypes on the entry points every
build has (apt_default_params, render_frame): the C2 frame (1920x1080, S=64, depth 8, K-mode, no flags) timed
with HIP events via torch, the same at depth 0 (ray-generate + accumulation only), with APT_FLAG_RETIRE, and the
frame's chunk hashes against tests/golden/fullsize_hashes.json (oracle-made).  One JSON line per library."""
import argparse, ctypes, hashlib, json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(name, path, reps, full_c4=False, only=""):
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    from ascendpathtracing_amd._lib import RenderParams
    lib = ctypes.CDLL(path)
    lib.apt_default_params.restype = None
    lib.apt_last_error.restype = ctypes.c_char_p
    sph_h = np.zeros(128, dtype=np.float32)
    assert lib.apt_gen_spheres_host(sph_h.ctypes.data_as(ctypes.c_void_p)) == 0
    sph = torch.from_numpy(sph_h).cuda()
    W, H, S = 1920, 1080, 64
    npix = W * H
    fb = torch.empty((3, npix), dtype=torch.float32, device="cuda")
    u8 = torch.empty((npix, 3), dtype=torch.uint8, device="cuda")

    def params(depth, flags=0, mode=0):
        p = RenderParams()
        lib.apt_default_params(ctypes.byref(p))
        p.width, p.height, p.samples, p.depth, p.flags, p.mode, p.seed = W, H, S, depth, flags, mode, 0
        return p

    def run(p, scene=None):
        rc = lib.render_frame(ctypes.byref(p), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), ctypes.c_void_p((sph if scene is None else scene).data_ptr()),
                              ctypes.c_uint64(0), ctypes.c_uint64(npix), ctypes.c_void_p(fb.data_ptr()), ctypes.c_void_p(u8.data_ptr()))
        assert rc == 0, lib.apt_last_error()

    def run_with(p, scene):
        run(p, scene)
        return u8

    def timeit(p, scene=None):
        run(p, scene); run(p, scene); torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for a, b in ev:
            a.record(); run(p, scene); b.record()
        torch.cuda.synchronize()
        t = sorted(a.elapsed_time(b) for a, b in ev)
        return round(t[0], 3), round(t[len(t) // 2], 3)

    out = {"lib": name}
    if only == "c4":
        c4_cases(lib, np, torch, hashlib, params, timeit, run_with, out, full_c4)
        print(json.dumps(out), flush=True)
        return
    if only == "queue":                       # the sample-queue kernel's cases only (+ the frame check)
        out["c2_retire_ms_min_med"] = timeit(params(8, flags=1))
        out["c5_d32_retire_ms_min_med"] = timeit(params(32, flags=1))
        out["c5_d32_rr_retire_ms_min_med"] = timeit(params(32, flags=3))
        out["c2_omode_retire_ms_min_med"] = timeit(params(8, flags=1, mode=1))
        with open(os.path.join(ROOT, "tests", "golden", "fullsize_hashes.json")) as f:
            case = json.load(f)["cases"]["C2"]
        run(params(8, flags=1)); torch.cuda.synchronize()
        fbh, u8h = fb.cpu().numpy(), u8.cpu().numpy()
        out["c2_retire_frame_equals_oracle"] = bool(all(
            hashlib.sha256(np.ascontiguousarray(fbh[:, b:b + c]).tobytes()).hexdigest() == case["fb_sha256"][k] and
            hashlib.sha256(np.ascontiguousarray(u8h[b:b + c]).tobytes()).hexdigest() == case["u8_sha256"][k]
            for k, (b, c) in enumerate(case["ranges"])))
        print(json.dumps(out), flush=True)
        return
    out["c2_ms_min_med"] = timeit(params(8))
    with open(os.path.join(ROOT, "tests", "golden", "fullsize_hashes.json")) as f:
        case = json.load(f)["cases"]["C2"]
    run(params(8)); torch.cuda.synchronize()
    fbh, u8h = fb.cpu().numpy(), u8.cpu().numpy()
    ok = all(hashlib.sha256(np.ascontiguousarray(fbh[:, b:b + c]).tobytes()).hexdigest() == case["fb_sha256"][k] and
             hashlib.sha256(np.ascontiguousarray(u8h[b:b + c]).tobytes()).hexdigest() == case["u8_sha256"][k]
             for k, (b, c) in enumerate(case["ranges"]))
    out["c2_frame_equals_oracle"] = bool(ok)
    out["depth0_ms_min_med"] = timeit(params(0))
    out["c2_retire_ms_min_med"] = timeit(params(8, flags=1))
    out["c2_omode_ms_min_med"] = timeit(params(8, mode=1))
    out["c5_d32_ms_min_med"] = timeit(params(32))
    out["c5_d32_retire_ms_min_med"] = timeit(params(32, flags=1))
    out["c5_d32_rr_retire_ms_min_med"] = timeit(params(32, flags=3))
    out["c5_d32_rr_ms_min_med"] = timeit(params(32, flags=2))      # one path per lane, no queue
    out["c2_omode_retire_ms_min_med"] = timeit(params(8, flags=1, mode=1))
    c4_cases(lib, np, torch, hashlib, params, timeit, run_with, out, full_c4)
    print(json.dumps(out), flush=True)


def c4_cases(lib, np, torch, hashlib, params, timeit, run_with, out, full_c4):
    """C4: the 10 000-sphere scene behind the uniform grid (sample-queue kernel, grid form) at 64 spp (and 256 with --full-c4), and the
    frame's hash (must not change)."""
    ns = 10000
    nfl = ctypes.c_size_t(0)
    assert lib.apt_gen_scene_host(ctypes.c_uint32(ns), ctypes.c_uint64(1), None, ctypes.byref(nfl)) == 0
    scene_h = np.zeros(nfl.value, dtype=np.float32)
    assert lib.apt_gen_scene_host(ctypes.c_uint32(ns), ctypes.c_uint64(1), scene_h.ctypes.data_as(ctypes.c_void_p), ctypes.byref(nfl)) == 0
    nby = ctypes.c_size_t(0)
    assert lib.apt_build_grid_host(scene_h.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint32(ns), None, ctypes.byref(nby)) == 0
    grid_h = np.zeros(nby.value // 4, dtype=np.uint32)
    assert lib.apt_build_grid_host(scene_h.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint32(ns), grid_h.ctypes.data_as(ctypes.c_void_p), ctypes.byref(nby)) == 0
    scene, grid = torch.from_numpy(scene_h).cuda(), torch.from_numpy(grid_h.view(np.int32)).cuda()

    def params4(s4, flags=0):
        p = params(8, flags=flags)
        p.samples, p.num_spheres, p.light_index, p.accel = s4, ns, ns - 1, grid.data_ptr()
        return p
    for s4, key in ((16, "c4_grid_s16"), (64, "c4_grid_s64")):
        if s4 == 64 and not full_c4:
            continue
        out[key + "_ms_min_med"] = timeit(params4(s4), scene)
        out[key + "_retire_ms_min_med"] = timeit(params4(s4, flags=1), scene)
    u8 = run_with(params4(16), scene); torch.cuda.synchronize()
    out["c4_grid_s16_u8_sha"] = hashlib.sha256(u8.cpu().numpy().tobytes()).hexdigest()[:16]


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(sys.argv[2], sys.argv[3], int(sys.argv[4]), full_c4=len(sys.argv) > 5 and sys.argv[5] == "1", only=sys.argv[6] if len(sys.argv) > 6 else "")
        sys.exit(0)
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--reps", type=int, default=7)
    ap.add_argument("--full-c4", action="store_true", help="also time C4 at its full 256 spp (S = 64: ~0.2 s per frame)")
    ap.add_argument("--only", default="", choices=["", "c4", "queue"], help="c4: only the grid cases; queue: only the sample-queue kernel's cases")
    a = ap.parse_args()
    for spec in a.libs:
        name, path = spec.split("=", 1)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", name, os.path.abspath(path), str(a.reps), "1" if a.full_c4 else "0", a.only])
        if r.returncode:
            print(json.dumps({"lib": name, "error": r.returncode}), flush=True)
