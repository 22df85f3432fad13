#!/usr/bin/env python3
"""Writes tests/golden/fullsize_hashes.json: SHA-256 of the float and 8-bit frames (or of stated pixel
ranges of them) that THE ORACLE (oracle/pt_oracle.c, the CPU restatement) produces for the BASELINE.json
configurations at their full sizes.  Run in the build container (8 vCPU: ~4 min):

    python tests/golden/make_fullsize_hashes.py [--threads 8] [--only C2,C5_rr]

The always-on `-m gpu` tests in tests/test_fullsize_hashes.py render the same frames on the HIP path and
compare the hashes, so every BASELINE configuration is checked at its real size on every driver run
without shipping 25 MB frames.  Only the oracle is run here (nothing from /root/reference): the oracle
itself is pinned to the reference by tests/test_oracle_golden.py.

Per case: the parameters, the pixel ranges hashed (x-major pixel index q = i*H + j, as render_frame takes
them), and per range sha256(fb float32 [3][count] bytes) / sha256(u8 [count][3] bytes).
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "fullsize_hashes.json")


def chunks(npix, n):
    base, extra = divmod(npix, n)
    out, b = [], 0
    for r in range(n):
        c = base + (1 if r < extra else 0)
        out.append([b, c])
        b += c
    return out


def band_edges(npix, world, edge):
    """First and last `edge` pixels of each of the `world` contiguous bands (dist.split_range)."""
    out = []
    for b, c in chunks(npix, world):
        out.append([b, edge])
        out.append([b + c - edge, edge])
    return out


def spread(npix, n, count):
    return [[(k * 2654435761) % (npix - count), count] for k in range(n)]


# name -> dict(params..., ranges).  scene: "demo" = gen_spheres(); ("scene", ns, seed) = gen_scene.
CASES = {
    # BASELINE configs[1]: the headline workload, whole frame, K-mode (the benchmarked arithmetic)
    "C2": dict(w=1920, h=1080, s=64, depth=8, mode="K", flags=0, seed=0, scene="demo", ranges=chunks(1920 * 1080, 16)),
    # the same frame in O-mode (the NumPy oracle's arithmetic) on 8 spread ranges of 16384 pixels
    "C2_omode": dict(w=1920, h=1080, s=64, depth=8, mode="O", flags=0, seed=0, scene="demo",
                     ranges=spread(1920 * 1080, 8, 16384)),
    # BASELINE configs[4] as named: 32 bounces with Russian-roulette termination (+ retirement), whole frame
    "C5_rr": dict(w=1920, h=1080, s=64, depth=32, mode="K", flags=oracle.FLAG_RR | oracle.FLAG_RETIRE, seed=0, rr_start=0,
                  scene="demo", ranges=chunks(1920 * 1080, 16)),
    # configs[4] without roulette (every path to depth 32; retirement is result preserving), whole frame
    "C5_full": dict(w=1920, h=1080, s=64, depth=32, mode="K", flags=0, seed=0, scene="demo",
                    ranges=chunks(1920 * 1080, 16)),
    # BASELINE configs[2]: 4096x4096, 1024 spp, 8 bounces: first and last image column (4096 pixels) of each of
    # the 8 bands the ranks render (band r = pixels [r*2^21, (r+1)*2^21))
    "C3_bands": dict(w=4096, h=4096, s=256, depth=8, mode="K", flags=0, seed=0, scene="demo",
                     ranges=band_edges(4096 * 4096, 8, 4096)),
    # BASELINE configs[3]: 10 000-sphere scene.  A whole small frame, brute force on the CPU ...
    "C4_crop": dict(w=480, h=270, s=1, depth=8, mode="K", flags=0, seed=0, scene=("scene", 10000, 1),
                    ranges=chunks(480 * 270, 4)),
    # ... and 8 spread ranges of 256 pixels of the real 1080p / 256 spp frame
    "C4_1080p": dict(w=1920, h=1080, s=64, depth=8, mode="K", flags=0, seed=0, scene=("scene", 10000, 1),
                     ranges=spread(1920 * 1080, 8, 256)),
    # Round 3: interior pixels of the big frames.  One WHOLE rank band of C3 (band 5 of 8: 2^21 pixels,
    # 2.1e9 paths, in 16 chunks), 64 spread ranges of 1024 pixels over the whole C3 frame (every band's interior), and 64 more
    # spread ranges of the C4 1080p frame.
    "C3_band5_whole": dict(w=4096, h=4096, s=256, depth=8, mode="K", flags=0, seed=0, scene="demo",
                           ranges=[[5 * (1 << 21) + b, c] for b, c in chunks(1 << 21, 16)]),
    "C3_spread": dict(w=4096, h=4096, s=256, depth=8, mode="K", flags=0, seed=0, scene="demo",
                      ranges=[[(k * 2654435761 + 12345) % (4096 * 4096 - 1024), 1024] for k in range(64)]),
    "C4_1080p_spread": dict(w=1920, h=1080, s=64, depth=8, mode="K", flags=0, seed=0, scene=("scene", 10000, 1),
                            ranges=[[(k * 40503 * 4099 + 977) % (1920 * 1080 - 128), 128] for k in range(64)]),
    # Round 4 (the C4 frame was oracle-pinned on 10 240 of its 2 073 600 pixels): 384 more ranges of 256 pixels,
    # 98 304 pixels = 4.7 % of the frame, ~1.5 hours of brute force over 10 000 spheres on 6 threads
    "C4_1080p_spread2": dict(w=1920, h=1080, s=64, depth=8, mode="K", flags=0, seed=0, scene=("scene", 10000, 1),
                             ranges=[[(k * 2246822519 + 3266489917) % (1920 * 1080 - 256), 256] for k in range(384)]),
}


# The reference's EXACT pipeline (MT19937 gen_rays -> test_soa arithmetic (O-mode) -> decode_color) on pixel ranges of
# the big configurations: "mt_state" names the committed generator state the window starts from (None = from the seed).
MT_CASES = {
    # the WHOLE C2 frame through the reference's exact pipeline, in 16 chunks whose generator states chain
    "C2_mt_whole": dict(w=1920, h=1080, s=64, depth=8, ranges=chunks(1920 * 1080, 16), mt_state=None, chain=True),
    "C2_mt_first_band": dict(w=1920, h=1080, s=64, depth=8, ranges=[[0, 8192]], mt_state=None),
    "C3_mt_last_column": dict(w=4096, h=4096, s=256, depth=8, ranges=[[4096 * 4096 - 4096, 4096]], mt_state="mt19937_state_c3.npz"),
}


def run_mt_case(name, case, threads):
    w, h, s = case["w"], case["h"], case["s"]
    sph = oracle.gen_spheres()
    t0 = time.time()
    fb_sha, u8_sha = [], []
    chained = None
    for b, c in case["ranges"]:
        first_path, count = b * 4 * s, c * 4 * s
        state, blk = None, first_path // 156
        if case["mt_state"]:
            f = np.load(os.path.join(ROOT, "tests", "golden", case["mt_state"]))
            assert int(f["block"]) == blk, (int(f["block"]), blk)
            state = f["state"]
        if case.get("chain") and chained is not None:
            state = chained                                   # raw state of block blk, handed on by the previous chunk
        rays, _, chained = oracle.gen_rays_window(w, h, s, blk, first_path, count, seed=0, state_in=state, want_end=True)
        # the band as an image of c x 1 pixels: the render does not look at pixel coordinates, decode_color only at the grouping
        p = oracle.make_params(c, 1, s, depth=case["depth"], mode=oracle.MODE_O, flags=oracle.FLAG_RETIRE)
        colors, _ = oracle.render_paths(p, rays, sph, threads=threads)
        _, fb, u8 = oracle.decode_color(colors, c, 1, s)
        fb_sha.append(sha(fb))
        u8_sha.append(sha(u8))
    entry = dict(case)
    entry.update({"fb_sha256": fb_sha, "u8_sha256": u8_sha, "oracle_seconds": round(time.time() - t0, 1),
                  "pipeline": "oracle_gen_rays_window (MT19937, np.random.seed(0)) -> oracle_render_paths O-mode -> oracle_decode_color"})
    print(f"{name}: {sum(c for _, c in case['ranges'])} pixels, {entry['oracle_seconds']} s", flush=True)
    return entry


def scene_of(case):
    sc = case["scene"]
    if sc == "demo":
        return oracle.gen_spheres(), 8
    _, ns, seed = sc
    return oracle.gen_scene(ns, seed=seed), ns


def oracle_params(case, ns, retire=True):
    # FLAG_RETIRE is result preserving (tests/test_oracle_golden.py checks it): always on here to save CPU time
    flags = case["flags"] | (oracle.FLAG_RETIRE if retire else 0)
    return oracle.make_params(case["w"], case["h"], case["s"], depth=case["depth"], num_spheres=ns,
                              mode=oracle.MODE_O if case["mode"] == "O" else oracle.MODE_K, flags=flags,
                              seed=case["seed"], rr_start=case.get("rr_start", 0))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def run_case(name, case, threads):
    sph, ns = scene_of(case)
    p = oracle_params(case, ns)
    t0 = time.time()
    fb_sha, u8_sha, traced = [], [], 0
    for b, c in case["ranges"]:
        fb, u8, _, tr = oracle.render_frame(p, sph, pixel_begin=b, pixel_count=c, threads=threads)
        fb_sha.append(sha(fb))
        u8_sha.append(sha(u8))
        traced += tr
    entry = {k: v for k, v in case.items() if k != "scene"}
    entry["scene"] = case["scene"] if case["scene"] == "demo" else {"generator": "gen_scene", "num_spheres": case["scene"][1],
                                                                   "seed": case["scene"][2]}
    entry.update({"num_spheres": ns, "fb_sha256": fb_sha, "u8_sha256": u8_sha, "segments_traced_with_retirement": traced,
                  "oracle_seconds": round(time.time() - t0, 1)})
    print(f"{name}: {len(case['ranges'])} ranges, {sum(c for _, c in case['ranges'])} pixels, {traced} segments, "
          f"{entry['oracle_seconds']} s", flush=True)
    return entry


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=min(8, oracle.max_threads()))
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    out = {}
    if os.path.exists(OUT):
        with open(OUT) as f:
            out = json.load(f)
    names = [n for n in CASES if not args.only or n in args.only.split(",")]
    for n in names:
        out.setdefault("cases", {})[n] = run_case(n, CASES[n], args.threads)
    for n in [n for n in MT_CASES if not args.only or n in args.only.split(",")]:
        out.setdefault("mt_cases", {})[n] = run_mt_case(n, MT_CASES[n], args.threads)
    out["generator"] = "tests/golden/make_fullsize_hashes.py (oracle/pt_oracle.c, gcc -O2 -ffp-contract=off)"
    out["layout"] = "per range: sha256(float32 fb[3][count]) and sha256(uint8 u8[count][3]); ranges are [pixel_begin, pixel_count]"
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
