#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING the reference's own
NumPy oracle (scripts/gen_data.py, scripts/data_visualization.py) in this container.

This script is test infrastructure.  It is the only file in the repo that touches
/root/reference, and it only runs where /root/reference exists (never on the GPU box).
It imports the reference modules with importlib (no source is copied); the files it
writes are DATA ONLY: inputs (rays/spheres/colour buffers) and the outputs the
reference computed for them.

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz, golden.json

Reference call sites exercised:
    gen_rays      scripts/gen_data.py:21-75     (np.random.seed(0) as in :438)
    gen_spheres   scripts/gen_data.py:92-132
    test_soa      scripts/gen_data.py:246-429   (depth via module global bounceMax, :10)
    test_scene    scripts/gen_data.py:134-188
    decode_color  scripts/data_visualization.py:20-59 (module global `samples`, :7)
"""
import contextlib
import hashlib
import importlib.util
import io
import json
import os
import sys
import tempfile
import time
import warnings

import numpy as np

REF = "/root/reference/scripts"
HERE = os.path.dirname(os.path.abspath(__file__))


def _load(name):
    spec = importlib.util.spec_from_file_location("ref_" + name, os.path.join(REF, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def sha(path_or_bytes):
    if isinstance(path_or_bytes, (bytes, bytearray, memoryview)):
        return hashlib.sha256(path_or_bytes).hexdigest()
    with open(path_or_bytes, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()


def main():
    gd = _load("gen_data")
    dv = _load("data_visualization")
    warnings.simplefilter("ignore", RuntimeWarning)  # sqrt of negative discriminants, gen_data.py:227

    work = tempfile.mkdtemp(prefix="golden_")
    os.chdir(work)
    os.makedirs("input")
    os.makedirs("output")

    meta = {"numpy": np.__version__, "cases": {}, "sha_only": {}, "decode": {}}
    arrays = {}

    def run_case(w, h, s, depths, keep=True, scene=False):
        np.random.seed(0)  # gen_data.py:438
        with contextlib.redirect_stdout(io.StringIO()):
            rays = gd.gen_rays(w, h, s)
            spheres = gd.gen_spheres()
        key = f"{w}x{h}_s{s}"
        rays_bin = np.fromfile("input/rays.bin", dtype=np.float32)
        sph_bin = np.fromfile("input/spheres.bin", dtype=np.float32)
        entry = {"w": w, "h": h, "s": s, "n": int(rays_bin.size // 6),
                 "rays_sha256": sha("input/rays.bin"),
                 "spheres_sha256": sha("input/spheres.bin"), "depth": {}}
        if keep:
            arrays[f"{key}_rays"] = rays_bin
            arrays["spheres"] = sph_bin
        for d in depths:
            gd.bounceMax = d
            t0 = time.time()
            with contextlib.redirect_stdout(io.StringIO()):
                gd.test_soa(rays, spheres)
            dt = time.time() - t0
            out = np.fromfile("output/test_soa.bin", dtype=np.float32)
            entry["depth"][str(d)] = {"test_soa_sha256": sha("output/test_soa.bin"),
                                      "oracle_seconds": round(dt, 3)}
            if keep:
                arrays[f"{key}_d{d}_soa"] = out
        gd.bounceMax = 5
        if scene:
            with contextlib.redirect_stdout(io.StringIO()):
                gd.test_scene(rays, spheres)
            out = np.fromfile("output/test_scene.bin", dtype=np.float32)
            entry["test_scene_sha256"] = sha("output/test_scene.bin")
            if keep:
                arrays[f"{key}_scene"] = out
        (meta["cases"] if keep else meta["sha_only"])[key] = entry
        print(key, "done", flush=True)

    # --- small cases: inputs and outputs are committed -----------------------------------
    run_case(16, 16, 1, [1, 2, 3, 5, 8], scene=True)       # reference default size (common.h:4-6)
    run_case(16, 16, 2, [5])                               # S > 1 index layout
    run_case(32, 32, 1, [5, 32])                           # deep: ties / all-miss semantics
    run_case(64, 64, 1, [5])                               # the Appendix-C 64x64 case
    run_case(24, 16, 1, [5])                               # non-square camera (cx = w*.5135/h)
    # --- larger cases: only SHA-256 of inputs/outputs are committed -----------------------
    run_case(64, 64, 1, [8, 32], keep=False)
    run_case(256, 256, 1, [4], keep=False)                 # BASELINE config C1

    # --- decode_color / write_ppm (R14) ----------------------------------------------------
    class _NpProxy:
        """Stands in for the module-level `np` of data_visualization so that the float64
        image the reference hands to np.clip (data_visualization.py:54) can be recorded."""
        captured = None

        def __getattr__(self, name):
            return getattr(np, name)

        def clip(self, a, lo, hi):
            _NpProxy.captured = np.array(a, copy=True)
            return np.clip(a, lo, hi)

    dv.np = _NpProxy()

    def run_decode(name, w, h, s, color):
        color.astype(np.float32).tofile("output/color.bin")
        dv.samples = s                                     # data_visualization.py:22 reads the GLOBAL
        with contextlib.redirect_stdout(io.StringIO()):
            ret = dv.decode_color("output/color.bin", w, h, s)
        arrays[f"decode_{name}_f64"] = _NpProxy.captured   # (w,h,3) float64, before clip
        with open("output/color.ppm", "rb") as f:
            ppm = f.read()
        arrays[f"decode_{name}_color"] = color.astype(np.float32)
        arrays[f"decode_{name}_u8"] = ret                  # (w,h,3) uint8 as returned
        arrays[f"decode_{name}_ppm"] = np.frombuffer(ppm, dtype=np.uint8)
        meta["decode"][name] = {"w": w, "h": h, "s": s, "ppm_sha256": sha(ppm)}
        print("decode", name, "done", flush=True)

    run_decode("soa16", 16, 16, 1, arrays["16x16_s1_d5_soa"])
    run_decode("soa16s2", 16, 16, 2, arrays["16x16_s2_d5_soa"])
    rng = np.random.RandomState(1234)
    for s in (3, 8, 33, 64, 200, 300):                     # pins the summation order of np.mean over s
        wh = 8 if s < 64 else 4
        run_decode(f"rand_s{s}", wh, wh, s,
                   (rng.rand(3 * wh * wh * 4 * s) * 1.3 - 0.1).astype(np.float32))

    np.savez_compressed(os.path.join(HERE, "golden.npz"), **arrays)
    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("wrote", os.path.join(HERE, "golden.npz"), os.path.getsize(os.path.join(HERE, "golden.npz")))


if __name__ == "__main__":
    sys.exit(main())
