#!/usr/bin/env python3
"""bench.py -- the reference's headline metric on MI355X: Mray/s (ray segments per second).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one full frame through the hot path: rays generated on the device, every one of the
W*H*4*S*D segments traced (APT_FLAG_RETIRE off, like the reference), 4*S samples per pixel accumulated on the
device, and (N > 1) the framebuffer bands gathered to rank 0 over RCCL (asynchronously, double-buffered: the
gather of frame k overlaps the render of frame k+1; all gathers are finished inside the timed region).  Inputs
(the 512-byte scene) are resident in HBM before the timed region.

Workload (BASELINE.json `configs`):
    N = 1   configs[1] "C2": 1920x1080, 256 spp (S=64), 8 bounces, demo scene -- the configuration the metric is quoted on
    N > 1   configs[2] "C3": 4096x4096, 1024 spp (S=256), 8 bounces, strong-sharded: rank r renders the contiguous pixel
            band dist.split_range gives it (the reference's own split, src/render.cpp:9-10,24-27), one RCCL gather.
            `scaling` is "strong" (total work fixed); `value` stays a rate (segments/s of the whole job), so the per-N
            values are directly comparable with the N = 1 line.  (--workload c2-weak: one C2 band per rank instead.)

The JSON line carries `roofline` (VALU-issue bound: SURVEY.md 8(d), F(Ns)=20*Ns+33 fp32 operations per segment against
the vector fp32 peak) and, at N=1 on rank 0, `cpu_baseline` (the oracle's C restatement timed on the host cores on a
bounded pixel sample of the same workload) plus `extra` (the same frame in O-mode, with compaction, and through the
reference's exact MT19937 pipeline).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

NS = 8
C2 = dict(name="C2", w=1920, h=1080, s=64, depth=8)
C3 = dict(name="C3", w=4096, h=4096, s=256, depth=8)
PEAK_FP32_TFLOPS = 157.3                      # MI355X_MICROARCH.md: vector fp32 (== f32 MFMA) peak, FMA counted as 2
PEAK_NOFMA_TOPS = 78.6                        # same lanes with FMA forbidden by bit-parity
MEASURED_NOFMA_TOPS = 68.6                    # v_pk_add/mul_f32 issue rate measured on MI355X (profiles/microbench/valu_rates_mi355x.txt)


def flops_per_segment(ns):
    return 20 * ns + 33                       # add/sub/mul/div/sqrt = 1 each


def effective_cpus():
    """Host threads this process may really use: the affinity mask and the cgroup CPU quota both count (a GPU box
    hands one job a share of the host's cores; oversubscribing them makes the OpenMP port slower, not faster)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], int(txt[1])
            else:
                quota, period = txt[0], int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1"):
                n = min(n, max(1, int(quota) // period))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(cfg, budget_s=14.0):
    """The oracle (kind "port": the reference's own CPU path needs Huawei CANN and cannot be built) timed on the host cores on a
    bounded sample of the same workload: 4096-pixel chunks of the frame (the full 256 spp x 8 bounces per pixel), BASELINE.md
    section 4's protocol -- kernel region only, MEDIAN of >= 5 runs per thread count, at 1 thread, 8 threads (the reference's own
    8-block split, src/render.cpp:9) and all the cores this job may use.  Checker code: imported here only to be TIMED."""
    import statistics
    from oracle import oracle
    W, H, S, D = cfg["w"], cfg["h"], cfg["s"], cfg["depth"]
    threads = min(oracle.max_threads(), effective_cpus())
    sph = oracle.gen_spheres()
    p = oracle.make_params(W, H, S, depth=D, num_spheres=NS, mode=oracle.MODE_K, seed=0)
    npix = W * H

    def runs(nthreads, pixels, budget, at_least=5):
        rates, seg, k, t_all = [], 0, 0, time.perf_counter()
        while len(rates) < at_least or time.perf_counter() - t_all < budget:
            t0 = time.perf_counter()
            _, _, _, tr = oracle.render_frame(p, sph, pixel_begin=(k * 2654435761) % (npix - pixels + 1), pixel_count=pixels, threads=nthreads)
            rates.append(tr / (time.perf_counter() - t0) / 1e6)
            seg += tr
            k += 1
        return {"median": round(statistics.median(rates), 3), "min": round(min(rates), 3), "max": round(max(rates), 3), "runs": len(rates),
                "segments": seg, "pixels_per_run": pixels}

    full = runs(threads, 4096, budget_s * 0.6)
    by_threads = {"1": runs(1, 128, budget_s * 0.2), "8": runs(min(8, threads), 1024, budget_s * 0.2), str(threads): full}
    return {"value": full["median"], "unit": "Mray/s", "cores": threads, "kind": "port", "by_threads": by_threads,
            "cpu_model": cpu_model(), "host_threads_visible": os.cpu_count(),
            "sample": f"median of {full['runs']} runs of 4096 pixels of the {W}x{H} frame x {4 * S} spp x {D} bounces "
                      f"({full['segments']} segments in all; K-mode C restatement, gcc -O2 -ffp-contract=off, OpenMP, {threads} threads); "
                      f"by_threads: the same at 1 and 8 threads on smaller runs"}


def quality_check(cfg, fb, u8, pixels=2048, o_mode=False):
    """BASELINE quality metric on a strided sample of the frame just rendered: per-channel RMS between
    the GPU's and the CPU restatement's float pixel values (after mean + clip, range [0,1]) and the
    number of differing 8-bit PPM values.  Target <= 1e-4; the GPU path is bit-identical, so 0."""
    import numpy as np
    from oracle import oracle
    W, H, S, D = cfg["w"], cfg["h"], cfg["s"], cfg["depth"]
    fb, u8 = fb.cpu().numpy(), u8.cpu().numpy()
    sph = oracle.gen_spheres()
    p = oracle.make_params(W, H, S, depth=D, num_spheres=NS, mode=oracle.MODE_O if o_mode else oracle.MODE_K, seed=0)
    run, npix = 16, W * H
    starts = [(k * 2654435761) % (npix - run) for k in range(pixels // run)]
    se, diff = np.zeros(3), 0
    for q0 in starts:
        fb_w, u8_w, _, _ = oracle.render_frame(p, sph, pixel_begin=q0, pixel_count=run, threads=min(oracle.max_threads(), 16))
        d = fb[:, q0:q0 + run].astype(np.float64) - fb_w.astype(np.float64)
        se += (d * d).sum(axis=1)
        diff += int((u8[q0:q0 + run] != u8_w).sum())
    n = len(starts) * run
    return {"rms_rgb": [float(x) for x in np.sqrt(se / n)], "differing_ppm_values": diff, "pixels_checked": n,
            "reference": "oracle C restatement, " + ("O-mode (= the reference's NumPy oracle test_soa, bit for bit on its goldens)" if o_mode else "K-mode"),
            "target_rms": 1e-4}


def gathered_frame_check(full_fb, full_u8, case_names=("C3_bands", "C3_spread", "C3_band5_whole")):
    """N > 1: the frame rank 0 holds after the gather against the oracle-made hashes committed under tests/golden/
    (data, made by tests/golden/make_fullsize_hashes.py): first and last image column of each of the reference's 8 bands,
    64 ranges of 1024 pixels spread over the whole frame (interior pixels of every band) and ALL 2^21 pixels of band 5."""
    import hashlib
    import numpy as np
    with open(os.path.join(ROOT, "tests", "golden", "fullsize_hashes.json")) as f:
        cases = json.load(f)["cases"]
    bad, n, pixels = [], 0, 0
    for name in case_names:
        case = cases[name]
        for k, (b, c) in enumerate(case["ranges"]):
            fb = np.ascontiguousarray(full_fb[:, b:b + c].cpu().numpy())
            u8 = np.ascontiguousarray(full_u8[b:b + c].cpu().numpy())
            if hashlib.sha256(fb.tobytes()).hexdigest() != case["fb_sha256"][k] or hashlib.sha256(u8.tobytes()).hexdigest() != case["u8_sha256"][k]:
                bad.append(f"{name}[{k}]")
            n += 1
            pixels += c
    return {"checked_against": "tests/golden/fullsize_hashes.json: " + ", ".join(case_names) + " (oracle-made hashes of ranges of the gathered frame)",
            "ranges": n, "pixels_checked": pixels, "mismatching_ranges": bad, "ok": not bad}


def timed(torch, fn, reps):
    fn(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return sum(a.elapsed_time(b) for a, b in ev) / len(ev)


def extras(torch, apt, render, gen_data, cfg, sph, steps):
    """N = 1 only, after the timed region: the same C2 frame (a) in O-mode (the NumPy oracle's arithmetic), (b) with
    result-preserving retirement + wave-queue compaction, (c) through the reference's exact pipeline on the device
    (MT19937 gen_rays -> buffer-mode render in O-mode -> decode_color; 19 GB of intermediates)."""
    W, H, S, D = cfg["w"], cfg["h"], cfg["s"], cfg["depth"]
    out = {}
    po = apt.make_params(W, H, S, depth=D, mode=apt.APT_MODE_ORACLE)
    out["c2_o_mode_kernel_ms"] = round(timed(torch, lambda: render.render_frame(po, sph), steps), 3)
    # (a') the same frame of a scene WITHOUT the reference table's equality pattern of centre coordinates (spheres 0 and 6 exchanged: the
    # mirror ball takes the left wall's slot, so cy[0..3] are no longer equal and scene8_shares_planes() fails): the general form of the
    # intersections, 64 instead of 49 packed instructions for the 8 discriminants (how much of the headline is scene-specific)
    perm = sph.clone().view(-1)[:80].view(10, 8).clone()
    perm[:, [0, 6]] = perm[:, [6, 0]]
    sph_general = sph.clone()
    sph_general[:80] = perm.reshape(-1)
    out["general_scene_ms"] = round(timed(torch, lambda: render.render_frame(apt.make_params(W, H, S, depth=D), sph_general), steps), 3)
    pr = apt.make_params(W, H, S, depth=D, flags=apt.APT_FLAG_RETIRE)
    rms = timed(torch, lambda: render.render_frame(pr, sph), steps)
    with render.TraceCounter() as tc:       # counted in a separate, untimed launch (the counter's atomics are slow)
        render.render_frame(pr, sph)
    nominal = W * H * 4 * S * D
    out["c2_retire"] = {"kernel_ms": round(rms, 3), "traced_segments": tc.value, "nominal_segments": nominal,
                        "nominal_mray_per_s": round(nominal / rms / 1e3, 1), "traced_mray_per_s": round(tc.value / rms / 1e3, 1)}
    try:
        t0 = time.time()
        ck = torch.from_numpy(gen_data.mt19937_checkpoints(W * H * 4 * S, seed=0, stride=64).view("int32")).cuda()
        t_ck = time.time() - t0
        rays = gen_data.gen_rays_device(W, H, S, checkpoints=ck, stride=64).reshape(-1)
        colors = torch.empty(3 * po.num_paths, device="cuda")
        ms_gen = timed(torch, lambda: gen_data.gen_rays_device(W, H, S, checkpoints=ck, stride=64), 2)
        ms_ren = timed(torch, lambda: render.render_do_ex(po, None, rays, sph, colors), 2)
        ms_dec = timed(torch, lambda: render.decode_color_device(po, colors), 2)
        out["c2_exact_reference_pipeline"] = {"total_ms": round(ms_gen + ms_ren + ms_dec, 3), "gen_rays_mt19937_ms": round(ms_gen, 3),
                                              "render_o_mode_buffer_ms": round(ms_ren, 3), "decode_color_ms": round(ms_dec, 3),
                                              "host_checkpoint_seconds_once": round(t_ck, 2), "hbm_gb_resident": round(36 * po.num_paths / 1e9, 1)}
        del rays, colors, ck
        torch.cuda.empty_cache()
    except Exception as e:                   # noqa: BLE001  (an extra must never cost the headline line)
        out["c2_exact_reference_pipeline"] = {"error": repr(e)[:200]}
    try:                                     # (d) the same pipeline in ONE kernel (round 3): MT19937 twisted in LDS, no ray / colour buffers
        import numpy as np
        t0 = time.time()
        ckg, g_lo = render.mt_group_checkpoints(W, H, S, seed=0)
        t_ck = time.time() - t0
        ckd = (torch.from_numpy(ckg.view(np.int32)).cuda(), g_lo)
        ms_o = timed(torch, lambda: render.render_reference_frame_fused(W, H, S, depth=D, seed=0, spheres=sph, checkpoints=ckd), 3)
        ms_k = timed(torch, lambda: render.render_reference_frame_fused(W, H, S, depth=D, seed=0, spheres=sph, checkpoints=ckd, mode=apt.APT_MODE_KERNEL), 3)
        out["c2_exact_reference_pipeline_fused"] = {"total_ms": round(ms_o, 3), "k_mode_arithmetic_ms": round(ms_k, 3),
                                                    "host_checkpoint_seconds_once": round(t_ck, 2), "hbm_gb_resident": round((ckg.nbytes + 15 * W * H) / 1e9, 3),
                                                    "kernel": "render_frame_mt_kernel (apt_render_frame_mt): MT19937 gen_rays + render (O-mode) + decode_color"}
        del ckd
        torch.cuda.empty_cache()
    except Exception as e:                   # noqa: BLE001
        out["c2_exact_reference_pipeline_fused"] = {"error": repr(e)[:200]}
    try:                                     # (e) BASELINE config C4: 10 000 spheres behind the uniform grid, 256 spp (sample-queue kernel, grid form)
        import numpy as np
        ns4 = 10000
        scene4 = torch.from_numpy(gen_data.gen_scene(ns4, seed=1)).cuda()
        grid4 = gen_data.build_grid_device(scene4, ns4)
        torch.cuda.synchronize()
        # APT_FLAG_GRID_SLOTS (apt_grid_flags reads the built grid's header): ONE launch per frame
        p4 = apt.make_params(W, H, S, depth=D, num_spheres=ns4, accel=grid4.data_ptr(), flags=gen_data.grid_flags(grid4, ns4))
        ms_full = timed(torch, lambda: render.render_frame(p4, scene4), 2)
        ms_ret = timed(torch, lambda: render.render_frame(p4.copy(flags=p4.flags | apt.APT_FLAG_RETIRE), scene4), 2)
        # walk statistics from a separate, untimed launch at 64 spp (the counting instantiation of the kernel; per-segment averages do
        # not depend on the sample count): cells visited and candidates tested per traced segment -> the EXECUTED work of a culled
        # traversal, 20 flops per candidate test + 33 per segment (SURVEY 8(d)'s per-pair / per-segment counts), against the vector peak
        with render.TraceCounter() as tc4:
            render.render_frame(p4.copy(samples=16), scene4)
        traced4, cells4, tests4 = tc4.stats
        tps, cps = tests4 / max(traced4, 1), cells4 / max(traced4, 1)
        flops_seg = 20.0 * tps + 33.0
        achieved4 = nominal * flops_seg / (ms_full * 1e-3) / 1e12
        out["c4_grid_10k_spheres"] = {"kernel_ms": round(ms_full, 3), "retire_kernel_ms": round(ms_ret, 3), "nominal_mray_per_s": round(nominal / ms_full / 1e3, 1),
                                      "tests_per_segment": round(tps, 2), "cells_per_segment": round(cps, 2),
                                      "flops_per_segment_executed": round(flops_seg, 1), "achieved": round(achieved4, 3), "unit": "TFLOP/s",
                                      "peak": PEAK_FP32_TFLOPS, "frac": round(achieved4 / PEAK_FP32_TFLOPS, 4), "bound": "valu",
                                      "kernel": "render_frame_queue8_kernel<..., grid> (pt_queue.h run_grid); frame bit-identical to the brute-force traversal"}
        del scene4, grid4
    except Exception as e:                   # noqa: BLE001
        out["c4_grid_10k_spheres"] = {"error": repr(e)[:200]}
    return out


def dry_run(args):
    """bench.py's N-rank control flow on CPU with gloo and a no-op in place of the render launch.  Nothing is
    measured and nothing of the oracle is touched: this only proves that `torchrun ... bench.py --gpus N` rendezvouses,
    shards C3's geometry (scaled down), gathers asynchronously into rank 0's frame and prints one well-formed line."""
    import torch
    import torch.distributed as dist
    import ascendpathtracing_amd as apt
    from ascendpathtracing_amd import dist as apt_dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with --nproc-per-node {args.gpus} (WORLD_SIZE is {world})")
    if world > 1:
        dist.init_process_group("gloo")
    cfg = dict(C3 if world > 1 else C2, w=64, h=64, s=8)          # same code path, toy size
    W, H, S, D = cfg["w"], cfg["h"], cfg["s"], cfg["depth"]
    p = apt.make_params(W, H, S, depth=D, num_spheres=NS, seed=0)
    shard = apt_dist.FrameShard(p, rank, world, device="cpu", slots=2, stripes=args.stripes if world > 1 else 1)
    slots = shard.alloc_slots()
    full = shard.alloc_full() if rank == 0 else (None, None)

    def fake_render(params, spheres, pixel_begin, pixel_count, fb=None, fb_u8=None):   # writes the pixel index: the gather is checkable
        idx = torch.arange(pixel_begin, pixel_begin + pixel_count, dtype=torch.float32)
        fb.copy_(idx.repeat(3, 1))
        fb_u8.copy_((idx.to(torch.int64) % 251).to(torch.uint8).repeat(3, 1).t())

    t0 = time.perf_counter()
    for k in range(args.warmup + args.steps):
        shard.render(slots[k % 2], None, fake_render, slot=k % 2)
        if world > 1:
            shard.gather_async(k % 2, *full)
    shard.finish()
    shard.drain()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ok = True
    if rank == 0 and world > 1:
        want = torch.arange(W * H, dtype=torch.float32)
        ok = bool(torch.equal(full[0][0], want) and torch.equal(full[1][:, 2], (torch.arange(W * H) % 251).to(torch.uint8)))
    if rank == 0:
        print(json.dumps({"metric": "dry run (no render, no measurement)", "dry_run": True, "value": None, "unit": "Mray/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "scaling": "strong" if world > 1 else "weak",
                          "config": {"workload": f"{cfg['name']} geometry at toy size {W}x{H}", "pixels_per_rank": shard.pixel_count,
                                     "stripes": shard.stripes, "collective": "gloo gather (rehearsal)", "peer_access": None},
                          "ranks": ({"gather": shard.timings()} if world > 1 else None),
                          "gathered_frame_complete": ok}), flush=True)
    if world > 1:
        dist.destroy_process_group()


HEADLINE_KERNEL = "render_frame_kernel<0, 0, 8, false, true>"      # K-mode, 8-sphere scene, 8 lanes per sub-pixel, no retirement, two paths per lane


def traffic_probe():
    """`bench.py --traffic-probe`: what measure_counters() wraps in rocprofv3 -- the headline launch (C2, K-mode, every segment traced)
    a few times, nothing else (no oracle, no timing, no output).  This process runs UNDER the profiler, whose preloaded library has
    initialised the GPU before main() starts: it must not spawn anything (a child would inherit the preload; make's recipe shells exec
    hipcc, and an exec in a process that holds the GPU takes the machine down on this pool -- ADVICE r5).  So it never builds: the
    un-profiled parent has, and a stale library is an error here."""
    import __graft_entry__
    if not __graft_entry__._is_current():
        print("bench.py --traffic-probe: the in-tree library is missing or older than its sources; build it first "
              "(python __graft_entry__.py) -- the probe itself never builds", file=sys.stderr)
        sys.exit(3)
    import torch
    import ascendpathtracing_amd as apt
    from ascendpathtracing_amd import gen_data, render
    apt._lib.require_gpu()
    p = apt.make_params(C2["w"], C2["h"], C2["s"], depth=C2["depth"], num_spheres=NS, mode=apt.APT_MODE_KERNEL, seed=0)
    sph = torch.from_numpy(gen_data.gen_spheres()).cuda()
    fb = torch.empty((3, C2["w"] * C2["h"]), dtype=torch.float32, device="cuda")
    u8 = torch.empty((C2["w"] * C2["h"], 3), dtype=torch.uint8, device="cuda")
    for _ in range(5):
        render.render_frame(p, sph, fb=fb, fb_u8=u8)
    torch.cuda.synchronize()


# The counter passes of the live roofline block: one rocprofv3 child process each (MI355X_MICROARCH.md: counters in their own passes, with
# --kernel-trace only).  The third is the VALU side north_star asks for ("achieved HBM GB/s and VALU occupancy").
PMC_PASSES = (("FETCH_SIZE",), ("WRITE_SIZE",),
              ("SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "GRBM_GUI_ACTIVE"))
N_SIMD, N_XCD = 1024, 8                       # MI355X: 256 CUs x 4 SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs


def counter_rows(out_dir, kernel=None):
    """{counter: [value per launch]}, [duration ns per launch] of the launches of `kernel` in a rocprofv3 --output-format csv directory.
    Launches are matched between the counter file and the trace by dispatch id; the FIRST launch of the process is dropped from both
    (it carries the write-back of the zero-filled buffers and the code-object load: profiles/summarize.py does the same)."""
    import csv
    import glob
    kernel = HEADLINE_KERNEL if kernel is None else kernel
    vals, dur = {}, {}
    for f in sorted(glob.glob(os.path.join(out_dir, "**", "*_counter_collection.csv"), recursive=True)):
        for row in csv.DictReader(open(f)):
            if kernel in row["Kernel_Name"]:       # (rows of one dispatch and counter -- instances, if the tool splits them -- add up)
                d = vals.setdefault(row["Counter_Name"], {})
                k = int(row.get("Dispatch_Id", 0) or len(d) + 1)
                d[k] = d.get(k, 0.0) + float(row["Counter_Value"])
    for f in sorted(glob.glob(os.path.join(out_dir, "**", "*_kernel_trace.csv"), recursive=True)):
        for row in csv.DictReader(open(f)):
            if kernel in row["Kernel_Name"]:
                dur[int(row.get("Dispatch_Id", 0) or len(dur) + 1)] = float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
    steady = lambda d: [v for _, v in sorted(d.items())][1:] if len(d) > 1 else list(d.values())     # noqa: E731
    return {k: steady(v) for k, v in vals.items()}, steady(dur)


def derive_valu(avg, kernel_ns, segments):
    """The VALU side of the roofline from the per-launch averages of PMC_PASSES[2] (formulas of profiles/summarize.py, which writes
    profiles/rNN_pmc.json from the committed passes: the two must agree)."""
    cyc = avg["GRBM_GUI_ACTIVE"] / N_XCD                          # shader cycles of the launch
    out = {"valu_insts_per_simd_cycle": round(avg["SQ_INSTS_VALU"] / (cyc * N_SIMD), 4),
           "waves_per_simd": round(avg["SQ_WAVE_CYCLES"] * 4 / (cyc * N_SIMD), 3),          # SQ_WAVE_CYCLES counts quad-cycles
           "valu_insts_per_segment": round(avg["SQ_INSTS_VALU"] * 64 / segments, 2),           # wave-level instructions x 64 lanes / lane-segments
           "effective_clock_ghz": round(cyc / kernel_ns, 3)}
    if "SQ_THREAD_CYCLES_VALU" in avg and avg.get("SQ_ACTIVE_INST_VALU"):
        out["lane_activity"] = round(avg["SQ_THREAD_CYCLES_VALU"] / (avg["SQ_ACTIVE_INST_VALU"] * 64), 4)
    return out


def measure_counters(timeout_s=150):
    """The roofline block's counters measured IN THIS RUN: HBM bytes per launch of the headline kernel (`--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`:
    separate passes, KiB, FETCH_SIZE doubled on gfx950, as MI355X_MICROARCH.md's HBM section prescribes) and its VALU issue rate, resident
    waves and lane activity (a third pass), each pass `rocprofv3 --kernel-trace --pmc ... -- python3 bench.py --traffic-probe` as a CHILD
    process before this process has touched the GPU (bench.py cannot wrap itself).  -> dict for the roofline block; a pass that fails
    leaves its keys out and says why under "<what>_probe_error" (no rocprofv3, a pass failed, already under a profiler)."""
    import shutil
    import signal
    import subprocess
    import tempfile
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return {"error": "this process runs under a profiler already"}
    tool = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if tool is None:
        return {"error": "rocprofv3 not found"}
    tmp = tempfile.mkdtemp(prefix="apt_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    avg, launches, probe_ns, errors = {}, {}, {}, {}
    try:
        for group in PMC_PASSES:
            out = os.path.join(tmp, group[0])
            cmd = [tool, "--kernel-trace", "--pmc", *group, "--output-format", "csv", "-d", out, "--",
                   sys.executable, os.path.abspath(__file__), "--traffic-probe"]
            what = "traffic" if group[0] in ("FETCH_SIZE", "WRITE_SIZE") else "valu"
            try:
                # own session: on a timeout the WHOLE group dies -- rocprofv3 and the probe's python underneath it, which would otherwise
                # keep rendering on the GPU while the parent times its steps
                proc = subprocess.Popen(cmd, env=env, cwd="/tmp", stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, start_new_session=True)
                try:
                    _, err = proc.communicate(timeout=timeout_s)
                except subprocess.TimeoutExpired:
                    try:
                        os.killpg(proc.pid, signal.SIGKILL)
                    except OSError:
                        pass
                    proc.communicate()
                    errors[what] = f"rocprofv3 --pmc {' '.join(group)} timed out after {timeout_s} s (process group killed)"
                    continue
                if proc.returncode != 0:
                    errors[what] = f"rocprofv3 --pmc {' '.join(group)} exited {proc.returncode}: {err.decode(errors='replace')[-200:]}"
                    continue
                vals, dur = counter_rows(out)
                missing = [c for c in group if not vals.get(c)]
                if missing:
                    errors[what] = f"no {missing} rows for {HEADLINE_KERNEL}"
                    continue
                for c in group:
                    avg[c] = sum(vals[c]) / len(vals[c])
                    launches[c] = len(vals[c])
                probe_ns[group[0]] = sum(dur) / max(1, len(dur))
            except (OSError, subprocess.SubprocessError, KeyError, ValueError) as e:
                errors[what] = repr(e)[:200]
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    res = {}
    if "FETCH_SIZE" in avg and "WRITE_SIZE" in avg:
        fetch, write = 2.0 * avg["FETCH_SIZE"] * 1024.0, avg["WRITE_SIZE"] * 1024.0
        res.update({"traffic": round(fetch + write), "traffic_fetch_bytes_x2": round(fetch), "traffic_write_bytes": round(write),
                    "traffic_launches_averaged": launches["WRITE_SIZE"], "traffic_probe_kernel_ms": round(probe_ns["WRITE_SIZE"] / 1e6, 3),
                    "traffic_source": "measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, KiB, "
                                      "FETCH_SIZE x2 on gfx950) of `bench.py --traffic-probe` as child processes, first launch dropped"})
    elif "traffic" in errors:
        res["traffic_probe_error"] = errors["traffic"]
    if all(c in avg for c in ("GRBM_GUI_ACTIVE", "SQ_INSTS_VALU", "SQ_WAVE_CYCLES")):
        seg = C2["w"] * C2["h"] * 4 * C2["s"] * C2["depth"]
        res.update(derive_valu(avg, probe_ns[PMC_PASSES[2][0]], seg))
        res["valu_probe_kernel_ms"] = round(probe_ns[PMC_PASSES[2][0]] / 1e6, 3)
        res["valu_launches_averaged"] = launches["SQ_INSTS_VALU"]
        res["valu_source"] = ("measured in this run: rocprofv3 --kernel-trace --pmc " + " ".join(PMC_PASSES[2]) + " of `bench.py --traffic-probe`; "
                              "valu_insts_per_simd_cycle = SQ_INSTS_VALU / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs) against ~0.238 issuable for this instruction "
                              "mix (profiles/microbench/issue_model_mi355x.txt); waves_per_simd = SQ_WAVE_CYCLES x 4 / the same cycles; "
                              "lane_activity = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU)")
    elif "valu" in errors:
        res["valu_probe_error"] = errors["valu"]
    return res


def self_launch(n):
    """`python bench.py --gpus N` typed as is (no torchrun): one process per GPU, started as a CHILD process before this one has
    touched the GPU (nothing here imports torch; a process that has initialised HIP must never exec another program on this pool),
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 bench.py <same arguments>`; rank 0's JSON
    line passes through on stdout, the exit code is the launcher's.  The reference's counterpart of this split is its 8-block launch,
    src/render.cpp:9-10,24-27 (blockDim = USE_CORE_NUM contiguous partitions of the ray range)."""
    import socket
    import subprocess
    with socket.socket() as sock:            # a free rendezvous port (the driver's own torchrun form passes --master-port itself)
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")    # torchrun would set it (with a warning) anyway
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def main():
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # before anything initialises HIP (dmabuf IPC only on this pool)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=["auto", "c2", "c3", "c2-weak"], default="auto")
    ap.add_argument("--stripes", type=int, default=1, help="interleaved stripes per rank (N > 1; see dist.py)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    ap.add_argument("--no-traffic-probe", action="store_true", help="do not measure roofline.traffic / the VALU counters with rocprofv3 child processes (N = 1)")
    ap.add_argument("--traffic-probe", action="store_true", help="internal: the launches measure_counters() profiles")
    ap.add_argument("--dry-run", action="store_true",
                    help="control-flow rehearsal without a GPU (tests/test_bench_dry_run.py): gloo, CPU tensors, a tiny frame and NO "
                         "render at all (the launch is a no-op) -- exercises the rendezvous, sharding, double-buffered gather, "
                         "reductions and the JSON line; the numbers mean nothing and the line says so")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args.gpus)         # plain `python bench.py --gpus N`: start the N ranks ourselves
    if args.dry_run:
        return dry_run(args)
    if args.traffic_probe:
        return traffic_probe()
    import __graft_entry__
    __graft_entry__.build()                   # incremental, serialised by a file lock; does not touch the GPU

    live = None
    if args.gpus == 1 and "WORLD_SIZE" not in os.environ and args.workload in ("auto", "c2") and not args.no_traffic_probe:
        live = measure_counters()             # child processes (they find the library built); this process has not touched the GPU yet

    import torch
    import torch.distributed as dist
    import ascendpathtracing_amd as apt
    from ascendpathtracing_amd import gen_data, render
    from ascendpathtracing_amd import dist as apt_dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with --nproc-per-node {args.gpus} "
                         f"(WORLD_SIZE is {world})")
    apt._lib.require_gpu()
    ndev = torch.cuda.device_count()
    if os.environ.get("APT_BENCH_SHARE_GPU") == "1" and ndev:   # rehearsal on a one-GPU box only (profiles/run_share_gpu.sh):
        local_rank %= ndev                                       # ranks share a card, the numbers are no measurement
    if local_rank >= ndev:
        raise SystemExit(f"LOCAL_RANK {local_rank} but only {ndev} GPU(s) visible: one process per GPU, no sharing")
    torch.cuda.set_device(local_rank)
    if world > 1:
        if os.environ.get("APT_BENCH_SHARE_GPU") == "1":   # RCCL refuses two ranks on one card ("Duplicate GPU detected")
            dist.init_process_group("gloo")                # gloo stages the CUDA tensors through the host
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    workload = args.workload
    if workload == "auto":
        workload = "c2" if world == 1 else "c3"
    cfg = dict(C3 if workload == "c3" else C2)
    if workload == "c2-weak":                  # the image grows by one 1920-column band per rank (x-major pixel index:
        cfg["w"] = cfg["w"] * world            # a band of columns is a contiguous pixel range)
    W, H, S, D = cfg["w"], cfg["h"], cfg["s"], cfg["depth"]
    p = apt.make_params(W, H, S, depth=D, num_spheres=NS, mode=apt.APT_MODE_KERNEL, seed=0)
    sph = torch.from_numpy(gen_data.gen_spheres()).cuda()
    # two packed slots: the RCCL gather of frame k overlaps the render of frame k+1 (separate streams)
    shard = apt_dist.FrameShard(p, rank, world, slots=2, stripes=args.stripes if world > 1 else 1)
    slots = shard.alloc_slots()
    full = shard.alloc_full() if rank == 0 else (None, None)

    def step(k, events=None):
        if events:
            events[0].record()                # torch's current stream == the stream the kernels are launched on
        shard.render(slots[k % 2], sph, render.render_frame, slot=k % 2)
        if events:
            events[1].record()
        if world > 1:
            shard.gather_async(k % 2, *full)  # finishes (waits + unpacks) the previous frame's gather first

    def sync():
        shard.finish()
        shard.drain()                         # the root's unpack runs on a side stream
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for k in range(args.warmup):
        step(k)
    sync()
    shard.reset_timings()                     # the gather / unpack split below covers the timed steps only
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k, e in enumerate(ev):
        step(k, e)
    sync()
    dt = time.perf_counter() - t0
    kern_ms = sum(a.elapsed_time(b) for a, b in ev) / len(ev)     # HIP events around this rank's launch(es)

    per_rank_ms = [kern_ms]
    if world > 1:
        t = torch.tensor([dt, kern_ms], dtype=torch.float64, device="cuda")
        allk = [torch.zeros(1, dtype=torch.float64, device="cuda") for _ in range(world)]
        dist.all_gather(allk, t[1:2].clone())
        per_rank_ms = [float(x) for x in allk]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, kern_ms = float(t[0]), float(t[1])

    seg_rank = shard.pixel_count * 4 * S * D
    seg_total = W * H * 4 * S * D
    ms_per_step = dt / args.steps * 1e3
    value = seg_total / (dt / args.steps) / 1e6
    achieved = seg_rank * flops_per_segment(NS) / (kern_ms * 1e-3) / 1e12
    traffic, traffic_tag = apt_dist.recorded_traffic(ROOT)
    try:
        build_id = apt._lib.build_id()        # hashes the sources beside the library; a measurement must not be lost when they are absent
    except OSError:
        build_id = None
    if workload != "c2" or world != 1:
        traffic, traffic_tag = None, None     # the committed PMC passes are of the N = 1 C2 launch
    traffic_extra = {}
    if live is not None:                      # measured in this very run: those are the figures
        traffic_extra = {k: v for k, v in live.items() if k not in ("traffic", "error")}   # ("error": no pass ran at all -> the *_probe_error keys below)
        if "traffic" in live:
            traffic, traffic_tag = live["traffic"], None
            traffic_extra["traffic_over_algorithmic"] = round(live["traffic"] / (W * H * 15 + 512), 4)
            traffic_extra["hbm_gbps"] = round(live["traffic"] / (kern_ms * 1e-3) / 1e9, 3)       # achieved HBM rate of the headline kernel (peak ~8000)
        else:
            traffic_extra.setdefault("traffic_probe_error", live.get("error"))
            traffic_extra["traffic_source"] = "profiles/hbm_traffic.json (committed PMC passes of this build)" if traffic is not None else None
        if "valu_insts_per_simd_cycle" not in live:
            traffic_extra.setdefault("valu_probe_error", live.get("error"))
    names = {"c2": f"C2: gen_spheres() 8-sphere scene, {W}x{H}, S={S} ({4 * S} spp), depth {D}",
             "c3": f"C3: gen_spheres() 8-sphere scene, {W}x{H}, S={S} ({4 * S} spp), depth {D}, strong-sharded over {world} rank(s): "
                   f"contiguous pixel bands (dist.split_range), {shard.pixel_count} pixels per rank",
             "c2-weak": f"C2 weak scaling: one 1920x{H} band per rank ({W}x{H} total), S={S}, depth {D}"}
    out = {
        "metric": f"Mray/s (ray segments per second) at {W}x{H}, {D} bounces, {4 * S} spp, demo scene"
                  + ("" if workload == "c2" else " (BASELINE metric: 1080p, 8 bounces, 256 spp -- this line is the multi-GPU workload, same rate unit)"),
        "value": round(value, 1), "unit": "Mray/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
        "scaling": "weak" if workload == "c2-weak" or world == 1 else "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": names[workload] + ", rays generated on device (counter RNG, seed 0), K-mode arithmetic, "
                               "all segments traced (no retirement)",
                   "paths_per_gpu": shard.pixel_count * 4 * S, "segments_per_gpu": seg_rank, "segments_per_step": seg_total,
                   "parallelism": (f"pixel bands x{world} ({shard.stripes} stripe(s) per rank), one RCCL gather to rank 0, "
                                   f"double-buffered") if world > 1 else "single GPU"},
        "roofline": {"bound": "valu", "kernel": "render_frame_kernel<K,ns8,group8>", "achieved": round(achieved, 3),
                     "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_FP32_TFLOPS, 4),
                     "frac_of_nofma_peak": round(achieved / PEAK_NOFMA_TOPS, 4),
                     "frac_of_measured_nofma_ceiling": round(achieved / MEASURED_NOFMA_TOPS, 4),
                     "flops_per_segment": flops_per_segment(NS), "kernel_ms": round(kern_ms, 3),
                     # the algorithmic count; 30 of the 193 belong to terms that spheres sharing a centre
                     # coordinate have in common and that the kernel evaluates once
                     "flops_per_segment_executed": 163,   # holds for the gen_spheres() table (its equality pattern), which every workload here renders
                     "frac_of_peak_by_executed_flops": round(achieved * 163.0 / flops_per_segment(NS) / PEAK_FP32_TFLOPS, 4),
                     "traffic": traffic, "algorithmic_bytes": shard.pixel_count * 15 + 512, **traffic_extra,
                     "traffic_recorded_for_build": traffic_tag, "build_id": build_id},
        "target_mray_per_gpu": 100.0,
    }
    if world > 1:       # what the first hardware run needs in order to explain itself: the collective's library, who can reach whom
        try:
            nccl = ".".join(str(x) for x in torch.cuda.nccl.version())
        except Exception:                     # noqa: BLE001
            nccl = None
        peers = [[bool(i == j or torch.cuda.can_device_access_peer(i, j)) for j in range(ndev)] for i in range(ndev)]
        out["config"].update({"collective": f"torch.distributed gather, backend {dist.get_backend()} (RCCL), NCCL API version {nccl}",
                              "visible_devices": ndev, "peer_access": peers,
                              "hsa_enable_ipc_mode_legacy": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")})
    if world > 1:       # strong scaling: the frame takes max(band kernel) + whatever the gather / sync leaves uncovered
        seg_r = [n * 4 * S * D for n in shard.pixel_counts()]            # every rank knows the whole split
        out["ranks"] = {"kernel_ms_per_rank": [round(x, 3) for x in per_rank_ms],
                        "roofline_per_rank": [{"rank": r, "segments": seg_r[r], "achieved": round(seg_r[r] * flops_per_segment(NS) / (per_rank_ms[r] * 1e-3) / 1e12, 3),
                                               "frac": round(seg_r[r] * flops_per_segment(NS) / (per_rank_ms[r] * 1e-3) / 1e12 / PEAK_FP32_TFLOPS, 4)}
                                              for r in range(world)],
                        "slowest_band_kernel_ms": round(max(per_rank_ms), 3),
                        "uncovered_gather_and_sync_ms_per_step": round(ms_per_step - max(per_rank_ms), 3),
                        # the root's side of it, per gather (events on the unpack stream): the wait for the collective that the next
                        # frame's render did not cover, and the scatter of the stripes into the frame (one strided copy per group of ranks)
                        "gather": shard.timings(),
                        "gather_bytes_per_rank": shard.packed_bytes,
                        "load_imbalance_max_over_mean": round(max(per_rank_ms) / (sum(per_rank_ms) / world), 4)}
    if rank == 0 and world > 1 and workload == "c3":
        out["quality"] = gathered_frame_check(*full)       # the last frame's gather has finished (sync above)
    if rank == 0 and world == 1:
        if not args.no_cpu_baseline:
            fbv, u8v = slots[(args.steps - 1) % 2][0]
            out["quality"] = quality_check(cfg, fbv, u8v)
            out["cpu_baseline"] = cpu_baseline(cfg)
        if workload == "c2":
            # The SAME workload in the arithmetic the reference pins at every depth: O-mode = scripts/gen_data.py:246-429
            # test_soa, whose two shading dot products accumulate float32 products in float64 (np.linalg.norm / np.dot, :347,:349).  K-mode
            # (the Ascend kernel's all-fp32 order, what `value` is quoted on) is only pinned by the reference up to depth 2.  Own kernel
            # timing (HIP events), own roofline fraction, own quality check against the oracle in THAT mode.
            po = apt.make_params(W, H, S, depth=D, num_spheres=NS, mode=apt.APT_MODE_ORACLE, seed=0)
            fbo = torch.empty((3, W * H), dtype=torch.float32, device="cuda")
            u8o = torch.empty((W * H, 3), dtype=torch.uint8, device="cuda")
            o_ms = timed(torch, lambda: render.render_frame(po, sph, fb=fbo, fb_u8=u8o), args.steps)
            o_ach = seg_total * flops_per_segment(NS) / (o_ms * 1e-3) / 1e12
            out["roofline_o_mode"] = {"bound": "valu", "kernel": "render_frame_kernel<O,ns8,group8>", "kernel_ms": round(o_ms, 3),
                                      "value_mray_per_s": round(seg_total / o_ms / 1e3, 1), "achieved": round(o_ach, 3), "peak": PEAK_FP32_TFLOPS,
                                      "unit": "TFLOP/s", "frac": round(o_ach / PEAK_FP32_TFLOPS, 4),
                                      "frac_of_nofma_peak": round(o_ach / PEAK_NOFMA_TOPS, 4), "flops_per_segment": flops_per_segment(NS),
                                      "over_k_mode": round(o_ms / kern_ms, 4),
                                      "arithmetic": "scripts/gen_data.py test_soa: float64-accumulated norm and dot (10 more issue slots per path and bounce "
                                                    "than the packed fp32 sums: 3 v_cvt_f64_f32 + 2 v_add_f64 + 1 v_cvt_f32_f64 per dot product and path instead of one packed add)"}
            if not args.no_cpu_baseline:
                out["roofline_o_mode"]["quality"] = quality_check(cfg, fbo, u8o, o_mode=True)
            del fbo, u8o
        if not args.no_extra and workload == "c2":
            out["extra"] = extras(torch, apt, render, gen_data, cfg, sph, max(3, args.steps // 2))
    if os.environ.get("APT_BENCH_SHARE_GPU") == "1":
        out["rehearsal"] = "ranks shared one GPU (APT_BENCH_SHARE_GPU=1): control flow and gather only, NOT a measurement"
        out["value"] = None
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    try:
        main()
    except SystemExit:
        raise
    except BaseException as e:               # noqa: BLE001 -- a multi-GPU run that dies must still say WHY on the one line the driver reads
        import traceback
        traceback.print_exc()
        if int(os.environ.get("RANK", "0")) == 0:
            print(json.dumps({"metric": "Mray/s (ray segments per second)", "value": None, "unit": "Mray/s",
                              "n_gpus": int(os.environ.get("WORLD_SIZE", "1")), "error": f"{type(e).__name__}: {e}"[:500]}), flush=True)
        sys.exit(1)
