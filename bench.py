#!/usr/bin/env python3
"""bench.py -- the reference's headline metric on MI355X: Mray/s on the demo scene at
1920x1080, 8 bounces, 256 spp (BASELINE.json configs[1], "C2").

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one full frame through the hot path on every rank: rays generated on the
device, all N*D ray segments traced, 4*S samples per pixel accumulated on the device, and
(N > 1) the framebuffer slices gathered to rank 0 over RCCL (asynchronously, double-buffered: the
gather of frame k overlaps the render of frame k+1; all gathers are finished inside the timed region).  Nothing is retired or
skipped in the timed kernel (APT_FLAG_RETIRE off): every one of the W*H*4*S*D segments is
traced, like the reference does.  Inputs (the 512-byte scene) are resident in HBM before
the timed region.  Weak scaling: each rank owns a 1920-column band of a (1920*N)x1080 image.

The JSON line carries `roofline` (fp32 VALU bound: SURVEY.md 8(d), F(Ns)=20*Ns+33 fp32
operations per segment) and, at N=1 on rank 0, `cpu_baseline` (the oracle's C restatement
timed on the host cores on a bounded pixel sample of the same workload).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W, H, S, D, NS = 1920, 1080, 64, 8, 8
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


def cpu_baseline(budget_s=12.0):
    """The oracle (kind "port": the reference's own CPU path needs Huawei CANN and cannot be
    built) on all host threads, on as many 4096-pixel chunks of the C2 frame as fit in
    ~budget_s seconds.  Checker code: imported here only to be TIMED as the baseline."""
    from oracle import oracle
    threads = min(oracle.max_threads(), effective_cpus())
    sph = oracle.gen_spheres()
    p = oracle.make_params(W, H, S, depth=D, num_spheres=NS, mode=oracle.MODE_K, seed=0)
    chunk, done, seg = 4096, 0, 0
    npix = W * H
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < budget_s and done < npix:
        cnt = min(chunk, npix - done)
        _, _, _, traced = oracle.render_frame(p, sph, pixel_begin=(done * 2654435761) % (npix - cnt + 1),
                                              pixel_count=cnt, threads=threads)
        seg += traced
        done += cnt
    dt = time.perf_counter() - t0

    def short_run(nthreads, budget):   # also 1 thread and 8 threads (the reference's 8-block split)
        t1, s1, k = time.perf_counter(), 0, 0
        while time.perf_counter() - t1 < budget:
            _, _, _, tr = oracle.render_frame(p, sph, pixel_begin=(k * 2654435761) % (npix - 256), pixel_count=256,
                                              threads=nthreads)
            s1 += tr
            k += 1
        return round(s1 / (time.perf_counter() - t1) / 1e6, 3)

    by_threads = {"1": short_run(1, 2.0), "8": short_run(min(8, threads), 2.0)}
    return {"value": round(seg / dt / 1e6, 3), "unit": "Mray/s", "cores": threads, "kind": "port", "by_threads": by_threads,
            "sample": f"{done} pixels of the {W}x{H} frame x {4 * S} spp x {D} bounces = {seg} segments in {dt:.1f} s "
                      f"(K-mode C restatement, gcc -O2 -ffp-contract=off, OpenMP)"}


def quality_check(frame, pixels=2048):
    """BASELINE quality metric on a strided sample of the frame just rendered: per-channel RMS between
    the GPU's and the CPU restatement's float pixel values (after mean + clip, range [0,1]) and the
    number of differing 8-bit PPM values.  Target <= 1e-4; the GPU path is bit-identical, so 0."""
    import numpy as np
    from oracle import oracle
    fb, u8 = frame[0].cpu().numpy(), frame[1].cpu().numpy()
    sph = oracle.gen_spheres()
    p = oracle.make_params(W, H, S, depth=D, num_spheres=NS, mode=oracle.MODE_K, seed=0)
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
            "reference": "oracle C restatement, K-mode", "target_rms": 1e-4}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--retire", action="store_true", help="also time the frame with APT_FLAG_RETIRE (extra field)")
    args = ap.parse_args()

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
    local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    # weak scaling: the image grows by one 1920-column band per rank (x-major pixel index, so a
    # band of columns is a contiguous pixel range: SURVEY.md 8(e))
    width = W * world
    p = apt.make_params(width, H, S, depth=D, num_spheres=NS, mode=apt.APT_MODE_KERNEL, seed=0)
    sph = torch.from_numpy(gen_data.gen_spheres()).cuda()
    # two packed slots: the RCCL gather of frame k overlaps the render of frame k+1 (separate streams)
    shard = apt_dist.FrameShard(p, rank, world, slots=2)
    slots = shard.alloc_slots()
    full = shard.alloc_full() if rank == 0 else (None, None)

    def step(k, events=None):
        fb, u8 = slots[k % 2]
        if events:
            events[0].record()                # torch's current stream == the stream the kernel is launched on
        render.render_frame(p, sph, shard.pixel_begin, shard.pixel_count, fb=fb, fb_u8=u8)
        if events:
            events[1].record()
        if world > 1:
            shard.gather_async(k % 2, *full)  # finishes (waits + unpacks) the previous frame's gather first

    def sync():
        shard.finish()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for k in range(args.warmup):
        step(k)
    sync()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k, e in enumerate(ev):
        step(k, e)
    sync()
    dt = time.perf_counter() - t0
    kern_ms = sum(a.elapsed_time(b) for a, b in ev) / len(ev)     # HIP events around each launch

    if world > 1:
        t = torch.tensor([dt, kern_ms], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, kern_ms = float(t[0]), float(t[1])

    seg_per_rank = shard.pixel_count * 4 * S * D
    seg_total = seg_per_rank * world
    ms_per_step = dt / args.steps * 1e3
    value = seg_total / (dt / args.steps) / 1e6
    achieved = seg_per_rank * flops_per_segment(NS) / (kern_ms * 1e-3) / 1e12
    out = {
        "metric": "Mray/s (ray segments per second) at 1080p, 8 bounces, 256 spp, demo scene",
        "value": round(value, 1), "unit": "Mray/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"C2: gen_spheres() 8-sphere scene, {W}x{H} per GPU ({width}x{H} total), "
                               f"S={S} (256 spp), depth {D}, rays generated on device (counter RNG, seed 0), "
                               f"K-mode arithmetic, all segments traced (no retirement)",
                   "paths_per_gpu": shard.pixel_count * 4 * S, "segments_per_gpu": seg_per_rank,
                   "parallelism": f"pixel-column bands x{world}, one RCCL gather" if world > 1 else "single GPU"},
        "roofline": {"bound": "valu", "kernel": "render_frame_kernel<K,ns8,group8>", "achieved": round(achieved, 3),
                     "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_FP32_TFLOPS, 4),
                     "frac_of_nofma_peak": round(achieved / PEAK_NOFMA_TOPS, 4),
                     "frac_of_measured_nofma_ceiling": round(achieved / MEASURED_NOFMA_TOPS, 4),
                     "flops_per_segment": flops_per_segment(NS), "kernel_ms": round(kern_ms, 3),
                     "traffic": apt_dist.recorded_traffic(ROOT)},
        "target_mray_per_gpu": 100.0,
    }
    if args.retire:   # same frame with result-preserving retirement + wave-queue compaction (bit-identical image)
        pr = p.copy(flags=apt.APT_FLAG_RETIRE)
        for _ in range(2):
            render.render_frame(pr, sph, shard.pixel_begin, shard.pixel_count, fb=slots[0][0], fb_u8=slots[0][1])
        torch.cuda.synchronize()
        evr = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
        for a, b in evr:
            a.record()
            render.render_frame(pr, sph, shard.pixel_begin, shard.pixel_count, fb=slots[0][0], fb_u8=slots[0][1])
            b.record()
        torch.cuda.synchronize()
        rms = sum(a.elapsed_time(b) for a, b in evr) / len(evr)
        with render.TraceCounter() as tc:       # counted in a separate, untimed launch (the counter's atomics are slow)
            render.render_frame(pr, sph, shard.pixel_begin, shard.pixel_count, fb=slots[0][0], fb_u8=slots[0][1])
        out["retire"] = {"kernel_ms": round(rms, 3), "traced_segments": tc.value, "nominal_segments": seg_per_rank,
                         "nominal_mray_per_s": round(seg_per_rank / rms / 1e3, 1),
                         "traced_mray_per_s": round(tc.value / rms / 1e3, 1)}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["quality"] = quality_check(slots[(args.steps - 1) % 2])
        out["cpu_baseline"] = cpu_baseline()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
