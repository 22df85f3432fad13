#!/usr/bin/env python3
"""Round 3: the rewritten sample-queue kernel (pt_queue.h) -- correctness against the oracle on small frames (frame bits and
traced-segment counts), then C2 / C5 / C5 with roulette timed against the pixels-per-wave knob.  (Round 3 also timed the round-2
queue here, through APT_OLD_QUEUE=1; that kernel was removed in round 4 -- profiles/history/r03_queue_check.jsonl keeps its numbers.)   python profiles/debug/queue8_check.py [--skip-check] > gpurun_out/queue8.jsonl"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import ascendpathtracing_amd as apt
from ascendpathtracing_amd import gen_data, render

ap = argparse.ArgumentParser()
ap.add_argument("--skip-check", action="store_true")
ap.add_argument("--ppw", default="2,4,8,16,32")
args = ap.parse_args()
sph_h = gen_data.gen_spheres()
sph = torch.from_numpy(sph_h).cuda()


def bits(t):
    return (t.cpu().numpy() if isinstance(t, torch.Tensor) else t).view(np.uint32)


if not args.skip_check:
    from oracle import oracle
    bad = 0
    for (w, h, s, d, mode, rr) in ((12, 8, 8, 5, 0, 0), (12, 8, 20, 9, 0, 0), (10, 6, 64, 32, 0, 0), (10, 6, 136, 5, 0, 0), (7, 5, 300, 4, 1, 0),
                                   (20, 15, 16, 16, 0, 1), (9, 9, 64, 8, 1, 0), (33, 3, 32, 12, 0, 1), (5, 4, 8, 1, 0, 0), (3, 1, 100, 7, 1, 1)):
        flags = apt.APT_FLAG_RETIRE | (apt.APT_FLAG_RR if rr else 0)
        oflags = oracle.FLAG_RETIRE | (oracle.FLAG_RR if rr else 0)
        for ppw in (1, 3, 16):
            render.set_debug("queue_ppw", ppw)
            for (b, c) in ((0, w * h), (min(3, w * h - 1), min(7, w * h - min(3, w * h - 1)))):
                p = apt.make_params(w, h, s, depth=d, mode=mode, flags=flags, seed=5)
                with render.TraceCounter() as tc:
                    fb, u8 = render.render_frame(p, sph, b, c)
                op = oracle.make_params(w, h, s, depth=d, mode=(oracle.MODE_O if mode else oracle.MODE_K), flags=oflags, seed=5)
                fb_w, u8_w, _, traced = oracle.render_frame(op, sph_h, pixel_begin=b, pixel_count=c, threads=8)
                ok = np.array_equal(bits(fb), bits(fb_w)) and np.array_equal(u8.cpu().numpy(), u8_w) and tc.value == traced
                bad += not ok
                if not ok:
                    print(json.dumps({"FAIL": [w, h, s, d, mode, rr, ppw, b, c], "traced": [tc.value, int(traced)],
                                      "fb_equal": bool(np.array_equal(bits(fb), bits(fb_w)))}), flush=True)
    print(json.dumps({"small_frame_checks_failed": bad}), flush=True)
    if bad:
        sys.exit(1)


def timeit(fn, reps=4):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return best


cases = ((8, apt.APT_FLAG_RETIRE, "c2_retire"), (32, apt.APT_FLAG_RETIRE, "c5_retire"), (32, apt.APT_FLAG_RETIRE | apt.APT_FLAG_RR, "c5_rr_retire"))
full = {}
for d, name in ((8, "c2_full"), (32, "c5_full")):
    p = apt.make_params(1920, 1080, 64, depth=d)
    full[name] = round(timeit(lambda: render.render_frame(p, sph), 3), 3)
    fbf, u8f = render.render_frame(p, sph)
    full[name + "_fb"] = fbf
print(json.dumps({k: v for k, v in full.items() if not k.endswith("_fb")}), flush=True)
for ppw in [int(x) for x in args.ppw.split(",")]:
    render.set_debug("queue_ppw", ppw)
    out = {"ppw": ppw}
    for d, flags, name in cases:
        p = apt.make_params(1920, 1080, 64, depth=d, flags=flags)
        with render.TraceCounter() as tc:
            fb, u8 = render.render_frame(p, sph)
        st = tc.stats
        ms = timeit(lambda: render.render_frame(p, sph))
        out[name] = {"ms": round(ms, 3), "traced": st[0], "traced_gray_per_s": round(st[0] / ms / 1e6, 1),
                     "lane_eff_bounce": round(st[0] / max(1, st[1]), 4), "gen_lane_slots_over_paths": round(st[2] / (1920 * 1080 * 256), 4),
                     "exact": tc.exact_reruns}
        if not (flags & apt.APT_FLAG_RR):
            ref = full["c2_full_fb" if d == 8 else "c5_full_fb"]
            out[name]["equals_full_trace_frame"] = bool(torch.equal(fb.view(torch.int32), ref.view(torch.int32)))
    print(json.dumps(out), flush=True)
