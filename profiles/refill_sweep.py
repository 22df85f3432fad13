"""APT_FLAG_RETIRE frame kernel against the refill threshold (apt_set_refill_lanes): time, lane-slots per path."""
import os, sys, torch, json
sys.path.insert(0, os.getcwd())
import ascendpathtracing_amd as apt
from ascendpathtracing_amd import gen_data, render
sph = torch.from_numpy(gen_data.gen_spheres()).cuda()
for d, flags, name in ((8, apt.APT_FLAG_RETIRE, "C2 D8 retire"), (32, apt.APT_FLAG_RETIRE, "C5 D32 retire"),
                       (32, apt.APT_FLAG_RETIRE | apt.APT_FLAG_RR, "C5 D32 RR+retire")):
    p = apt.make_params(1920, 1080, 64, depth=d, flags=flags)
    for lanes in (8, 16, 24, 32, 40, 48, 56, 64):
        render.set_refill_lanes(lanes)
        render.render_frame(p, sph); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(3):
            a.record(); render.render_frame(p, sph); b.record(); torch.cuda.synchronize(); best = min(best, a.elapsed_time(b))
        with render.TraceCounter() as tc:
            render.render_frame(p, sph)
        t, bb, g = tc.stats; n = p.num_paths
        print(json.dumps({"case": name, "refill_lanes": lanes, "ms": round(best, 3), "traced_per_path": round(t / n, 3),
                          "bounce_slots_per_path": round(bb / n, 3), "raygen_slots_per_path": round(g / n, 3)}), flush=True)
render.set_refill_lanes(32)
