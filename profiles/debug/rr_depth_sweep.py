"""GPU vs oracle on a small frame for depth 1..32, flags 0 / RR / RR|RETIRE: first depth that differs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import ascendpathtracing_amd as apt
from ascendpathtracing_amd import render, gen_data
from oracle import oracle
sph_h = gen_data.gen_spheres(); sph = torch.from_numpy(sph_h).cuda()
for flags, of in ((0, 0), (2, 2), (3, 3), (1, 1)):
    bad = []
    for d in list(range(1, 13)) + [16, 20, 24, 32]:
        for rr in ((0,) if not flags & 2 else (0, 1, 5)):
            p = apt.make_params(48, 32, 16, depth=d, flags=flags, seed=3, rr_start=rr)
            fb, u8 = render.render_frame(p, sph)
            torch.cuda.synchronize()
            fw, uw, _, _ = oracle.render_frame(oracle.make_params(48, 32, 16, depth=d, flags=of, seed=3, rr_start=rr), sph_h, threads=16)
            if not np.array_equal(fb.cpu().numpy().view(np.uint32), fw.view(np.uint32)):
                nb = int((fb.cpu().numpy().view(np.uint32) != fw.view(np.uint32)).sum())
                bad.append((d, rr, nb))
    print("flags", flags, "mismatches (depth, rr_start, values):", bad, flush=True)
