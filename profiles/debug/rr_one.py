import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import ascendpathtracing_amd as apt
from ascendpathtracing_amd import render, gen_data
from oracle import oracle
sph_h = gen_data.gen_spheres(); sph = torch.from_numpy(sph_h).cuda()
for d, rr in ((1, 1), (3, 0)):
    p = apt.make_params(16, 8, 8, depth=d, flags=2, seed=3, rr_start=rr)
    fb, u8 = render.render_frame(p, sph); torch.cuda.synchronize()
    fw, uw, _, _ = oracle.render_frame(oracle.make_params(16, 8, 8, depth=d, flags=2, seed=3, rr_start=rr), sph_h)
    g = fb.cpu().numpy(); bad = np.argwhere(g.view(np.uint32) != fw.view(np.uint32))
    print(os.environ.get("APT_LIB_PATH", "default"), "d", d, "rr", rr, "bad", len(bad), [(int(c), int(q), float(g[c, q]), float(fw[c, q])) for c, q in bad[:6]], flush=True)
    # buffer mode for the same check (one path per lane, no accumulation)
    rays = gen_data.gen_rays(16, 8, 1, seed=0)
    pb = apt.make_params(16, 8, 1, depth=d, flags=2, seed=3, rr_start=rr)
    col = render.render_paths(pb, torch.from_numpy(rays.ravel()).cuda(), sph); torch.cuda.synchronize()
    want, _ = oracle.render_paths(oracle.make_params(16, 8, 1, depth=d, flags=2, seed=3, rr_start=rr), rays, sph_h)
    c = col.cpu().numpy(); b2 = np.argwhere(c.view(np.uint32) != want.view(np.uint32))
    print("   buffer mode bad", len(b2), [(int(ch), int(q), float(c[ch, q]), float(want[ch, q])) for ch, q in b2[:6]], flush=True)
