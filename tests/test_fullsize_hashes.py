"""Every BASELINE.json configuration at its FULL size on the HIP path, compared by SHA-256 with the frames the
oracle produced in the build container (tests/golden/fullsize_hashes.json, written by
tests/golden/make_fullsize_hashes.py).  Always on under -m gpu: the whole file costs ~2 s of GPU time.

    C2        1920x1080, 256 spp, 8 bounces, demo scene           whole frame, K-mode (+ O-mode ranges)
    C5_rr     1920x1080, 256 spp, 32 bounces, Russian roulette    whole frame (BASELINE configs[4] as named)
    C5_full   the same without roulette                           whole frame
    C3_bands  4096x4096, 1024 spp, 8 bounces                      first/last column of each of the 8 rank bands
    C3_band5_whole / C3_spread (round 3)                          ALL 2^21 pixels of rank band 5; 64 ranges of 1024 pixels
                                                                  spread over the whole frame (every band's interior)
    C4_1080p_spread (round 3)                                     64 more ranges of the 10 000-sphere 1080p frame
    C4_1080p_spread2 (round 4)                                    384 more ranges of 256 pixels of it (4.7 % of the frame)
    C4_*      10 000-sphere scene                                 a whole 480x270 frame (brute force and grid) and
                                                                  ranges of the 1080p / 256 spp frame (grid)
The CPU test at the bottom re-runs the oracle on a few ranges so that the committed file cannot drift from it.
"""
import hashlib
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with open(os.path.join(ROOT, "tests", "golden", "fullsize_hashes.json")) as _f:
    HASHES = json.load(_f)["cases"]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def apt():
    import __graft_entry__ as g
    g.build()
    import ascendpathtracing_amd as pkg
    from ascendpathtracing_amd import _lib, gen_data, render
    _lib.require_gpu()          # fail loudly: no silent fallback
    pkg.render, pkg.gen_data = render, gen_data
    return pkg


def _scene(apt, case, torch):
    """-> (device scene table, num_spheres, device grid or None)"""
    if case["scene"] == "demo":
        return torch.from_numpy(apt.gen_data.gen_spheres()).cuda(), 8, None
    ns, seed = case["scene"]["num_spheres"], case["scene"]["seed"]
    host = apt.gen_data.gen_scene(ns, seed=seed)
    grid = torch.from_numpy(apt.gen_data.build_grid(host, ns).view(np.int32)).cuda()
    return torch.from_numpy(host).cuda(), ns, grid


def _params(apt, case, ns, flags=None, accel=0):
    return apt.make_params(case["w"], case["h"], case["s"], depth=case["depth"], num_spheres=ns,
                           mode=apt.APT_MODE_ORACLE if case["mode"] == "O" else apt.APT_MODE_KERNEL,
                           flags=case["flags"] if flags is None else flags, seed=case["seed"],
                           rr_start=case.get("rr_start", 0), accel=accel)


def _check_ranges(case, fb, u8, base, what):
    """fb [3][n] / u8 [n][3] device tensors holding pixels [base, base+n); compares every range inside."""
    fb, u8 = fb.cpu().numpy(), u8.cpu().numpy()
    n, bad = fb.shape[1], []
    for k, (b, c) in enumerate(case["ranges"]):
        if b < base or b + c > base + n:
            continue
        if sha(fb[:, b - base:b - base + c]) != case["fb_sha256"][k] or sha(u8[b - base:b - base + c]) != case["u8_sha256"][k]:
            bad.append(k)
    assert not bad, f"{what}: ranges {bad} differ from the oracle's frame"


def _whole_frame(apt, name, flag_sets):
    import torch
    case = HASHES[name]
    sph, ns, _ = _scene(apt, case, torch)
    assert sum(c for _, c in case["ranges"]) == case["w"] * case["h"]       # the ranges tile the frame
    for flags in flag_sets:
        fb, u8 = apt.render.render_frame(_params(apt, case, ns, flags=flags), sph)
        torch.cuda.synchronize()
        _check_ranges(case, fb, u8, 0, f"{name} flags={flags}")


@pytest.mark.gpu
def test_c2_whole_frame_k_mode(apt):
    """BASELINE configs[1], all 2 073 600 pixels / 4.25 G segments: the benchmarked kernel (no flags) and its
    compaction variant (APT_FLAG_RETIRE, result preserving)."""
    _whole_frame(apt, "C2", [0, apt.APT_FLAG_RETIRE])


@pytest.mark.gpu
def test_c2_o_mode_ranges(apt):
    import torch
    case = HASHES["C2_omode"]
    sph, ns, _ = _scene(apt, case, torch)
    for b, c in case["ranges"]:
        fb, u8 = apt.render.render_frame(_params(apt, case, ns), sph, b, c)
        torch.cuda.synchronize()
        _check_ranges(case, fb, u8, b, "C2 O-mode")


@pytest.mark.gpu
def test_c5_as_named_32_bounces_with_russian_roulette(apt):
    """BASELINE configs[4] as named: 1080p, 256 spp, depth 32, APT_FLAG_RR (rr_start 3) -- with the wave-queue
    compaction (the timed C5 configuration) and without it."""
    _whole_frame(apt, "C5_rr", [apt.APT_FLAG_RR | apt.APT_FLAG_RETIRE, apt.APT_FLAG_RR])


@pytest.mark.gpu
def test_c5_32_bounces_full_trace(apt):
    _whole_frame(apt, "C5_full", [0, apt.APT_FLAG_RETIRE])


@pytest.mark.gpu
def test_c3_band_edges_of_all_eight_ranks(apt):
    """BASELINE configs[2] (4096x4096, 1024 spp): each of the 8 ranks' bands rendered as that rank would render it
    (one launch per band, 17.2 G segments each), first and last image column compared."""
    import torch
    from ascendpathtracing_amd import dist as apt_dist
    case = HASHES["C3_bands"]
    sph, ns, _ = _scene(apt, case, torch)
    npix = case["w"] * case["h"]
    for r in range(8):
        b, c = apt_dist.split_range(npix, r, 8)
        fb, u8 = apt.render.render_frame(_params(apt, case, ns), sph, b, c)
        torch.cuda.synchronize()
        _check_ranges(case, fb, u8, b, f"C3 band {r}")
        if r == 5:       # round 3: every pixel of this band (16 chunks that tile it) ...
            whole = HASHES["C3_band5_whole"]
            assert whole["ranges"][0][0] == b and sum(n for _, n in whole["ranges"]) == c
            _check_ranges(whole, fb, u8, b, "C3 band 5, all pixels")
        _check_ranges(HASHES["C3_spread"], fb, u8, b, f"C3 band {r}, spread interior ranges")   # ... and interior ranges of every band
    assert len(case["ranges"]) == 16 and len(HASHES["C3_spread"]["ranges"]) == 64


@pytest.mark.gpu
def test_c4_ten_thousand_spheres(apt):
    """BASELINE configs[3]: a whole 480x270 frame by brute force (LDS tiles) and through the grid, then the real
    1080p / 256 spp frame through the grid (ranges hashed) and by brute force on those ranges only."""
    import torch
    case = HASHES["C4_crop"]
    sph, ns, grid = _scene(apt, case, torch)
    for accel, flags in ((0, 0), (grid.data_ptr(), 0), (grid.data_ptr(), apt.APT_FLAG_RETIRE)):
        fb, u8 = apt.render.render_frame(_params(apt, case, ns, flags=flags, accel=accel), sph)
        torch.cuda.synchronize()
        _check_ranges(case, fb, u8, 0, f"C4 crop accel={bool(accel)} flags={flags}")
    case = HASHES["C4_1080p"]
    fb, u8 = apt.render.render_frame(_params(apt, case, ns, accel=grid.data_ptr()), sph)
    torch.cuda.synchronize()
    _check_ranges(case, fb, u8, 0, "C4 1080p grid")
    _check_ranges(HASHES["C4_1080p_spread"], fb, u8, 0, "C4 1080p grid, 64 more ranges")
    _check_ranges(HASHES["C4_1080p_spread2"], fb, u8, 0, "C4 1080p grid, 384 more ranges (round 4: 98 304 pixels)")
    for b, c in case["ranges"]:
        fb, u8 = apt.render.render_frame(_params(apt, case, ns), sph, b, c)
        torch.cuda.synchronize()
        _check_ranges(case, fb, u8, b, "C4 1080p brute force")


@pytest.mark.gpu
def test_c4_whole_1080p_frame_three_traversals_agree(apt):
    """Every pixel of the real C4 frame (10 000 spheres, 1080p, 256 spp): the sample-queue kernel's grid form (with and without
    retirement), the nested item walk and the brute-force traversal over LDS tiles give the same floats and bytes (the oracle
    pins 128 ranges of it above; a CPU pass over the whole frame would take hours)."""
    import torch
    case = HASHES["C4_1080p"]
    sph, ns, grid = _scene(apt, case, torch)
    fb_q, u8_q = apt.render.render_frame(_params(apt, case, ns, accel=grid.data_ptr()), sph)
    fb_r, u8_r = apt.render.render_frame(_params(apt, case, ns, accel=grid.data_ptr(), flags=apt.APT_FLAG_RETIRE), sph)
    with apt.render.debug_knob("grid_walk", 1):
        fb_n, u8_n = apt.render.render_frame(_params(apt, case, ns, accel=grid.data_ptr()), sph)
    fb_b, u8_b = apt.render.render_frame(_params(apt, case, ns), sph)            # brute force: ~11 s
    torch.cuda.synchronize()
    for name, (f, u) in {"retire": (fb_r, u8_r), "nested": (fb_n, u8_n), "brute force": (fb_b, u8_b)}.items():
        assert torch.equal(f.view(torch.int32), fb_q.view(torch.int32)) and torch.equal(u, u8_q), name
    _check_ranges(case, fb_q, u8_q, 0, "C4 1080p grid form")


def test_committed_hashes_are_the_oracles(oracle):
    """CPU: the committed file has every BASELINE configuration and still equals what the oracle computes
    (re-run on the cheapest ranges: two C3 band edges, one C2 O-mode range, one C2 chunk prefix is too large)."""
    assert set(HASHES) >= {"C2", "C2_omode", "C5_rr", "C5_full", "C3_bands", "C4_crop", "C4_1080p"}
    for name, case in HASHES.items():
        assert len(case["ranges"]) == len(case["fb_sha256"]) == len(case["u8_sha256"]), name
    for name, picks in (("C3_bands", [0, 15]), ("C2_omode", [3]), ("C3_spread", [7, 40])):
        case = HASHES[name]
        p = oracle.make_params(case["w"], case["h"], case["s"], depth=case["depth"],
                               mode=oracle.MODE_O if case["mode"] == "O" else oracle.MODE_K,
                               flags=case["flags"] | oracle.FLAG_RETIRE, seed=case["seed"])
        for k in picks:
            b, c = case["ranges"][k]
            fb, u8, _, _ = oracle.render_frame(p, oracle.gen_spheres(), pixel_begin=b, pixel_count=c,
                                               threads=min(8, oracle.max_threads()))
            assert sha(fb) == case["fb_sha256"][k] and sha(u8) == case["u8_sha256"][k], (name, k)


def _mt_cases():
    with open(os.path.join(ROOT, "tests", "golden", "fullsize_hashes.json")) as f:
        return json.load(f).get("mt_cases", {})


@pytest.mark.gpu
def test_exact_reference_pipeline_in_bands_c2_and_c3(apt):
    """(f)1: the reference's exact pipeline (MT19937 gen_rays -> O-mode render -> decode_color) band by band in a bounded
    buffer, against the oracle's hashes: the first 8192 pixels of C2, and the LAST image column of C3 (path indices
    around 1.7e10, generator state taken from the committed fixture instead of walking 1.1e8 blocks)."""
    import torch
    cases = _mt_cases()
    assert set(cases) >= {"C2_mt_first_band", "C3_mt_last_column", "C2_mt_whole"}
    for name, case in cases.items():
        if case.get("chain"):
            continue                               # the whole-frame case is checked by the banded-vs-unbanded test below
        state = None
        if case["mt_state"]:
            f = np.load(os.path.join(ROOT, "tests", "golden", case["mt_state"]))
            state = (int(f["block"]), f["state"])
        for k, (b, c) in enumerate(case["ranges"]):
            torch.cuda.reset_peak_memory_stats()
            base = torch.cuda.memory_allocated()
            fb, u8, _ = apt.render.render_reference_frame(case["w"], case["h"], case["s"], depth=case["depth"], seed=0,
                                                          band_pixels=1024, pixel_begin=b, pixel_count=c, mt_state=state)
            torch.cuda.synchronize()
            assert torch.cuda.max_memory_allocated() - base < (1 << 30)            # bounded intermediates
            assert sha(fb.cpu().numpy()) == case["fb_sha256"][k] and sha(u8.cpu().numpy()) == case["u8_sha256"][k], (name, k)


@pytest.mark.gpu
def test_exact_reference_pipeline_fused_kernel_at_full_size(apt):
    """Round 3: the fused MT19937 frame kernel (apt_render_frame_mt) on the WHOLE C2 frame against the oracle's hashes of the
    reference's exact pipeline, and on the last image column of C3 from the committed generator state."""
    import torch
    cases = _mt_cases()
    whole = cases["C2_mt_whole"]
    w, h, s = whole["w"], whole["h"], whole["s"]
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    fb, u8 = apt.render.render_reference_frame_fused(w, h, s, depth=whole["depth"], seed=0)
    torch.cuda.synchronize()
    assert torch.cuda.max_memory_allocated() - base < (1 << 28)                  # the frame and the generator states: no 19 GB of rays
    fbh, u8h = fb.cpu().numpy(), u8.cpu().numpy()
    bad = [k for k, (b, c) in enumerate(whole["ranges"])
           if sha(fbh[:, b:b + c]) != whole["fb_sha256"][k] or sha(u8h[b:b + c]) != whole["u8_sha256"][k]]
    assert not bad, bad
    case = cases["C3_mt_last_column"]
    f = np.load(os.path.join(ROOT, "tests", "golden", case["mt_state"]))
    b, c = case["ranges"][0]
    fb, u8 = apt.render.render_reference_frame_fused(case["w"], case["h"], case["s"], depth=case["depth"], seed=0, pixel_begin=b, pixel_count=c,
                                                     mt_state=(int(f["block"]), f["state"]))
    torch.cuda.synchronize()
    assert sha(fb.cpu().numpy()) == case["fb_sha256"][0] and sha(u8.cpu().numpy()) == case["u8_sha256"][0]


@pytest.mark.gpu
def test_exact_reference_pipeline_whole_c2_banded_equals_unbanded(apt):
    """render_reference_frame at C2 in bands of 65536 pixels (0.6 GB of intermediates instead of
    19 GB) is bit-equal to the three whole-frame launches."""
    import torch
    w, h, s = 1920, 1080, 64
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    fb_b, u8_b, _ = apt.render.render_reference_frame(w, h, s, depth=8, seed=0, band_pixels=65536)
    torch.cuda.synchronize()
    peak = torch.cuda.max_memory_allocated() - base
    assert peak < (1 << 30), peak
    fb, u8, colors = apt.render.render_reference_frame(w, h, s, depth=8, seed=0)
    torch.cuda.synchronize()
    del colors
    assert torch.equal(fb.view(torch.int32), fb_b.view(torch.int32)) and torch.equal(u8, u8_b)
    case = _mt_cases()["C2_mt_first_band"]
    b, c = case["ranges"][0]
    assert sha(fb[:, b:b + c].cpu().numpy()) == case["fb_sha256"][0] and sha(u8[b:b + c].cpu().numpy()) == case["u8_sha256"][0]
    whole = _mt_cases()["C2_mt_whole"]            # all 2 073 600 pixels of the reference's exact pipeline against the oracle
    fbh, u8h = fb.cpu().numpy(), u8.cpu().numpy()
    assert sum(c for _, c in whole["ranges"]) == w * h
    bad = [k for k, (b, c) in enumerate(whole["ranges"])
           if sha(fbh[:, b:b + c]) != whole["fb_sha256"][k] or sha(u8h[b:b + c]) != whole["u8_sha256"][k]]
    assert not bad, bad
