"""CPU-only: the C-ABI library loads and exports every symbol include/render_mi355x.h
declares; the host-side helpers (no GPU needed) reproduce the reference's input files; the
compute entry points refuse to run without a GPU instead of falling back."""
import ctypes
import hashlib
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def apt():
    import __graft_entry__ as g
    g.build()
    import ascendpathtracing_amd as pkg
    from ascendpathtracing_amd import gen_data
    pkg.gen_data = gen_data
    return pkg


def test_every_declared_symbol_is_exported(apt):
    hdr = open(os.path.join(ROOT, "include", "render_mi355x.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(render_do(?:_ex)?|render_frame|apt_[a-z0-9_]+)\s*\(", hdr))
    declared.discard("apt_render_params")
    assert {"render_do", "render_do_ex", "render_frame", "apt_gen_rays_host", "apt_write_ppm"} <= declared
    h = ctypes.CDLL(apt._lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(h, name), f"{name} declared in include/render_mi355x.h but not exported"
    assert set(apt._lib.ABI_SYMBOLS) == declared
    assert h.apt_abi_version() == 3


def test_the_library_exports_its_c_abi_and_nothing_else(apt):
    """-fvisibility=hidden + csrc/apt_exports.map -- `nm -D --defined-only` shows the symbols the header declares
    and the C++-mangled render_do of src/main.cpp:9-10, no context internals (apt_context::*, apt::default_context, apt::set_error)."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", apt._lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {line.split()[-1] for line in out.splitlines() if line.strip()}
    assert exported == set(apt._lib.ABI_SYMBOLS) | {apt._lib.CXX_RENDER_DO}, sorted(exported ^ (set(apt._lib.ABI_SYMBOLS) | {apt._lib.CXX_RENDER_DO}))
    assert not [s_ for s_ in exported if s_.startswith("_ZN")]


def test_no_launch_path_reads_the_process_environment(apt):
    """getenv() appears once in the library -- where an apt_context is created (host_helpers.cpp) -- and in no
    launch path (render_kernels.hip, the kernels' headers)."""
    csrc = os.path.join(ROOT, "ascendpathtracing_amd", "csrc")
    for f in os.listdir(csrc):
        if f.endswith((".hip", ".h")) or f == "render_do_cxx.cpp":
            src = re.sub(r"//.*", "", open(os.path.join(csrc, f)).read())
            assert "getenv" not in src, f
    body = re.sub(r"//.*", "", open(os.path.join(csrc, "host_helpers.cpp")).read())
    ctor = body[body.index("double env_number"):body.index("apt_context::Values apt_context::snapshot()")]
    assert body.count("getenv") == ctor.count("getenv") == 2


def test_debug_knobs_argument_checks_without_a_gpu(apt):
    lib = apt._lib.lib()
    ctx = ctypes.c_void_p(lib.apt_context_create())
    assert lib.apt_context_set_debug(ctx, b"queue_ppw", ctypes.c_double(16)) == 0
    assert lib.apt_context_set_debug(ctx, b"queue_ppw", ctypes.c_double(0)) == 0
    assert lib.apt_context_set_debug(ctx, b"grid_spheres_per_cell", ctypes.c_double(0.7)) == 0
    for key, v in ((b"queue_ppw", 4097), (b"queue_ppw", 2.5), (b"queue_nbuf", 1), (b"queue_nbuf", 17), (b"grid_walk", 2), (b"nope", 1),
                   (b"grid_spheres_per_cell", 0.001)):
        assert lib.apt_context_set_debug(ctx, key, ctypes.c_double(v)) == 1 and lib.apt_last_status() == 1, (key, v)
    assert lib.apt_context_set_debug(None, b"queue_ppw", ctypes.c_double(1)) == 1
    assert lib.apt_context_set_debug(ctx, None, ctypes.c_double(1)) == 1
    lib.apt_context_destroy(ctx)


def test_debug_knob_reads_back_and_restores_what_it_found(apt):
    """debug_knob reset a knob to 0 on exit, losing an initial value from the environment or an enclosing block."""
    from ascendpathtracing_amd import render
    lib = apt._lib.lib()
    ctx = render.Context()
    ctx.set_debug("queue_ppw", 24)
    ctx.set_debug("grid_spheres_per_cell", 0.7)
    assert ctx.get_debug("queue_ppw") == 24 and ctx.get_debug("grid_spheres_per_cell") == 0.7 and ctx.get_debug("grid_walk") == 0
    with pytest.raises(apt.AptError):
        ctx.get_debug("nope")
    v = ctypes.c_double(-1)
    assert lib.apt_context_get_debug(None, b"queue_ppw", ctypes.byref(v)) == 1 and lib.apt_get_debug(b"queue_ppw", None) == 1
    ctx.close()
    before = render.get_debug("queue_ppw")
    try:
        render.set_debug("queue_ppw", 8)
        with render.debug_knob("queue_ppw", 32):
            assert render.get_debug("queue_ppw") == 32
            with render.debug_knob("queue_ppw", 4):
                assert render.get_debug("queue_ppw") == 4
            assert render.get_debug("queue_ppw") == 32
        assert render.get_debug("queue_ppw") == 8
    finally:
        render.set_debug("queue_ppw", before)


def test_params_struct_layout_matches_header(apt):
    p = apt.default_params()
    assert ctypes.sizeof(apt.RenderParams) == 80 == p.struct_size
    # the reference's compile-time constants: common.h:4-10, render.cpp:141,194
    assert (p.width, p.height, p.samples, p.depth, p.num_spheres, p.light_index) == (16, 16, 1, 5, 8, 7)
    assert np.float32(p.eps) == np.float32(1e-4) and p.gain == 12.0 and p.mode == apt.APT_MODE_KERNEL and p.flags == 0


def test_host_gen_rays_and_spheres_are_the_reference_files(apt, golden):
    data, meta = golden
    for key, (w, h, s) in {"16x16_s1": (16, 16, 1), "16x16_s2": (16, 16, 2), "24x16_s1": (24, 16, 1),
                           "64x64_s1": (64, 64, 1)}.items():
        rays = apt.gen_data.gen_rays(w, h, s, seed=0)
        assert hashlib.sha256(rays.tobytes()).hexdigest() == meta["cases"][key]["rays_sha256"]
        assert np.array_equal(rays.ravel().view(np.uint32), data[f"{key}_rays"].view(np.uint32))
    sph = apt.gen_data.gen_spheres()
    assert hashlib.sha256(sph.tobytes()).hexdigest() == meta["cases"]["16x16_s1"]["spheres_sha256"]


def test_file_contract_of_gen_data(apt, tmp_path, golden):
    _, meta = golden
    apt.gen_data.gen_rays(16, 16, 1, seed=0, out_dir=str(tmp_path))
    apt.gen_data.gen_spheres(out_dir=str(tmp_path))
    assert os.path.getsize(tmp_path / "rays.bin") == 16 * 16 * 4 * 6 * 4       # main.cpp:47
    assert os.path.getsize(tmp_path / "spheres.bin") == 512                    # main.cpp:48
    assert hashlib.sha256((tmp_path / "rays.bin").read_bytes()).hexdigest() == meta["cases"]["16x16_s1"]["rays_sha256"]


def test_scene_generator_matches_restatement_and_keeps_walls_and_light(apt, oracle):
    for ns in (8, 9, 100, 10000):
        a, b = apt.gen_data.gen_scene(ns, seed=42), oracle.gen_scene(ns, seed=42)
        assert a.size % 128 == 0 and np.array_equal(a.view(np.uint32), b.view(np.uint32))
        tab = a[:10 * ns].reshape(10, ns)
        ref = apt.gen_data.gen_spheres()[:80].reshape(10, 8)
        assert np.array_equal(tab[:, :6], ref[:, :6]) and np.array_equal(tab[:, ns - 1], ref[:, 7])
        if ns > 8:
            mid = tab[:, 6:ns - 1]
            assert (mid[0] >= 0.25).all() and (mid[0] <= 4.0).all() and (mid[7:10] >= 0.1).all()
    with pytest.raises(apt.AptError):
        apt.gen_data.gen_scene(7)


def test_write_ppm_matches_reference_bytes(apt, golden, tmp_path):
    from ascendpathtracing_amd import data_visualization as dv
    data, meta = golden
    for name, m in meta["decode"].items():
        w, h = m["w"], m["h"]
        u8 = data[f"decode_{name}_u8"][:, ::-1, :].reshape(-1, 3)    # undo the y flip -> x-major pixels
        out = tmp_path / f"{name}.ppm"
        dv.write_ppm(w, h, u8, str(out))
        assert out.read_bytes() == data[f"decode_{name}_ppm"].tobytes()
    dv.write_ppm(3, 2, np.arange(18, dtype=np.uint8), str(tmp_path / "ns.ppm"))   # non-square: h rows of w pixels
    lines = (tmp_path / "ns.ppm").read_text().splitlines()
    assert lines[:3] == ["P3", "3 2", "255"] and len(lines) == 5
    assert lines[3].split() == ["3", "4", "5", "9", "10", "11", "15", "16", "17"]   # top row = y 1


def test_compute_entries_fail_loudly_without_a_gpu(apt):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from ascendpathtracing_amd import render
    with pytest.raises(apt.AptError, match="no HIP device"):
        render.render_frame(apt.make_params(), torch.zeros(128))
    with pytest.raises(apt.AptError, match="no HIP device"):
        render.render_do(8, None, 0, torch.zeros(6), torch.zeros(128), torch.zeros(3))
    assert apt._lib.lib().apt_device_count() == 0


def test_grid_builder_layout(apt):
    """apt_build_grid_host: header, always-tested list (walls + light), every small sphere listed in every
    cell its inflated sphere touches (checked per cell against the definition), ascending sphere order inside a cell."""
    ns = 500
    scene = apt.gen_data.gen_scene(ns, seed=4)
    g = apt.gen_data.build_grid(scene, ns)
    hdr = g[:12]
    assert hdr[0] == 0x47524944 and hdr[1] == ns
    n, ncells, nlarge, nitems = hdr[2:5], hdr[5], hdr[6], hdr[7]
    off_large, off_cells, off_items, off_geom = hdr[8:12]
    assert int(np.prod(n)) == ncells and nlarge == 7
    assert sorted(g[off_large:off_large + nlarge].tolist()) == [0, 1, 2, 3, 4, 5, ns - 1]
    starts = g[off_cells:off_cells + ncells + 1]
    assert starts[0] == 0 and starts[-1] == nitems and (np.diff(starts.astype(np.int64)) >= 0).all()
    items = g[off_items:off_items + nitems]
    for c in range(0, ncells, max(1, ncells // 50)):
        lst = items[starts[c]:starts[c + 1]]
        assert (np.diff(lst.astype(np.int64)) > 0).all()
    assert set(items.tolist()) == set(range(6, ns - 1))
    # the lists against their definition: sphere k is in cell c iff the box of c is within rad + margin of the centre (pt_core.h grid_cell_touches)
    fl = g[13:26].view(np.float32)
    gmin, cellw, margin = fl[0:3].astype(np.float64), fl[6:9].astype(np.float64), float(fl[12])
    tab64 = scene[:10 * ns].reshape(10, ns).astype(np.float64)
    for c in range(0, ncells, max(1, ncells // 40)):
        z, rem_ = divmod(c, int(n[0]) * int(n[1])); y, x = divmod(rem_, int(n[0]))
        lo = gmin + np.array([x, y, z]) * cellw
        hi = lo + cellw
        ctr = tab64[1:4, 6:ns - 1]
        d = np.maximum(np.maximum(lo[:, None] - ctr, ctr - hi[:, None]), 0.0)
        dist = np.sqrt((d * d).sum(axis=0))
        reach = np.sqrt(tab64[0, 6:ns - 1]) + margin
        listed = set(items[starts[c]:starts[c + 1]].tolist())
        must = set((np.nonzero(dist <= reach * 0.999)[0] + 6).tolist())      # clearly touching: must be listed
        may = set((np.nonzero(dist <= reach * 1.01)[0] + 6).tolist())        # clearly apart: must not be
        assert must <= listed <= may, (c, sorted(must - listed), sorted(listed - may))
    geom = g[off_geom:off_geom + 4 * ns].view(np.float32).reshape(ns, 4)
    tab = scene[:10 * ns].reshape(10, ns)
    assert np.array_equal(geom[:, 0], tab[1]) and np.array_equal(geom[:, 3], tab[0])
    # round 3: the pair-slot tables of the sample-queue kernel's grid form (pt_core.h GridHeader)
    full = g[:32]
    off_cellslot, off_slots, off_slot_ids, nslots, slot_base, off_sphere8 = (int(x) for x in full[26:32])
    assert off_cellslot and off_slots % 8 == 0 and off_sphere8 % 8 == 0 and slot_base == (nlarge + 1) // 2
    slot_geom = g[off_slots:off_slots + 8 * nslots].reshape(nslots, 8)
    slot_ids = g[off_slot_ids:off_slot_ids + 2 * nslots].reshape(nslots, 2)
    gw = g[off_geom:off_geom + 4 * ns].reshape(ns, 4)                        # raw words (NaN-safe comparisons)

    def check_list(first_slot, ids):
        for k, sid in enumerate(ids):
            sl, half = first_slot + k // 2, k % 2
            assert slot_ids[sl, half] == sid and np.array_equal(slot_geom[sl, half::2], gw[sid])
        if len(ids) % 2:                                                       # odd list: a NaN sphere with no id pads the last slot
            sl = first_slot + len(ids) // 2
            assert slot_ids[sl, 1] == 0xffffffff and (slot_geom[sl, 1::2] == 0x7fc00000).all()

    check_list(0, g[off_large:off_large + nlarge].tolist())
    used = np.zeros(nslots, dtype=bool)
    used[:slot_base] = True
    # round 4: cellslot is indexed by BORDERED cell coordinates; the layer around the grid holds the "outside" mark (count field 62, slot 0)
    n0, n1, n2 = (int(x) for x in n)
    cs = g[off_cellslot:off_cellslot + (n0 + 2) * (n1 + 2) * (n2 + 2)].reshape(n2 + 2, n1 + 2, n0 + 2)
    border = np.ones(cs.shape, dtype=bool)
    border[1:-1, 1:-1, 1:-1] = False
    assert (cs[border] == 62).all() and off_slots >= off_cellslot + cs.size
    inner = cs[1:-1, 1:-1, 1:-1].reshape(-1)                                 # the grid's own cells, in linear cell order
    for c in range(ncells):
        b, e = int(starts[c]), int(starts[c + 1])
        entry = int(inner[c])
        first, cnt = entry >> 7, entry & 63
        n_real = (e - b + 1) // 2
        assert first == slot_base + ((b + c + 1) >> 1) and cnt == (n_real if n_real < 62 else 63)
        n_sl = (e - b + 1) // 2
        assert not used[first:first + n_sl].any() and first + n_sl <= nslots   # lists never overlap
        used[first:first + n_sl] = True
        if c % max(1, ncells // 50) == 0:
            check_list(first, items[b:e].tolist())
    s8 = g[off_sphere8:off_sphere8 + 8 * ns].view(np.float32).reshape(ns, 8)
    assert np.array_equal(s8[:, :3].T, tab[1:4]) and np.array_equal(s8[:, 3], tab[0]) and np.array_equal(s8[:, 4:7].T, tab[7:10])


def test_missing_library_fails_loudly(apt, monkeypatch, tmp_path):
    """No silent fallback: without librender_mi355x.so every entry raises AptError."""
    from ascendpathtracing_amd import _lib, gen_data
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "librender_mi355x.so"))
    with pytest.raises(apt.AptError, match="is missing"):
        gen_data.gen_spheres()
    with pytest.raises(apt.AptError, match="is missing"):
        apt.default_params()


def test_run_sh_refuses_cpu_mode():
    import subprocess
    r = subprocess.run(["bash", os.path.join(ROOT, "run.sh"), "-r", "cpu", "-v", "Ascend310P1"], capture_output=True, text=True)
    assert r.returncode != 0 and "no CPU fallback" in r.stdout


def test_mt19937_checkpoints_match_numpy_generator_state(apt):
    """apt_mt19937_checkpoints_host: checkpoint i must be NumPy's legacy MT19937 key array after
    (i*stride + 1) * 624 words have been drawn (np.random.seed / rand of gen_data.py:438,37)."""
    for seed in (0, 12345):
        ck = apt.gen_data.mt19937_checkpoints(156 * 10, seed=seed, stride=3)      # 10 blocks -> checkpoints 0, 3, 6, 9
        assert ck.shape == (4, 624)
        rs = np.random.RandomState(seed)
        for i, blk in enumerate((0, 3, 6, 9)):
            rs2 = np.random.RandomState(seed)
            rs2.random_sample((blk + 1) * 312)                                   # 2 words per double
            key, pos = rs2.get_state()[1], rs2.get_state()[2]
            assert pos == 624 and np.array_equal(ck[i], key), (seed, blk)
    with pytest.raises(apt.AptError):
        apt._lib.check(apt._lib.lib().apt_mt19937_checkpoints_host(0, 0, 1, None), "checkpoints")


def test_c_abi_argument_validation_needs_no_gpu(apt):
    """Every device entry validates its arguments before touching HIP: the error codes of
    include/render_mi355x.h come back on a machine without a GPU as well."""
    L = apt._lib.lib()
    p = apt.default_params()
    one = ctypes.c_void_p(16)      # never dereferenced: validation fails first
    bad = apt.default_params(); bad.struct_size = 8
    assert L.render_do_ex(ctypes.byref(bad), None, one, one, one) == 2                      # APT_ERR_STRUCT
    assert L.render_do_ex(None, None, one, one, one) == 1                                   # APT_ERR_ARG
    assert L.render_do_ex(ctypes.byref(p), None, None, one, one) == 1
    z = apt.make_params(0, 16, 1)
    assert L.render_do_ex(ctypes.byref(z), None, one, one, one) == 1
    s0 = apt.make_params(16, 16, 1, num_spheres=8, light_index=9)
    assert L.render_do_ex(ctypes.byref(s0), None, one, one, one) == 3                       # APT_ERR_SCENE
    assert b"light_index" in L.apt_last_error()
    rng = apt.make_params(16, 16, 1, path_begin=1000, path_count=100)
    assert L.render_do_ex(ctypes.byref(rng), None, one, one, one) == 1
    assert L.render_frame(ctypes.byref(p), None, one, ctypes.c_uint64(0), ctypes.c_uint64(10 ** 9), one, None) == 1
    big = apt.make_params(16, 16, 1 << 20)                                                  # pairwise plan too long
    assert L.render_frame(ctypes.byref(big), None, one, ctypes.c_uint64(0), ctypes.c_uint64(1), one, None) == 1
    em = apt.make_params(16, 16, 1, light_index=-1, flags=apt.APT_FLAG_EMISSION)
    assert L.render_do_ex(ctypes.byref(em), None, one, one, one) == 3
    assert L.apt_set_refill_lanes(0) == 1 and L.apt_set_refill_lanes(65) == 1 and L.apt_set_refill_lanes(32) == 0
    assert L.apt_gen_rays_mt_device(ctypes.byref(p), None, one, ctypes.c_uint32(4), ctypes.c_uint64(0), one) == 1   # too few checkpoints
    assert L.apt_build_grid_host(None, 8, None, None) == 1
    e = apt.make_params(16, 16, 1, path_begin=1024, path_count=0)                           # empty range is a no-op, not an error
    assert L.render_do_ex(ctypes.byref(e), None, one, one, one) == 0


def test_header_is_plain_c_and_links_from_c(apt, tmp_path):
    """include/render_mi355x.h must be usable from a C99 translation unit (the reference's host is C++, a cgo / FFI
    binding wants plain C): compile with -pedantic, link against the library, run the host-only entry points."""
    import subprocess
    src = tmp_path / "abi_c.c"
    src.write_text('#include "render_mi355x.h"\n#include <stdio.h>\n'
                   'int main(void) {\n  apt_render_params p; float sph[128];\n  apt_default_params(&p);\n'
                   '  if (apt_gen_spheres_host(sph)) return 2;\n'
                   '  printf("%u %u %u %u %d %g\\n", p.width, p.height, p.samples, p.depth, apt_abi_version(), sph[6]);\n'
                   '  return p.struct_size == sizeof p ? 0 : 1;\n}\n')
    libdir = os.path.dirname(apt._lib.LIB_PATH)
    exe = tmp_path / "abi_c"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"), str(src),
                    "-o", str(exe), "-L", libdir, "-lrender_mi355x", "-Wl,-rpath," + libdir], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    assert out == ["16", "16", "1", "5", "3", "272.25"]


# The reference's own declaration of the boundary, verbatim from src/main.cpp:9-10 (an interface, not code):
# C++ linkage, no extern "C".  A translation unit holding it must link against the library unchanged.
REF_DECL_TU = r"""
#include <stdint.h>
extern void render_do(uint32_t coreDim, void *l2ctrl, void *stream,
                      uint8_t *rays, uint8_t *spheres,uint8_t *colors);
extern "C" void call_through_reference_declaration(uint32_t coreDim, void *stream, uint8_t *rays, uint8_t *spheres,
                                                   uint8_t *colors) {
    render_do(coreDim, nullptr, stream, rays, spheres, colors);   // src/main.cpp:74
}
"""


def build_reference_decl_shim(apt, out_dir):
    """g++-compiles REF_DECL_TU into a shared object linked against librender_mi355x.so -> its path."""
    import subprocess
    src = os.path.join(out_dir, "ref_decl_tu.cpp")
    so = os.path.join(out_dir, "libref_decl_tu.so")
    with open(src, "w") as f:
        f.write(REF_DECL_TU)
    libdir = os.path.dirname(apt._lib.LIB_PATH)
    subprocess.run(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-Wl,--no-undefined", src, "-o", so, f"-L{libdir}",
                    "-lrender_mi355x", f"-Wl,-rpath,{libdir}"], check=True)
    return so


def test_reference_cxx_declaration_links_unmodified(apt, tmp_path):
    """src/main.cpp:9-10 declares render_do with C++ linkage: the library exports the mangled symbol, so a TU with
    that declaration links (--no-undefined) without touching the declaration."""
    import subprocess
    so = build_reference_decl_shim(apt, str(tmp_path))
    undefined = subprocess.run(["nm", "-D", "--undefined-only", so], capture_output=True, text=True, check=True).stdout
    assert "_Z9render_dojPvS_PhS0_S0_" in undefined            # the TU really binds the C++-mangled name
    exported = subprocess.run(["nm", "-D", "--defined-only", apt._lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert " T _Z9render_dojPvS_PhS0_S0_" in exported and " T render_do\n" in exported and " T apt_render_do" in exported


def test_error_record_is_per_call(apt):
    """apt_last_error()/apt_last_status() describe the LAST call on this thread: a failure must not stick."""
    h = apt._lib.lib()
    bad = apt.default_params()
    bad.num_spheres = 0
    assert h.apt_set_default_params(ctypes.byref(bad)) != 0
    assert h.apt_last_status() != 0 and b"num_spheres" in h.apt_last_error()
    assert h.apt_gen_spheres_host(None) == 1 and b"apt_gen_spheres_host" in h.apt_last_error()   # host helpers set it too
    buf = np.zeros(128, dtype=np.float32)
    assert h.apt_gen_spheres_host(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    assert h.apt_last_status() == 0 and h.apt_last_error() == b""
    assert h.apt_write_ppm(b"/nonexistent_dir/x.ppm", 1, 1, buf.ctypes.data_as(ctypes.c_void_p)) == 5
    assert b"/nonexistent_dir/x.ppm" in h.apt_last_error()
    assert h.apt_set_default_params(ctypes.byref(apt.default_params())) == 0 and h.apt_last_error() == b""


def test_contexts_are_independent_and_thread_safe(apt):
    """Setters of one context do not leak into another or into the default one; concurrent setters on one
    context do not corrupt it (no GPU needed: only the host-side state is exercised)."""
    import threading
    h = apt._lib.lib()
    h.apt_context_create.restype = ctypes.c_void_p
    a, b = ctypes.c_void_p(h.apt_context_create()), ctypes.c_void_p(h.apt_context_create())
    assert a and b
    pa = apt.make_params(64, 32, 2, depth=7)
    assert h.apt_context_set_params(a, ctypes.byref(pa)) == 0
    bad = apt.default_params()
    bad.struct_size = 4
    assert h.apt_context_set_params(b, ctypes.byref(bad)) == 2          # APT_ERR_STRUCT, b unchanged
    assert h.apt_context_set_refill_lanes(a, 0) == 1 and h.apt_context_set_refill_lanes(a, 64) == 0
    assert h.apt_context_set_params(None, ctypes.byref(pa)) == 1
    errs = []

    def hammer(ctx, k):
        for i in range(2000):
            p = apt.make_params(16 + k, 16 + k, 1 + (i & 3), depth=k + 1)
            if h.apt_context_set_params(ctx, ctypes.byref(p)) != 0 or h.apt_last_status() != 0:
                errs.append((k, i))
            q = apt.default_params()
            q.num_spheres = 0
            if h.apt_context_set_params(ctx, ctypes.byref(q)) != 3 or b"num_spheres" not in h.apt_last_error():
                errs.append(("err", k, i))

    ts = [threading.Thread(target=hammer, args=(a, k)) for k in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs
    h.apt_context_destroy(a)
    h.apt_context_destroy(b)


def test_grid_flags_reads_exactly_the_header_whatever_the_buffer_type(apt):
    """grid_flags() sliced 32 ELEMENTS of whatever it was given and apt_grid_flags copies 128 BYTES: a uint8 view or a short
    buffer was a host out-of-bounds read.  Now: any element type gives the same flags, a buffer shorter than a header raises."""
    import torch
    ns = 60
    scene = apt.gen_data.gen_scene(ns, seed=2)
    g = apt.gen_data.build_grid(scene, ns)
    want = apt.gen_data.grid_flags(g, ns)
    assert want == apt.APT_FLAG_GRID_SLOTS
    assert apt.gen_data.grid_flags(g.view(np.uint8), ns) == want
    assert apt.gen_data.grid_flags(g.view(np.uint8)[:128], ns) == want
    assert apt.gen_data.grid_flags(g.view(np.uint16), ns) == want
    assert apt.gen_data.grid_flags(torch.from_numpy(g.view(np.int32)), ns) == want          # (a CPU tensor takes the tensor path)
    assert apt.gen_data.grid_flags(g, ns + 1) == 0                                          # another scene's grid earns nothing
    for short in (g[:31], g.view(np.uint8)[:127], torch.from_numpy(g.view(np.int32))[:31]):
        with pytest.raises(apt.AptError, match="shorter than a grid header"):
            apt.gen_data.grid_flags(short, ns)
