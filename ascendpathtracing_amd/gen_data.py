"""Host-side input writers, the counterpart of the reference's scripts/gen_data.py:
gen_rays (:21-75), gen_spheres (:92-132).  The arithmetic runs in librender_mi355x.so
(apt_gen_rays_host / apt_gen_spheres_host / apt_gen_scene_host); outputs are bit-identical
to the reference's rays.bin / spheres.bin for the same (w, h, s, seed)."""
import ctypes
import os

import numpy as np

from ._lib import check, lib

width, height, samples = 16, 16, 1      # gen_data.py:6-8
eps, bounceMax = 1e-4, 5                # gen_data.py:9-10


def _fptr(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def gen_rays(w, h, s, seed=0, out_dir=None):
    """-> float32 [6][N] planes ox,oy,oz,dx,dy,dz; writes <out_dir>/rays.bin when given
    (the reference always writes ./input/rays.bin, gen_data.py:71)."""
    n = w * h * 4 * s
    rays = np.empty(6 * n, dtype=np.float32)
    check(lib().apt_gen_rays_host(ctypes.c_uint32(w), ctypes.c_uint32(h), ctypes.c_uint32(s), ctypes.c_uint32(seed),
                                  _fptr(rays)), "apt_gen_rays_host")
    if out_dir is not None:
        rays.tofile(os.path.join(out_dir, "rays.bin"))
    return rays.reshape(6, n)


def gen_spheres(out_dir=None):
    """-> the 128-float (512-byte) [10][8] table; writes <out_dir>/spheres.bin when given."""
    sph = np.zeros(128, dtype=np.float32)
    check(lib().apt_gen_spheres_host(_fptr(sph)), "apt_gen_spheres_host")
    if out_dir is not None:
        sph.tofile(os.path.join(out_dir, "spheres.bin"))
    return sph


def gen_scene(num_spheres, seed=0, out_dir=None):
    """Build-defined large scene (BASELINE config 4): six walls, Ns-7 random small spheres,
    light at index Ns-1.  -> zero-padded [10][Ns] table."""
    n = ctypes.c_size_t(0)
    check(lib().apt_gen_scene_host(ctypes.c_uint32(num_spheres), ctypes.c_uint64(seed), None, ctypes.byref(n)),
          "apt_gen_scene_host")
    sph = np.zeros(n.value, dtype=np.float32)
    check(lib().apt_gen_scene_host(ctypes.c_uint32(num_spheres), ctypes.c_uint64(seed), _fptr(sph), None),
          "apt_gen_scene_host")
    if out_dir is not None:
        sph.tofile(os.path.join(out_dir, "spheres.bin"))
    return sph


if __name__ == "__main__":              # gen_data.py:435-446
    os.makedirs("input", exist_ok=True)
    gen_rays(width, height, samples, seed=0, out_dir="./input")
    gen_spheres(out_dir="./input")
    print("===========Python Script Done=============")
