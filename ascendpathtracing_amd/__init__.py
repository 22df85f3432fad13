"""MI355X-native drop-in for the hot path of KVM-Explorer/AscendPathTracing.

    ray-generate -> ray/sphere intersect -> mirror-reflect / throughput loop -> colour -> PPM

The compute lives in ``librender_mi355x.so`` (hand-written gfx950 HIP kernels behind the
C-ABI of ``include/render_mi355x.h``); this package is the host-side mirror of the
reference's own interfaces for that path:

    render              render_do / render_do_ex / render_frame   (src/main.cpp, src/render.cpp)
    gen_data            gen_rays, gen_spheres, gen_scene           (scripts/gen_data.py)
    data_visualization  decode_color, write_ppm                    (scripts/data_visualization.py)
    dist                pixel-range sharding over ranks + one framebuffer gather (RCCL)

There is no CPU fallback: every compute entry raises if the HIP library or a GPU is missing.
"""
from . import _lib  # noqa: F401
from ._lib import (APT_FLAG_RETIRE, APT_FLAG_RR, APT_FLAG_EMISSION, APT_FLAG_BAND_BUFFERS, APT_FLAG_GRID_SLOTS, APT_MODE_KERNEL, APT_MODE_ORACLE, RenderParams, AptError, default_params,
                   make_params)

__all__ = ["APT_FLAG_RETIRE", "APT_FLAG_RR", "APT_FLAG_EMISSION", "APT_FLAG_BAND_BUFFERS", "APT_FLAG_GRID_SLOTS", "APT_MODE_KERNEL", "APT_MODE_ORACLE", "RenderParams", "AptError", "default_params",
           "make_params"]
