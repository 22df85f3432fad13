"""Per-sample colours -> image, the counterpart of scripts/data_visualization.py:
decode_color (:20-59) runs on the GPU (apt_decode_color_device), write_ppm (:11-17) in the
library's host code.  No CPU fallback for decode_color."""
import ctypes
import sys

import numpy as np
import torch

from . import render
from ._lib import check, lib, make_params

width, height, samples = 16, 16, 1      # data_visualization.py:5-7


def write_ppm(w, h, data, path="./output/color.ppm"):
    """data: uint8 [W*H][3] in x-major pixel order (q = i*H + j), y not flipped."""
    data = np.ascontiguousarray(data, dtype=np.uint8).reshape(-1)
    if data.size != w * h * 3:
        raise ValueError("write_ppm: data does not match w*h*3")
    check(lib().apt_write_ppm(path.encode(), ctypes.c_uint32(w), ctypes.c_uint32(h),
                              data.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))), "apt_write_ppm")


def decode_color(path, w, h, s, out_path="./output/color.ppm", device="cuda"):
    """Read color.bin ([3][N] float32), average the 4*s samples of each pixel as the reference
    does, clip, scale to 8 bits, write the P3 file.  Returns the (w,h,3) uint8 array the
    reference's decode_color returns (second index already y-flipped)."""
    colors = np.fromfile(path, dtype=np.float32)
    if colors.size != 3 * w * h * 4 * s:
        raise ValueError(f"{path}: expected {3 * w * h * 4 * s} floats, found {colors.size}")
    p = make_params(w, h, s)
    fb, u8 = render.decode_color_device(p, torch.from_numpy(colors).to(device))
    torch.cuda.synchronize()
    u8 = u8.cpu().numpy()
    write_ppm(w, h, u8, out_path)
    return u8.reshape(w, h, 3)[:, ::-1, :].copy()


if __name__ == "__main__":              # data_visualization.py:93-100
    try:
        decode_color(sys.argv[1], width, height, samples)
        print("Generate Result Image")
    except Exception as e:              # noqa: BLE001 - mirrors the reference CLI
        print(e)
        sys.exit(1)
