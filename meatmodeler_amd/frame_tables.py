"""Fixed-point tables of the 8-bit BGR <-> CIE L*a*b* conversion used by `increaseContrast` (reference
processor.py:12-26 calls cv2.cvtColor(..., COLOR_BGR2LAB / COLOR_Lab2BGR)).  OpenCV's own 8-bit path is table driven
too, but its tables cannot be reproduced offline (OpenCV absent: parity unpinned), so the conversion is DEFINED here:
sRGB primaries, D65 white, 12-bit linear light, f(t) = cbrt(t) (t > 0.008856) else 7.787 t + 16/116.  The same
tables feed the HIP kernels (meatmodeler_amd/csrc/contrast.hip) and the CPU oracle (oracle/frame_oracle.c), whose
per-pixel arithmetic is integer only -- the two agree bit for bit.

  gamma     [256]  u16   round(4095 * srgb_decode(c / 255))
  cbrt_tab  [4096] u16   round(32768 * f(i / 4095))
  gamma_inv [4096] u8    round(255 * srgb_encode(i / 4095))
(f^-1 needs no table: f^3, or the linear branch, in 64-bit integer arithmetic.)
"""
import functools

import numpy as np


@functools.lru_cache(maxsize=1)
def lab_tables():
    c = np.arange(256) / 255.0
    lin = np.where(c <= 0.04045, c / 12.92, ((c + 0.055) / 1.055) ** 2.4)
    gamma = np.rint(4095.0 * lin).astype(np.uint16)
    t = np.arange(4096) / 4095.0
    f = np.where(t > 0.008856, np.cbrt(t), 7.787 * t + 16.0 / 116.0)
    cbrt_tab = np.rint(32768.0 * f).astype(np.uint16)
    u = np.arange(4096) / 4095.0
    enc = np.where(u <= 0.0031308, 12.92 * u, 1.055 * u ** (1 / 2.4) - 0.055)
    gamma_inv = np.rint(255.0 * np.clip(enc, 0.0, 1.0)).astype(np.uint8)
    return gamma, cbrt_tab, gamma_inv
