"""ctypes wrapper of oracle/frame_oracle.c (pyramidal LK, Shi-Tomasi corners, CLAHE / LAB / grey).  TEST INFRASTRUCTURE
ONLY -- see the header of the C file: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libframe_oracle.so")


def _lib():
    global _L
    try:
        return _L
    except NameError:
        pass
    src = os.path.join(_HERE, "frame_oracle.c")
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    _L = C.CDLL(_SO)
    _L.frc_good_features.restype = C.c_int
    return _L


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def pyr_down(img):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    out = np.zeros(((h + 1) // 2, (w + 1) // 2), np.uint8)
    _lib().frc_pyr_down(_p(img), C.c_int(w), C.c_int(h), C.c_int(w), _p(out), C.c_int(out.shape[1]))
    return out


def pyramid(img, max_level):
    levels = [np.ascontiguousarray(img, np.uint8)]
    for _ in range(max_level):
        levels.append(pyr_down(levels[-1]))
    return levels


def lk_track(prev, nxt, pts, win=(21, 21), max_level=3, max_count=30, epsilon=0.01):
    """cv2.calcOpticalFlowPyrLK(prev, nxt, pts, None, winSize=win, maxLevel=max_level, criteria=(3, max_count, epsilon))
    -> (next [n,2] f32, status [n] u8, err [n] f32)."""
    pp, pn = pyramid(prev, max_level), pyramid(nxt, max_level)
    L = max_level + 1
    arr = C.c_void_p * L
    pa, na = arr(*[a.ctypes.data for a in pp]), arr(*[a.ctypes.data for a in pn])
    ints = C.c_int * L
    ws, hs = ints(*[a.shape[1] for a in pp]), ints(*[a.shape[0] for a in pp])
    pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 2)
    n = len(pts)
    out = np.zeros((n, 2), np.float32)
    st = np.zeros(n, np.uint8)
    err = np.zeros(n, np.float32)
    eps = min(max(float(epsilon), 0.0), 10.0) ** 2
    _lib().frc_lk_track(pa, na, ws, hs, ws, C.c_int(L), _p(pts), C.c_int(n), C.c_int(win[0]), C.c_int(win[1]),
                        C.c_int(min(max(int(max_count), 0), 100)), C.c_double(eps), _p(out), _p(st), _p(err))
    return out, st, err


def min_eig(img, block_size=3):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    e = np.zeros((h, w), np.float64)
    _lib().frc_min_eig(_p(img), C.c_int(w), C.c_int(h), C.c_int(w), C.c_int(block_size), _p(e))
    return e


def good_features(img, max_corners, quality, min_distance, block_size=3):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    cap = max_corners if max_corners > 0 else h * w
    out = np.zeros((cap, 2), np.float32)
    n = _lib().frc_good_features(_p(img), C.c_int(w), C.c_int(h), C.c_int(w), C.c_int(max_corners), C.c_double(quality),
                                 C.c_double(min_distance), C.c_int(block_size), _p(out), C.c_int(cap))
    return out[:n]


def clahe(plane, clip=3.5, tiles=(8, 8)):
    plane = np.ascontiguousarray(plane, np.uint8)
    h, w = plane.shape
    out = np.zeros_like(plane)
    _lib().frc_clahe(_p(plane), C.c_int(w), C.c_int(h), C.c_int(w), _p(out), C.c_int(w), C.c_int(tiles[0]), C.c_int(tiles[1]),
                     C.c_double(clip))
    return out


def increase_contrast(bgr, tables, clip=3.5, tiles=(8, 8)):
    bgr = np.ascontiguousarray(bgr, np.uint8)
    h, w, _ = bgr.shape
    g, cb, gi = (np.ascontiguousarray(t) for t in tables)
    out = np.zeros_like(bgr)
    _lib().frc_increase_contrast(_p(bgr), C.c_int(w), C.c_int(h), _p(g), _p(cb), _p(gi), C.c_double(clip),
                                 C.c_int(tiles[0]), C.c_int(tiles[1]), _p(out))
    return out


def bgr_to_grey(bgr):
    bgr = np.ascontiguousarray(bgr, np.uint8)
    out = np.zeros(bgr.shape[:2], np.uint8)
    _lib().frc_bgr_to_grey(_p(bgr), C.c_size_t(out.size), _p(out))
    return out


def lab_roundtrip(bgr, tables):
    """-> (lab [..,3] u8, bgr' [..,3] u8): the fixed-point conversion there and back, no CLAHE."""
    bgr = np.ascontiguousarray(bgr, np.uint8)
    g, cb, gi = (np.ascontiguousarray(t) for t in tables)
    lab, out = np.zeros_like(bgr), np.zeros_like(bgr)
    _lib().frc_lab_roundtrip(_p(bgr), C.c_size_t(bgr.size // 3), _p(g), _p(cb), _p(gi), _p(lab), _p(out))
    return lab, out
