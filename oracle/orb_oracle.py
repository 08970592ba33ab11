"""ctypes wrapper of the C oracle (oracle/orb_oracle.c).  TEST INFRASTRUCTURE ONLY — see the header of the
C file: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liborb_oracle.so")


def build(force=False):
    src = os.path.join(_HERE, "orb_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


def _lib():
    global _L
    try:
        return _L
    except NameError:
        pass
    if not os.path.exists(_SO):
        build()
    _L = C.CDLL(_SO)
    _L.orc_detect_compute.restype = C.c_int
    _L.orc_ratio_filter.restype = C.c_int
    return _L


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def level_sizes(H, W, nfeatures, nlevels=8, scale_factor=1.2):
    w = np.zeros(nlevels, np.int32)
    h = np.zeros(nlevels, np.int32)
    n = np.zeros(nlevels, np.int32)
    s = np.zeros(nlevels, np.float32)
    _lib().orc_level_sizes(C.c_int(H), C.c_int(W), C.c_int(nfeatures), C.c_int(nlevels), C.c_float(scale_factor),
                           _p(w), _p(h), _p(n), _p(s))
    return w, h, n, s


def umax():
    u = np.zeros(17, np.int32)
    _lib().orc_umax(_p(u))
    return u[:16]


def resize(src, wd, hd):
    src = np.ascontiguousarray(src, np.uint8)
    dst = np.zeros((hd, wd), np.uint8)
    _lib().orc_resize(_p(src), C.c_int(src.shape[1]), C.c_int(src.shape[0]), C.c_int(src.shape[1]), _p(dst),
                      C.c_int(wd), C.c_int(hd), C.c_int(wd))
    return dst


def detect_compute(img, nfeatures, pattern, nlevels=8, scale_factor=1.2, fast_threshold=20):
    """img [H,W] u8, pattern [256,4] int8 -> dict(xy[n,2] f32, meta[n,4] i32, resp[n] f32, mom[n,2] i32, desc[n,32])."""
    img = np.ascontiguousarray(img, np.uint8)
    pattern = np.ascontiguousarray(pattern, np.int8)
    H, W = img.shape
    xy = np.zeros((nfeatures, 2), np.float32)
    meta = np.zeros((nfeatures, 4), np.int32)
    resp = np.zeros(nfeatures, np.float32)
    mom = np.zeros((nfeatures, 2), np.int32)
    desc = np.zeros((nfeatures, 32), np.uint8)
    n = _lib().orc_detect_compute(_p(img), C.c_int(H), C.c_int(W), C.c_int(W), C.c_int(nfeatures), C.c_int(nlevels),
                                  C.c_float(scale_factor), C.c_int(fast_threshold), _p(pattern), _p(xy), _p(meta),
                                  _p(resp), _p(mom), _p(desc))
    return dict(xy=xy[:n], meta=meta[:n], resp=resp[:n], mom=mom[:n], desc=desc[:n], n=n)


def bf_knn2(q, t):
    q = np.ascontiguousarray(q, np.uint8)
    t = np.ascontiguousarray(t, np.uint8)
    idx = np.zeros((len(q), 2), np.int32)
    dist = np.zeros((len(q), 2), np.int32)
    _lib().orc_bf_knn2(_p(q), C.c_int(len(q)), _p(t), C.c_int(len(t)), _p(idx), _p(dist))
    return idx, dist


def ratio_filter(idx, dist, threshold=0.75):
    idx = np.ascontiguousarray(idx, np.int32)
    dist = np.ascontiguousarray(dist, np.int32)
    pairs = np.zeros((len(idx), 2), np.int32)
    m = _lib().orc_ratio_filter(_p(idx), _p(dist), C.c_int(len(idx)), C.c_double(threshold), _p(pairs))
    return pairs[:m]


# ---- single stages (tests/test_oracle_definitions.py) ---------------------------------------------------------------
def fast_score_map(img, t=20):
    img = np.ascontiguousarray(img, np.uint8)
    H, W = img.shape
    out = np.zeros((H, W), np.uint8)
    _lib().orc_fast_score_map(_p(img), C.c_int(H), C.c_int(W), C.c_int(W), C.c_int(t), _p(out))
    return out


def harris25_at(img, x, y):
    img = np.ascontiguousarray(img, np.uint8)
    f = _lib().orc_harris25_at
    f.restype = C.c_longlong
    return int(f(_p(img), C.c_int(img.shape[1]), C.c_int(int(x)), C.c_int(int(y))))


def blurred_at(img, x, y):
    img = np.ascontiguousarray(img, np.uint8)
    f = _lib().orc_blurred_at
    f.restype = C.c_int
    return int(f(_p(img), C.c_int(img.shape[1]), C.c_int(int(x)), C.c_int(int(y))))


def describe(img, x, y, pattern):
    img = np.ascontiguousarray(img, np.uint8)
    pattern = np.ascontiguousarray(pattern, np.int8)
    desc = np.zeros(32, np.uint8)
    m10, m01 = C.c_int(0), C.c_int(0)
    _lib().orc_describe(_p(img), C.c_int(img.shape[1]), C.c_int(int(x)), C.c_int(int(y)), _p(pattern), _p(desc),
                        C.byref(m10), C.byref(m01))
    return desc, m10.value, m01.value
