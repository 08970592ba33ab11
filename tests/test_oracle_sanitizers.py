"""The C oracles under AddressSanitizer + UndefinedBehaviorSanitizer (CPU build only; SURVEY.md section 5: "compile CPU
restatement with -fsanitize=address,undefined in tests").  The instrumented libraries are built into a temp directory
and exercised in a child process (ASan has to be preloaded before Python loads them through ctypes)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import ctypes as C, sys, numpy as np
sys.path.insert(0, {root!r})
import oracle.orb_oracle as oo, oracle.frame_oracle as fo
oo._L = C.CDLL({orb!r}); oo._L.orc_detect_compute.restype = C.c_int; oo._L.orc_ratio_filter.restype = C.c_int
fo._L = C.CDLL({frame!r}); fo._L.frc_good_features.restype = C.c_int
from meatmodeler_amd import synth, frame_tables
from meatmodeler_amd.orb_pattern import brief_pattern
frames, _, _ = synth.render_orbit_frames(2, 200, 160, arc_deg=2.0, seed=1)
d0 = oo.detect_compute(frames[0], 150, brief_pattern()); d1 = oo.detect_compute(frames[1], 150, brief_pattern())
idx, dist = oo.bf_knn2(d0["desc"], d1["desc"]); oo.ratio_filter(idx, dist, 0.75)
oo.bf_knn2(d0["desc"][:0], d1["desc"]); oo.bf_knn2(d0["desc"], d1["desc"][:1])
c = fo.good_features(frames[0], 40, 0.05, 6, 5)
pts = np.vstack([c, [[0.0, 0.0], [199.0, 159.0], [-50.0, 3.0], [400.0, 400.0]]]).astype(np.float32)
fo.lk_track(frames[0], frames[1], pts, (21, 21), 3, 30, 0.01)
fo.lk_track(frames[0], frames[1], pts, (5, 9), 0, 3, 0.5)
fo.good_features(frames[0], 0, 0.5, 0.0, 3); fo.good_features(np.zeros((9, 7), np.uint8), 5, 0.1, 2, 7)
bgr = np.stack([frames[0], frames[1], 255 - frames[0]], -1)[:157, :197].copy()
fo.increase_contrast(bgr, frame_tables.lab_tables()); fo.bgr_to_grey(bgr); fo.clahe(frames[0][:33, :41].copy())
fo.lab_roundtrip(np.random.default_rng(0).integers(0, 256, (500, 1, 3), dtype=np.uint8), frame_tables.lab_tables())
print("sanitized run ok", d0["n"], len(c))
"""


def test_c_oracles_clean_under_asan_ubsan(tmp_path):
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not libasan or not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("libasan not available")
    libs = {}
    for name in ("orb_oracle", "frame_oracle"):
        so = str(tmp_path / f"lib{name}_san.so")
        subprocess.check_call(["gcc", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined",
                               "-fno-sanitize-recover=undefined", "-ffp-contract=off", "-shared", "-fPIC", "-o", so,
                               os.path.join(ROOT, "oracle", name + ".c"), "-lm"])
        libs[name] = so
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-c", CHILD.format(root=ROOT, orb=libs["orb_oracle"], frame=libs["frame_oracle"])],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "sanitized run ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


HOST_CHILD = r"""
import ctypes as C, sys, os, numpy as np
L = C.CDLL({so!r})
vp = C.c_void_p
p = lambda a: a.ctypes.data_as(vp)
# goodFeaturesToTrack's greedy selection
rng = np.random.default_rng(0)
w, h = 97, 61
pos = rng.permutation(w * h)[:900].astype(np.int32)
out = np.zeros((64, 2), np.float32)
L.mm_gftt_select.argtypes = [vp, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_double, vp, C.c_int]
for md in (0.0, 1.0, 5.5, 200.0):
    m = L.mm_gftt_select(p(pos), len(pos), w, h, 50, md, p(out), 64)
    assert 0 < m <= 50
    if md >= 1:
        q = out[:m]
        d = np.linalg.norm(q[:, None] - q[None], axis=2) + 1e9 * np.eye(m)
        assert d.min() >= md - 1e-6
assert L.mm_gftt_select(p(pos), 0, w, h, 5, 3.0, p(out), 64) == 0
# PLY writer
L.mm_write_ply.argtypes = [C.c_char_p, vp, C.c_int64]
xyz = rng.normal(size=(11, 3))
assert L.mm_write_ply({ply!r}.encode(), p(xyz), 11) == 0 and os.path.getsize({ply!r}) > 11 * 24
# BA index build (CSR by point / by camera) incl. the range check
L.mm_ba_build_index.argtypes = [C.c_int, C.c_int, C.c_int64] + [vp] * 6
fi = rng.integers(0, 7, 200).astype(np.int32); pi = np.sort(rng.integers(0, 30, 200)).astype(np.int32)
pt_ptr = np.zeros(31, np.int32); pt_obs = np.zeros(200, np.int32); cam_ptr = np.zeros(8, np.int32); cam_obs = np.zeros(200, np.int32)
assert L.mm_ba_build_index(7, 30, 200, p(fi), p(pi), p(pt_ptr), p(pt_obs), p(cam_ptr), p(cam_obs)) == 0
assert pt_ptr[-1] == 200 and cam_ptr[-1] == 200 and sorted(cam_obs.tolist()) == list(range(200))
fi[3] = 9
assert L.mm_ba_build_index(7, 30, 200, p(fi), p(pi), p(pt_ptr), p(pt_obs), p(cam_ptr), p(cam_obs)) != 0
# track linking over a clip (hash join) incl. a malformed match index
L.mm_link_tracks_clip.restype = C.c_int64
L.mm_link_tracks_clip.argtypes = [C.c_int, C.c_int, vp, vp, C.c_int, vp, vp, C.c_int64, C.c_int64, vp, vp, vp, vp]
F, cap = 5, 40
kc = np.full(F, cap, np.int32); xy = rng.uniform(0, 100, (F, cap, 2)).astype(np.float32)
mc = np.full(F - 1, 25, np.int32); mm = np.stack([np.stack([rng.permutation(cap)[:25], rng.integers(0, cap, 25)], 1) for _ in range(F - 1)]).astype(np.int32)
tot = int(mc.sum())
tp = np.zeros(tot + 2, np.int64); of = np.zeros(2 * tot + 2, np.int32); ok = np.zeros(2 * tot + 2, np.int32); no = C.c_int64(0)
nt = L.mm_link_tracks_clip(F, cap, p(kc), p(xy), 25, p(mc), p(mm), tot + 1, 2 * tot + 1, p(tp), p(of), p(ok), C.byref(no))
assert nt > 0 and no.value >= 2 * nt and tp[nt] == no.value
print("sanitized host run ok", nt, no.value)
"""


def test_host_side_of_the_library_clean_under_asan_ubsan(tmp_path):
    """meatmodeler_amd/csrc/host_index.cpp (track linking, BA index build, goodFeaturesToTrack's greedy selection, PLY
    writer) is plain C++: compiled alone with g++ -fsanitize=address,undefined and exercised through ctypes."""
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not libasan or not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("libasan not available")
    so = str(tmp_path / "libhost_san.so")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fno-omit-frame-pointer", "-fsanitize=address,undefined",
                           "-fno-sanitize-recover=undefined", "-shared", "-fPIC", "-o", so,
                           os.path.join(ROOT, "meatmodeler_amd", "csrc", "host_index.cpp"), "-lpthread"])
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1")
    r = subprocess.run([sys.executable, "-c", HOST_CHILD.format(so=so, ply=str(tmp_path / "c.ply"))], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "sanitized host run ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
