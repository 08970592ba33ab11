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
