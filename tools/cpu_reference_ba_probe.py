#!/usr/bin/env python3
"""CPU probe: build a real-match BA problem with the ORACLE front end (C ORB + BF match + python linking + DLT) on a
rendered clip, run the reference's SciPy recipe on it, and save problem + result so the GPU solver can be compared on
the identical (outlier-laden) input.  Usage: python tools/cpu_reference_ba_probe.py F W H N out.npz"""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from meatmodeler_amd import synth
from meatmodeler_amd.orb_pattern import brief_pattern
from oracle import orb_oracle as oo, ba_oracle as bo

F, W, H, N, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
K = synth.default_K(W, H, f=525.0 * W / 640.0)
frames, ext_gt, _ = synth.render_orbit_frames(F, W, H, arc_deg=0.72 * F, seed=7, tex_size=2048, K=K)
rng = np.random.default_rng(5)
ext = ext_gt.copy()
for f in range(F):
    ext[f, :, :3] = synth.rodrigues(rng.normal(0, 5e-4, 3)) @ ext_gt[f, :, :3]
    ext[f, :, 3] += rng.normal(0, 2e-3, 3)
t0 = time.time()
det = [oo.detect_compute(frames[i], N, brief_pattern()) for i in range(F)]
tracks, popped = [], []
for k in range(1, F):
    idx, dist = oo.bf_knn2(det[k - 1]["desc"], det[k]["desc"])
    good = oo.ratio_filter(idx, dist, 0.75)
    pm = det[k - 1]["xy"][good[:, 0]].astype(np.float64)
    cm = det[k]["xy"][good[:, 1]].astype(np.float64)
    p, tracks = bo.point_tracking(tracks, k - 1, pm, k, cm)
    popped += p
final = popped + tracks
proj = np.einsum("ij,fjk->fik", K, ext)
f0 = np.array([t.getTriangulationData()[0] for t in final]); f1 = np.array([t.getTriangulationData()[1] for t in final])
x0 = np.array([t.getTriangulationData()[2] for t in final]); x1 = np.array([t.getTriangulationData()[3] for t in final])
X = bo.triangulate_dlt(proj[f0], proj[f1], x0, x1)
for t, x in zip(final, X):
    t.setPoint(x[None])
pts, coords, fi, pi = bo.manage_points(final)
pts = np.array(pts).reshape(-1, 3); coords = np.array(coords); fi = np.array(fi); pi = np.array(pi)
print(f"front end {time.time() - t0:.1f}s: {len(final)} tracks, {len(fi)} observations", flush=True)
t0 = time.time()
import io, contextlib
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    p2, e2, res = bo.adjust_points(ext, K, pts[:, None, :], coords, fi, pi, verbose=2, return_result=True)
print(f"scipy TRF: {time.time() - t0:.1f}s nfev={res.nfev} status={res.status} cost={res.cost:.6e}", flush=True)
print(buf.getvalue()[-1500:])
np.savez_compressed(out, ext=ext, K=K, pts0=pts, obs=coords, fi=fi, pi=pi, x_ref=res.x, cost_ref=res.cost,
                    nfev_ref=res.nfev, status_ref=res.status, table=buf.getvalue())
