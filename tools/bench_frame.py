"""Kernel times of the per-frame front end (rows (f)-3 / (f)-4) at 1080p: contrast + grey on a batch of BGR frames, the
LK pyramid, Lucas-Kanade tracking of `npts` points, the Shi-Tomasi map + candidates.  Algorithmic HBM bytes per launch
are printed beside the achieved rate.  usage: python tools/bench_frame.py [batch] [npts] [reps]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from meatmodeler_amd import ops, synth  # noqa: E402
from meatmodeler_amd._lib import default_context  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
npts = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
W, H = 1920, 1080
ctx = default_context()
dev = ctx.device
K = synth.default_K(W, H, f=525.0 * W / 640.0)
frames, _, _ = synth.render_orbit_frames_torch(B, W, H, dev, arc_deg=0.72 * B, seed=7, K=K)
bgr = torch.stack([frames, frames.roll(5, 2), 255 - frames], -1).contiguous()
pts = ops.good_features(frames[0].contiguous(), npts, 0.01, 7, 7, ctx)
pts_d = torch.as_tensor(pts).to(dev)
print(f"{B} frames {W}x{H}, {len(pts)} points")


def run():
    out, grey = ops.increase_contrast(bgr, want_grey=True, ctx=ctx)
    p0 = ops.pyramid(grey[0], 3, ctx)
    p1 = ops.pyramid(grey[1], 3, ctx)
    ops.lk_track(p0, p1, pts_d, (21, 21), 30, 0.01, ctx)
    ops.min_eig(grey[0], 7, ctx)


for _ in range(3):
    run()
ctx.sync()
ctx.profile(1)
for _ in range(reps):
    run()
ctx.sync()
rep = ctx.profile_report()
ctx.profile(0)
px = W * H
alg = {"lab_forward_kernel": B * px * 6, "clahe_lut_kernel": B * px, "clahe_apply_kernel": B * px * (3 + 3 + 1),
       "pyr_down_kernel": None, "min_eig_kernel": px * (1 + 8), "lk_track_kernel": None}
for name, (n, ms) in sorted(rep.items(), key=lambda kv: -kv[1][1]):
    us = ms / n * 1e3
    extra = ""
    if name == "pyr_down_kernel":
        extra = "  (3 levels per pyramid; level 0->1: %.0f GB/s algorithmic at the mean launch time x3/1.3125)" % (
            px * 1.25 / (us * 1e-6) / 1e9 * 1.3125 / 3)
    elif alg.get(name):
        extra = "  %.0f GB/s algorithmic (%.1f MB per launch)" % (alg[name] / (us * 1e-6) / 1e9, alg[name] / 1e6)
    elif name == "lk_track_kernel":
        extra = "  %.2f us per point (4 levels, <= 30 iterations, 21x21 window)" % (us / max(len(pts), 1))
    print(f"{name:28s} {n:5d} launches  {us:9.1f} us each{extra}")
