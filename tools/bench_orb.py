#!/usr/bin/env python3
"""ORB detect + describe alone on the bench's frames (F x 1080p, 4000 key points): ms per clip, median of ORB_REPS."""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from meatmodeler_amd import synth
from meatmodeler_amd.pipeline import ClipPipeline
from meatmodeler_amd._lib import default_context, Timer

dev = torch.device("cuda", 0)
ctx = default_context()
F = int(os.environ.get("ORB_F", 500))
H, W, N = int(os.environ.get("ORB_H", 1080)), int(os.environ.get("ORB_W", 1920)), int(os.environ.get("ORB_N", 4000))
frames, ext, K = synth.render_orbit_frames_torch(F, W, H, dev, arc_deg=0.72 * F)
pipe = ClipPipeline(H, W, N, batch=min(F, 512), device=dev, ctx=ctx)
t = Timer(ctx)
ms = []
ref = None
for rep in range(int(os.environ.get("ORB_REPS", 4))):
    t.start()
    det = pipe.detect(frames)
    t.stop()
    ms.append(t.elapsed_ms())
    sig = (int(det["n"].sum()), int(det["desc"].long().sum()))
    assert ref is None or sig == ref
    ref = sig
print(f"detect {F} x {W}x{H}, {N} kpts: " + " ".join(f"{m:.2f}" for m in ms) + f" ms; key points {ref[0]}, descriptor checksum {ref[1]}")
