"""cProfile of the host side of SchurTRF.solve on a C3-shaped synthetic problem (where does the host spend the GPU's idle time)."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench_schur import c3_like_problem  # noqa: E402
from meatmodeler_amd import ops  # noqa: E402
from meatmodeler_amd._lib import default_context  # noqa: E402
from meatmodeler_amd.bundleAdjuster import SchurTRF, frameParameters  # noqa: E402

ctx = default_context()
dev = ctx.device
K, ext, pts0, fi, pi, obs = c3_like_problem(P=200000)
F, P = len(ext), len(pts0)
pb = ops.BADevice(K, fi, pi, obs, F, P, dev, ctx)
cams = torch.as_tensor(frameParameters(ext).reshape(F, 6)).to(dev)
pts = torch.as_tensor(pts0).to(dev)
solver = SchurTRF(pb)
res = solver.solve(cams, pts, ftol=1e-15, xtol=1e-15, gtol=1e-15, max_nfev=15)
ctx.sync()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
res = solver.solve(cams, pts, ftol=1e-15, xtol=1e-15, gtol=1e-15, max_nfev=40)
ctx.sync()
pr.disable()
dt = time.perf_counter() - t0
print(f"nfev {res.nfev} iterations {res.iterations}: {dt * 1e3:.1f} ms, {dt * 1e3 / max(res.iterations, 1):.3f} ms per iteration, "
      f"host segments {res.host_segments_ms}")
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
