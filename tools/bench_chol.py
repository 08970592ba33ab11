"""Time mm_chol_solve on the reduced-camera-system shape of the 500-frame clip (n = 3000, half bandwidth 528).
usage: MM_CHOL_FUSED={0,1,2} MM_CHOL_TWISTED={0,1} python tools/bench_chol.py [n] [hb] [reps] [sym]
sym = 1 (default): mm_chol_solve_sym (solution only; narrow bands are eliminated from both ends); 0: mm_chol_solve."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from meatmodeler_amd import ops  # noqa: E402
from meatmodeler_amd._lib import default_context  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
hb = int(sys.argv[2]) if len(sys.argv) > 2 else 528
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
sym = int(sys.argv[4]) if len(sys.argv) > 4 else 1
solve = (lambda A_, b_: ops.chol_solve_sym(A_, b_, ctx, half_bandwidth=hb, both_triangles=True)) if sym else \
    (lambda A_, b_: ops.chol_solve(A_, b_, ctx, half_bandwidth=hb))
rng = np.random.default_rng(0)
M = np.tril(np.triu(rng.normal(size=(n, n)), -hb // 2))
A = M @ M.T + n * np.eye(n)
b = rng.normal(size=n)
ctx = default_context()
dev = torch.device("cuda:0")
A0 = torch.as_tensor(A).to(dev)
b0 = torch.as_tensor(b).to(dev)
ref = np.linalg.solve(A, b)
worst = 0.0
for it in range(3):
    Ad, bd = A0.clone(), b0.clone()
    info = solve(Ad, bd)
    assert int(info) == 0, int(info)
    worst = max(worst, float(np.abs(bd.cpu().numpy() - ref).max() / np.abs(ref).max()))
ctx.sync()
ts = []
for it in range(reps):
    Ad, bd = A0.clone(), b0.clone()
    ctx.sync()
    t0 = time.perf_counter()
    info = solve(Ad, bd)
    ctx.sync()
    ts.append((time.perf_counter() - t0) * 1e3)
    worst = max(worst, float(np.abs(bd.cpu().numpy() - ref).max() / np.abs(ref).max()))
print(f"MM_CHOL_FUSED={os.environ.get('MM_CHOL_FUSED', 'default')} MM_CHOL_TWISTED={os.environ.get('MM_CHOL_TWISTED', 'default')} sym={sym} n={n} hb={hb}: median {np.median(ts):.3f} ms, "
      f"min {min(ts):.3f} ms per factor+solve, worst rel err {worst:.2e}")
