"""Where the ~2 ms between 'ba' and 'ba_solve' of a bench step go: BADevice (CSR, pair list) section by section."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench_schur import c3_like_problem
from meatmodeler_amd import ops
from meatmodeler_amd._lib import default_context

ctx = default_context(); dev = ctx.device
K, ext, pts0, fi, pi, obs = c3_like_problem(P=405000)
F, P = len(ext), len(pts0)
fi_d = torch.as_tensor(fi).to(dev); pi_d = torch.as_tensor(pi).to(dev); obs_d = torch.as_tensor(obs).to(dev)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pb = ops.BADevice(K, fi_d, pi_d, obs_d, F, P, dev, ctx)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"BADevice: {(t1 - t0) * 1e3:.2f} ms  (O {pb.O}, pairs {pb.n_pairs}, chunks {pb.pb.n_chunks})")
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    pb = ops.BADevice(K, fi_d, pi_d, obs_d, F, P, dev, ctx)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=18, max_name_column_width=60))
