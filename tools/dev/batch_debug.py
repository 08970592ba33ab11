"""mm_ba_trf_batched against mm_ba_trf on problems shaped like sliding windows (wide band, multi-chunk segments)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from meatmodeler_amd import ops, synth, bundleAdjuster
from meatmodeler_amd._lib import default_context
ctx = default_context()
DEV = torch.device("cuda", 0)
specs = [(50, 20000, 40, 1), (50, 15000, 30, 2), (50, 3000, 6, 3), (40, 9000, 20, 4)]
probs, x0 = [], []
for F, P, L, seed in specs:
    pr = synth.make_ba_problem(F, P, L, seed=seed)
    with np.errstate(all="ignore"):
        cams0 = bundleAdjuster.frameParameters(pr["ext"]).reshape(F, 6)
    pb = ops.BADevice(pr["K"], pr["fi"], pr["pi"], pr["obs"], F, P, DEV, ctx)
    print("problem", F, P, L, "cam_span", pb.cam_span, "pairs", pb.n_pairs, "segments", pb.pb.n_seg, "chunks", pb.pb.n_chunks)
    probs.append(pb)
    x0.append((torch.as_tensor(cams0).to(DEV), torch.as_tensor(pr["pts0"].copy()).to(DEV)))
alone = []
for pb, (c0, p0) in zip(probs, x0):
    c, p_ = c0.clone(), p0.clone()
    rep, _ = pb.trf_solve(c, p_, 1e-6, 1e-8, 1e-8)
    alone.append((c, p_, rep))
    print("alone: nfev", rep.nfev, "status", rep.status, "cost", rep.cost)
for sub in ([0, 1, 2, 3], [0, 1], [2, 3], [0, 2]):
    cb, pbs_ = [x0[k][0].clone() for k in sub], [x0[k][1].clone() for k in sub]
    reps, sa = ops.trf_solve_batched([probs[k] for k in sub], cb, pbs_, 1e-6, 1e-8, 1e-8, ctx=ctx)
    print("batch", sub, "solved_alone", sa, "batch_last", ctx.control(ctx.CTL_BATCH_LAST))
    for k, c2, p2, r2 in zip(sub, cb, pbs_, reps):
        c, p_, r = alone[k]
        print("   problem", k, "nfev", r2.nfev, r.nfev, "cost", r2.cost, r.cost, "equal", bool(torch.equal(c, c2) and torch.equal(p_, p2)))
