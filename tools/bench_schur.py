"""Time the reduced-camera-system build (schur_pairs_kernel) alone and overlapped with the factorisation, on a synthetic
problem shaped like the 500-frame clip: 500 cameras, ~405 k points, ~1.63 M observations, ~7 M co-observation pairs,
camera span 87.  usage: python tools/bench_schur.py [reps]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from meatmodeler_amd import ops, synth  # noqa: E402
from meatmodeler_amd._lib import default_context  # noqa: E402


def c3_like_problem(F=500, P=405000, seed=0, span=87):
    rng = np.random.default_rng(seed)
    K = synth.default_K(1920, 1080)
    ext = synth.orbit_cameras(F, arc_deg=360.0)
    L = np.minimum(2 + rng.geometric(0.42, P) - 1, 40)
    L[: P // 2000] = rng.integers(40, span + 2, P // 2000)       # a few long tracks set the band
    L[0] = span + 1
    start = (rng.random(P) * (F - L + 1)).astype(np.int64)
    fi = np.concatenate([s + np.arange(l) for s, l in zip(start, L)]).astype(np.int32)
    pi = np.repeat(np.arange(P, dtype=np.int32), L)
    pts = rng.uniform(-2, 2, (P, 3))
    Xc = np.einsum("oij,oj->oi", ext[fi, :, :3], pts[pi]) + ext[fi, :, 3]
    u = Xc @ K.T
    obs = u[:, :2] / u[:, 2:3] + rng.normal(0, 0.5, (len(fi), 2))
    return K, ext, pts + rng.normal(0, 0.02, pts.shape), fi, pi, obs


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    from meatmodeler_amd.bundleAdjuster import frameParameters
    ctx = default_context()
    dev = ctx.device
    K, ext, pts0, fi, pi, obs = c3_like_problem()
    F, P = len(ext), len(pts0)
    pb = ops.BADevice(K, fi, pi, obs, F, P, dev, ctx)
    cams = torch.as_tensor(frameParameters(ext).reshape(F, 6)).to(dev)
    pts = torch.as_tensor(pts0).to(dev)
    B, gc, C, gp = pb.normal_eq(cams, pts)
    Bd = B + 1e-3 * torch.diag_embed(torch.diagonal(B, dim1=1, dim2=2))
    Cd = C.clone()
    Cd[:, [0, 3, 5]] *= 1.001
    print(f"F {F} P {P} O {pb.O} pairs {pb.n_pairs} cam_span {pb.cam_span} segments {pb.pb.n_seg} chunks {pb.pb.n_chunks}")
    half_bw = 6 * pb.cam_span + 5
    for mode, label in ((0, "schur alone"), (1, "schur_solve overlapped"), (2, "schur then chol (serial)")):
        for rep in range(3):      # warm up
            if mode == 0:
                pb.schur(cams, pts, Bd, Cd, gc, gp)
            else:
                pb.overlap = mode == 1
                info, v, _ = pb.schur_solve(cams, pts, Bd, Cd, gc, gp, half_bw)
                assert int(info) == 0, int(info)
        ctx.sync()
        ctx.profile(1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for rep in range(reps):
            if mode == 0:
                pb.schur(cams, pts, Bd, Cd, gc, gp)
            else:
                pb.schur_solve(cams, pts, Bd, Cd, gc, gp, half_bw)
        e1.record()
        ctx.sync()
        rep_ = ctx.profile_report()
        ctx.profile(0)
        line = ", ".join(f"{k} {ms / n * 1e3:.0f} us" for k, (n, ms) in sorted(rep_.items(), key=lambda kv: -kv[1][1])[:5])
        print(f"{label}: {e0.elapsed_time(e1) / reps * 1e3:.0f} us per call  [{line}]")


if __name__ == "__main__":
    main()
