"""Time one adjustPoints solve with the Python-sequenced loop and with the loop inside the library (mm_ba_trf)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from meatmodeler_amd import ops, bundleAdjuster
from meatmodeler_amd._lib import default_context
from meatmodeler_amd import synth
from meatmodeler_amd.bundleAdjuster import frameParameters

def run(F, P, L, drv, reps=3):
    os.environ["MM_TRF_DRIVER"] = drv
    pr = synth.make_ba_problem(F, P, L, seed=1)
    ctx = default_context(); dev = ctx.device
    pb = ops.BADevice(pr["K"], pr["fi"], pr["pi"], pr["obs"], F, P, dev, ctx)
    cams0 = torch.as_tensor(frameParameters(pr["ext"]).reshape(F, 6), device=dev); pts0 = torch.as_tensor(pr["pts0"], device=dev)
    best = None
    for _ in range(reps):
        s = bundleAdjuster.SchurTRF(pb)
        torch.cuda.synchronize(); t = time.perf_counter()
        r = s.solve(cams0, pts0, ftol=1e-4)
        torch.cuda.synchronize(); dt = time.perf_counter() - t
        best = dt if best is None else min(best, dt)
    print(f"F={F} P={P} L={L} {drv:8s}: {best*1e3:8.2f} ms, nfev {r.nfev}, {best*1e3/r.nfev:.3f} ms/eval, cost {r.cost:.6e}")

if __name__ == "__main__":
    for F, P, L in ((500, 200000, 5), (50, 1000, 4), (10, 300, 4)):
        for drv in ("python", "library"):
            run(F, P, L, drv)
