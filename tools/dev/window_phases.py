"""Where a sliding window's time goes (selection / problem build / solve), C5 window shape on a short clip."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from meatmodeler_amd import synth, ops
from meatmodeler_amd.pipeline import ClipPipeline
from meatmodeler_amd.bundleAdjuster import SchurTRF, frameParameters
dev = torch.device("cuda", 0)
F, H, W, N = 200, 2160, 3840, 8000
frames, ext, K = synth.render_orbit_frames_torch(F, W, H, dev, arc_deg=0.72 * F)
pipe = ClipPipeline(H, W, N, batch=64, device=dev)
out = pipe.run(frames, K, ext, ba=False)
tp64 = out["track_ptr_dev"].long(); of_ = out["obs_frame_dev"]
first_f, last_f = of_[tp64[:-1]], of_[tp64[1:] - 1]; lens_all = tp64[1:] - tp64[:-1]
cams = torch.as_tensor(frameParameters(np.asarray(ext, float)[:, :3, :]).reshape(F, 6)).to(dev)
pts = out["points0"].clone()
for rep in range(2):
    for lo, hi in ((0, 50), (25, 75), (50, 100), (100, 150)):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        sel, fi, pi, coords, P, O = pipe._window_problem(out, first_f, last_f, lens_all, tp64, lo, hi, F)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        pb = ops.BADevice(K, fi, pi, coords, hi - lo, P, dev, pipe.ctx)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        res = SchurTRF(pb).solve(cams[lo:hi].contiguous(), pts[sel].contiguous(), ftol=1e-4)
        torch.cuda.synchronize(); t3 = time.perf_counter()
        print(f"window {lo}-{hi}: P {P} O {O} pairs {pb.n_pairs} | select {1e3*(t1-t0):.2f} ms, BADevice {1e3*(t2-t1):.2f} ms, solve {1e3*(t3-t2):.2f} ms "
              f"({res.nfev} evaluations, {1e3*(t3-t2)/res.nfev:.3f} ms each)")
