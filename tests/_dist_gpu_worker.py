"""Worker of tests/test_gpu_parity.py::test_two_ranks_on_one_gpu_* : one rank of a world-size-2 `gloo` group, both
ranks on cuda:0, running the PRODUCT's sharded paths with the real HIP kernels (ops.BADevice, band all-reduce, device
gathers).  Started as a fresh process per rank: `python tests/_dist_gpu_worker.py <outdir>` with RANK / WORLD_SIZE /
MASTER_ADDR / MASTER_PORT in the environment.  World size 1 runs the same code without a process group (the reference
result the test compares against)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def long_track_problem(synth):
    """Well-conditioned BA problem (230 cameras, short tracks) whose LAST point is also seen in cameras 2 and 215: the shard
    that holds it has camera span > 192 -> no pair list (general Schur kernel), the other shard has one."""
    pr = synth.make_ba_problem(230, 1500, 6, seed=11)
    K, ext, X = pr["K"], pr["ext_gt"], pr["pts_gt"][-1]
    own = pr["fi"][pr["pi"] == 1499]
    far = np.array([f for f in (2, 215) if f not in own])
    rng = np.random.default_rng(3)
    u = np.einsum("ij,fj->fi", K, np.einsum("fij,j->fi", ext[far, :, :3], X) + ext[far, :, 3])
    extra = u[:, :2] / u[:, 2:3] + rng.normal(0, 0.5, (len(far), 2))
    # keep frames ascending inside the point (managePoints order)
    fi_p = np.append(own, far)
    ob_p = np.vstack([pr["obs"][pr["pi"] == 1499], extra])
    order = np.argsort(fi_p, kind="stable")
    keep = pr["pi"] != 1499
    pr["fi"] = np.concatenate([pr["fi"][keep], fi_p[order]])
    pr["pi"] = np.concatenate([pr["pi"][keep], np.full(len(fi_p), 1499)])
    pr["obs"] = np.vstack([pr["obs"][keep], ob_p[order]])
    return pr


def main():
    outdir = sys.argv[1]
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from meatmodeler_amd import synth, parallel, ops
    from meatmodeler_amd._lib import default_context
    from meatmodeler_amd.bundleAdjuster import SchurTRF, frameParameters
    from meatmodeler_amd.pipeline import ClipPipeline
    ctx = default_context()
    out = {}

    def sharded_ba(tag, pr):
        F, P = len(pr["ext"]), len(pr["pts0"])
        with np.errstate(all="ignore"):
            cams0 = torch.as_tensor(frameParameters(pr["ext"]).reshape(F, 6)).to(dev)
        lo, hi, mask = parallel.partition_points(pr["fi"], pr["pi"], P, rank, world)
        pb = ops.BADevice(pr["K"], pr["fi"][mask], pr["pi"][mask] - lo, pr["obs"][mask], F, hi - lo, dev, ctx)
        solver = SchurTRF(pb, allreduce=parallel.AllReduce() if world > 1 else None)
        res = solver.solve(cams0, torch.as_tensor(pr["pts0"][lo:hi].copy()).to(dev), ftol=1e-8, xtol=1e-10, max_nfev=60)
        out.update({f"{tag}_cams": res.cams.cpu().numpy(), f"{tag}_pts": res.pts.cpu().numpy(), f"{tag}_lo": lo,
                    f"{tag}_hi": hi, f"{tag}_cost": res.cost, f"{tag}_nfev": res.nfev, f"{tag}_status": res.status,
                    f"{tag}_n_pairs": pb.n_pairs, f"{tag}_cam_span": pb.cam_span,
                    f"{tag}_collectives": getattr(res, "collectives", -1), f"{tag}_iterations": getattr(res, "iterations", -1),
                    f"{tag}_fallbacks": getattr(res, "chol_fallbacks", -1) or 0})

    # (1) banded exchange: every shard has a pair list (golden G5 case c)
    sharded_ba("band", synth.make_ba_problem(40, 2000, 6, seed=1))
    # (2) one shard has a track spanning > 192 cameras: all ranks must take the SAME (dense) exchange
    sharded_ba("long", long_track_problem(synth))
    # (3) whole clip through ClipPipeline.run(dist=...) and the sharded sliding-window adjustment
    frames, ext, K = synth.render_orbit_frames(9, 640, 480, arc_deg=9.0)
    pipe = ClipPipeline(480, 640, 800, batch=5, device=dev, ctx=ctx)
    o = pipe.run(torch.as_tensor(frames).to(dev), K, ext, ba=True, ftol=1e-4, dist=dist)
    out.update(clip_track_ptr=o["track_ptr_dev"].cpu().numpy(), clip_obs_frame=o["obs_frame_dev"].cpu().numpy(),
               clip_obs_kp=o["obs_kp_dev"].cpu().numpy(), clip_match_count=o["match_count"], clip_kp_count=o["kp_count"],
               clip_cost=o["ba"].cost, clip_nfev=o["ba"].nfev, clip_cams=o["ba"].cams.cpu().numpy(),
               clip_n_pairs=o["n_pairs"], clip_points0=o["points0"].cpu().numpy())
    # (3b) ONE trust-region iteration of the same outlier-laden real-match problem (max_nfev = 2: initial cost, one trial
    # step): the full path above is chaotic in the iteration count, a single step is not -- 1 rank vs 2 ranks differ by the
    # order of the sums only.  This is the regression detector of the SHARDED arithmetic on real matches.
    o1 = pipe.run(torch.as_tensor(frames).to(dev), K, ext, ba=True, ftol=1e-4, dist=dist, max_nfev=2)
    out.update(step_cams=o1["ba"].cams.cpu().numpy(), step_cost=o1["ba"].cost, step_nfev=o1["ba"].nfev,
               step_status=o1["ba"].status, step_cost0=float(getattr(o1["ba"], "cost0", np.nan)))
    lo_, hi_, _, _ = (parallel.partition_tracks(o1["track_ptr_dev"], rank, world) if world > 1
                      else (0, o1["n_tracks"], 0, 0))
    pts_full = torch.zeros((o1["n_tracks"], 3), dtype=torch.float64, device=dev)
    pts_full[lo_:hi_] = o1["ba"].pts
    if world > 1:
        parallel.AllReduce()(pts_full)
    out.update(step_pts=pts_full.cpu().numpy())
    # (3c) ADVICE round 3: the choice between the two factorisation paths must be GLOBAL.  Rank 1 alone is told to avoid
    # the single launch / rank 0 alone has its next factorisation abandoned: every rank must end on the same path with
    # bit-identical replicated cameras.
    pr = synth.make_ba_problem(120, 3000, 6, seed=4)           # 120 cameras: the banded single-launch path
    for tag, knob, who in (("avoid", ctx.CTL_CHOL_AVOID_FUSED, 1), ("abandon", ctx.CTL_CHOL_FORCE_ABANDON, 0)):
        if world > 1 and rank == who:
            ctx.control(knob, 1)
        try:
            sharded_ba(tag, pr)
        finally:
            ctx.control(ctx.CTL_CHOL_AVOID_FUSED, 0)
            ctx.control(ctx.CTL_CHOL_FORCE_ABANDON, 0)
        out[f"{tag}_last_path"] = ctx.control(ctx.CTL_CHOL_LAST_PATH)
    w = pipe.adjust_windows(o, K, ext, window=5, stride=2, ftol=1e-4, dist=dist)
    out.update(win_cams=w["cams"].cpu().numpy(), win_pts=w["points"].cpu().numpy(),
               win_nfev=np.array([s["nfev"] for s in w["windows"]]), win_cost=np.array([s["cost"] for s in w["windows"]]),
               win_points=np.array([s["points"] for s in w["windows"]]))
    wf = pipe.adjust_windows(o, K, ext, window=5, stride=2, ftol=1e-4, dist=dist, order="wavefront")
    out.update(wf_cams=wf["cams"].cpu().numpy(), wf_pts=wf["points"].cpu().numpy(),
               wf_table=np.array([[s["lo"], s["hi"], s["points"], s["observations"], s["nfev"], s["status"], s["cost"], s["colour"]]
                                  for s in wf["windows"]]))
    np.savez(os.path.join(outdir, f"w{world}_rank{rank}.npz"), **out)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
