"""Worker of tests/test_gpu_parity.py::test_library_dist_loop_over_rccl_one_rank: ONE rank, backend "nccl" (= RCCL), running
the sharded library loop (mm_ba_trf_dist) with an all-reduce that always goes through torch.distributed -- the collective is
trivial with one rank, but every call travels the real path: ProcessGroupNCCL on tensor views into the library's workspace,
ordered on the library's stream.  Started as a fresh process: `python tests/_rccl_one_rank_worker.py <out.npz>`."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_path = sys.argv[1]
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    import torch.distributed as dist
    dist.init_process_group("nccl", rank=0, world_size=1)
    from meatmodeler_amd import synth, parallel, ops
    from meatmodeler_amd._lib import default_context
    from meatmodeler_amd.bundleAdjuster import SchurTRF, frameParameters

    class ForcedAllReduce(parallel.AllReduce):
        calls = 0

        def __call__(self, tensor, op="sum"):      # (the base class returns early for one rank)
            ForcedAllReduce.calls += 1
            d = self.dist
            d.all_reduce(tensor, op=d.ReduceOp.SUM if op == "sum" else d.ReduceOp.MAX, group=self.group)
            return tensor

    ctx = default_context()
    pr = synth.make_ba_problem(40, 2000, 6, seed=1)
    F, P = len(pr["ext"]), len(pr["pts0"])
    with np.errstate(all="ignore"):
        cams0 = torch.as_tensor(frameParameters(pr["ext"]).reshape(F, 6)).to(dev)
    pts0 = torch.as_tensor(pr["pts0"].copy()).to(dev)
    res = {}
    for tag, ar in (("plain", None), ("rccl", ForcedAllReduce())):
        pb = ops.BADevice(pr["K"], pr["fi"], pr["pi"], pr["obs"], F, P, dev, ctx)
        r = SchurTRF(pb, allreduce=ar).solve(cams0.clone(), pts0.clone(), ftol=1e-8, xtol=1e-10, max_nfev=60)
        res[tag] = r
    # the clip path (fixed-shape device all-gathers of key points / matches, then the sharded adjustment) through the same
    # backend: one rank, trivial collectives, real ProcessGroupNCCL calls on device tensors
    from meatmodeler_amd.pipeline import ClipPipeline
    frames, ext, K = synth.render_orbit_frames(6, 640, 480, arc_deg=6.0)
    pipe = ClipPipeline(480, 640, 600, batch=6, device=dev, ctx=ctx)
    fr = torch.as_tensor(frames).to(dev)
    gathers = {"n": 0}
    orig_gather = parallel.gather_blocks

    def counting_gather(*a_, **k_):
        gathers["n"] += 1
        return orig_gather(*a_, **k_)
    parallel.gather_blocks = counting_gather
    o_plain = pipe.run(fr, K, ext, ba=True, ftol=1e-6, max_nfev=4)
    o_rccl = pipe.run(fr, K, ext, ba=True, ftol=1e-6, dist=dist, force_collectives=True, max_nfev=4)
    parallel.gather_blocks = orig_gather
    clip = dict(clip_gathers=gathers["n"],
                clip_tracks_equal=bool(torch.equal(o_plain["track_ptr_dev"], o_rccl["track_ptr_dev"])
                                       and torch.equal(o_plain["obs_frame_dev"], o_rccl["obs_frame_dev"])
                                       and torch.equal(o_plain["obs_kp_dev"], o_rccl["obs_kp_dev"])
                                       and torch.equal(o_plain["points0"], o_rccl["points0"])),
                clip_nfev=np.array([o_plain["ba"].nfev, o_rccl["ba"].nfev]),
                clip_cost=np.array([o_plain["ba"].cost, o_rccl["ba"].cost]),
                clip_collectives=getattr(o_rccl["ba"], "collectives", -1))
    a, b = res["plain"], res["rccl"]
    np.savez(out_path, **clip, nfev=np.array([a.nfev, b.nfev]), cost=np.array([a.cost, b.cost]), status=np.array([a.status, b.status]),
             cams_plain=a.cams.cpu().numpy(), cams_rccl=b.cams.cpu().numpy(), calls=ForcedAllReduce.calls,
             collectives=getattr(b, "collectives", -1), backend=dist.get_backend())
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
