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
    a, b = res["plain"], res["rccl"]
    np.savez(out_path, nfev=np.array([a.nfev, b.nfev]), cost=np.array([a.cost, b.cost]), status=np.array([a.status, b.status]),
             cams_plain=a.cams.cpu().numpy(), cams_rccl=b.cams.cpu().numpy(), calls=ForcedAllReduce.calls,
             collectives=getattr(b, "collectives", -1), backend=dist.get_backend())
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
