import sys, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from meatmodeler_amd import synth, ops
from meatmodeler_amd._lib import default_context
from meatmodeler_amd.pipeline import ClipPipeline
ctx = default_context(); dev = ctx.device
F,H,W,N = 500,1080,1920,4000
K = synth.default_K(W,H,f=525.0*W/640.0)
frames, ext_gt, _ = synth.render_orbit_frames_torch(F,W,H,dev,arc_deg=360.0,seed=7,K=K)
pipe = ClipPipeline(H,W,N,batch=32,device=dev,ctx=ctx)
o = pipe.run(frames,K,ext_gt,ba=False)
tp, of_, ok = o["track_ptr_dev"], o["obs_frame_dev"], o["obs_kp_dev"]
P = tp.shape[0]-1; O = of_.shape[0]
lens = (tp[1:]-tp[:-1]).long()
pi = torch.repeat_interleave(torch.arange(P,dtype=torch.int32,device=dev), lens, output_size=O)
coords = o["xy_dev"][of_.long(), ok.long()].double()
pb = ops.BADevice(K, of_, pi, coords, F, P, dev, ctx)
print("P",P,"O",O,"pairs",pb.n_pairs,"span",pb.cam_span,"seg",pb.pb.n_seg,"chunks",pb.pb.n_chunks)
cnt = (pb.seg_chunk_ptr[1:]-pb.seg_chunk_ptr[:-1])
# pairs per segment
begin = pb.chunk_begin[pb.seg_chunk_ptr[:-1].long()].long(); 
end = torch.cat([begin[1:], torch.tensor([pb.n_pairs],device=dev)])
sz = (end-begin).cpu().numpy()
d = (pb.seg_ids % (pb.cam_span+1)).cpu().numpy()
for lo,hi in [(1,2),(2,4),(4,8),(8,16),(16,32),(32,64),(64,128),(128,256),(256,1024),(1024,10**9)]:
    m=(sz>=lo)&(sz<hi); print(f"pairs/seg [{lo},{hi}): segs {m.sum()} pairs {sz[m].sum()}")
print("by d: ", [(int(k), int(sz[d==k].sum())) for k in (0,1,2,3,4,5,8,12,20,40,87)])
tl = lens.cpu().numpy(); print("track len hist", np.bincount(np.minimum(tl,20)))
