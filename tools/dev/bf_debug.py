import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from meatmodeler_amd import ops
dev = torch.device("cuda", 0)
g = torch.Generator(device="cpu").manual_seed(0)
for n in (64, 96, 2000):
    q = torch.randint(0, 256, (1, n, 32), dtype=torch.uint8, generator=g).to(dev)
    t = torch.randint(0, 256, (1, n, 32), dtype=torch.uint8, generator=g).to(dev)
    res = {}
    for v in (114, 200, 300):
        os.environ["MM_BF_VARIANT"] = str(v)
        idx, dist = ops.bf_knn2_batched(q, t)
        res[v] = (idx.cpu().numpy()[0], dist.cpu().numpy()[0])
    for v in (200, 300):
        bad = np.nonzero((res[v][0] != res[114][0]).any(1) | (res[v][1] != res[114][1]).any(1))[0]
        print(f"n={n} variant {v}: {len(bad)} of {n} queries differ", bad[:8])
        for b in bad[:3]:
            print("   query", b, "got idx", res[v][0][b], "dist", res[v][1][b], "want idx", res[114][0][b], "dist", res[114][1][b])
