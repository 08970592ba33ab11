import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from meatmodeler_amd import ops
dev = torch.device("cuda", 0)
os.environ["MM_BF_VARIANT"] = "300"
n = 64
g = torch.Generator(device="cpu").manual_seed(0)
t = torch.randint(0, 256, (1, n, 32), dtype=torch.uint8, generator=g)
# (a) queries == trains
idx, dist = ops.bf_knn2_batched(t.to(dev), t.to(dev))
print("q == t: idx0", idx[0, :12, 0].tolist(), "dist0", dist[0, :12, 0].tolist())
# (b) all trains zero except train 10 == all ones; query all ones
tz = torch.zeros((1, n, 32), dtype=torch.uint8); tz[0, 10] = 255
q1 = torch.full((1, n, 32), 255, dtype=torch.uint8)
idx, dist = ops.bf_knn2_batched(q1.to(dev), tz.to(dev))
print("ones vs zeros (train 10 ones): idx", idx[0, 0].tolist(), "dist", dist[0, 0].tolist())
# (c) train k has its first k bits set (k < 64), query zero -> dist = k
tk = torch.zeros((1, n, 32), dtype=torch.uint8)
for k in range(n):
    bits = np.zeros(256, np.uint8); bits[:k] = 1
    tk[0, k] = torch.from_numpy(np.packbits(bits, bitorder="little"))
q0 = torch.zeros((1, n, 32), dtype=torch.uint8)
idx, dist = ops.bf_knn2_batched(q0.to(dev), tk.to(dev))
print("query 0 vs train k with k bits: idx", idx[0, 0].tolist(), "dist", dist[0, 0].tolist())
# (d) query has first 100 bits set; expect nearest train = 63 (dist 37), 62 (38)
qb = torch.zeros((1, n, 32), dtype=torch.uint8); bits = np.zeros(256, np.uint8); bits[:100] = 1
qb[0, :] = torch.from_numpy(np.packbits(bits, bitorder="little"))
idx, dist = ops.bf_knn2_batched(qb.to(dev), tk.to(dev))
print("query 100 bits: idx", idx[0, 0].tolist(), "dist", dist[0, 0].tolist())
