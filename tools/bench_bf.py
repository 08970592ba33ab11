#!/usr/bin/env python3
"""Tuning run for the BF Hamming kernel: times every (queries-per-lane, unroll) variant on the C3 shape
(499 pairs of 4000 x 4000 descriptors) in one process, interleaved rounds (guide rule 24)."""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from meatmodeler_amd import ops
from meatmodeler_amd._lib import default_context, Timer

dev = torch.device("cuda", 0)
ctx = default_context()
F, N = int(os.environ.get("BF_F", 500)), int(os.environ.get("BF_N", 4000))
g = torch.Generator(device="cpu").manual_seed(0)
desc = torch.randint(0, 256, (F, N, 32), dtype=torch.uint8, generator=g).to(dev)
pairs = (F - 1) * N * N
variants = [int(os.environ['BF_ONLY'])] if os.environ.get('BF_ONLY') else [114, 200, 300, 310, 314]
res = {v: [] for v in variants}
t = Timer(ctx)
ref = None
for rnd in range(6):
    for v in variants:
        os.environ["MM_BF_VARIANT"] = str(v)
        t.start()
        idx, dist = ops.bf_knn2_batched(desc[:-1], desc[1:])
        t.stop()
        ms = t.elapsed_ms()
        if rnd:
            res[v].append(ms)
        if ref is None:
            ref = (idx.clone(), dist.clone())
        else:
            assert torch.equal(idx, ref[0]) and torch.equal(dist, ref[1]), v
for v in variants:
    m = np.array(res[v])
    kind = "matrix cores (FP4 MFMA, tile minima)" if v == 310 else "matrix cores (FP4 MFMA, tile minima, 4 waves/SIMD)" if v == 314 else "matrix cores (FP4 MFMA)" if v == 300 else "matrix cores (int8 MFMA)" if v == 200 else "LDS-fed xor/popcount" if v >= 100 else "SGPR-fed xor/popcount"
    print(f"variant {v} ({kind}): median {np.median(m):.3f} ms  min {m.min():.3f} ms  {pairs / np.median(m) / 1e9:.2f} T pairs/s")
