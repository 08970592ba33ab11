import os, sys, ctypes as C, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from meatmodeler_amd import ops, synth, frame_tables
from meatmodeler_amd._lib import default_context, lib, ptr
from oracle import frame_oracle as fo
ctx = default_context(); d = ctx.device
frames, _, _ = synth.render_orbit_frames(2, 640, 480, arc_deg=3.0, seed=8)
bgr = np.stack([frames[0], np.roll(frames[0], 7, 1), 255 - frames[1]], -1)
t = torch.as_tensor(bgr).to(d).unsqueeze(0).contiguous()
ops.increase_contrast(t)
g, cb, gi = ops._LAB_TABLES[str(d)]
H, W = 480, 640
wsb = lib.mm_contrast_workspace_bytes(1, W, H, 8, 8)
ws = torch.zeros(wsb, dtype=torch.uint8, device=d); out = torch.empty_like(t)
ctx.check(lib.mm_increase_contrast(ctx.h, ptr(t), 1, W, H, ptr(g), ptr(cb), ptr(gi), 3.5, 8, 8, ptr(out), None, ptr(ws), wsb), "x")
plane = (W * H + 255) // 256 * 256
wsh = ws.cpu().numpy()
L, A, B = (wsh[k * plane:k * plane + W * H].reshape(H, W) for k in range(3))
lab, _ = fo.lab_roundtrip(bgr, frame_tables.lab_tables())
for name, p, r in (("L", L, lab[..., 0]), ("A", A, lab[..., 1]), ("B", B, lab[..., 2])):
    bad = np.argwhere(p != r); print(name, "mismatch", len(bad), bad[:5].tolist(), p[0, :8], r[0, :8])
ref = fo.increase_contrast(bgr, frame_tables.lab_tables()); o = out[0].cpu().numpy()
bad = np.argwhere((o != ref).any(-1)); print("out mismatch", len(bad), bad[:8].tolist(), np.unique(bad[:, 1] % 4, return_counts=True))
