"""Phase timestamps of the row-head workgroups of the single-launch banded factorisation (build csrc with
-DMM_CHOL_TRACE): where the ~21 us per block column of the chain go."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from meatmodeler_amd import ops
from meatmodeler_amd._lib import lib, default_context, LIB_PATH
ctx = default_context(); dev = ctx.device
n, hb = int(os.environ.get("N", 3000)), int(os.environ.get("HB", 527))
rng = np.random.default_rng(0)
A = np.zeros((n, n))
for i in range(n):
    lo = max(0, i - hb)
    A[i, lo:i] = rng.normal(size=i - lo) * 0.01
A = A + A.T + np.eye(n) * 4
b = rng.normal(size=n)
for rep in range(3):
    S = torch.as_tensor(A, device=dev).clone(); v = torch.as_tensor(b, device=dev).clone()
    info = ops.chol_solve_sym(S, v, ctx, hb, True)
    torch.cuda.synchronize()
x = v.cpu().numpy()
print("info", int(info), "residual", np.abs(A @ x - b).max())
raw = C.CDLL(LIB_PATH)
buf = (C.c_ulonglong * (128 * 32))()
assert raw.mm_debug_chol_trace(buf) == 0
full = np.array(buf[:], dtype=np.int64).reshape(128, 32) * 10e-3   # 100 MHz -> us
t = full[:, :10]
rows = [r for r in range(64) if t[r, 9] > 0]
rows1 = [r for r in range(64, 128) if t[r, 9] > 0]
print('side 1: row end of its rows (us):', [round(float(t[r, 9] - t[rows, 0].min()), 1) for r in rows1])
print('side 1: diagonal block factored at (us):', [round(float(t[r, 7] - t[rows, 0].min()), 1) for r in rows1])
t0 = t[rows, 0].min()
# stamps: 0 row start, 1 tile products of the earlier columns done, 2 last inverse X_33 of the block above seen, 3 last
# 16 columns of L_{r,r-1} solved, 4 streamed solve + rank-64 update done, 5 block (r, r-1) published, 6 diagonal block
# staged, 7 factored (its last stage is out), 8 = 7, 9 row end (inverse blocks, forward substitution)
names = ["start", "acc_done", "x33_seen", "fin3", "solved", "lastpan", "staged", "factor", "lastcol", "row_end"]
print("row  " + " ".join(f"{x:>9s}" for x in names) + "   step(factor done - prev)")
prev = None
for r in rows:
    rel = t[r] - t0
    step = "" if prev is None else f"{t[r, 7] - prev:8.2f}"
    print(f"{r:3d}  " + " ".join(f"{x:9.2f}" for x in rel) + "   " + step)
    prev = t[r, 7]
d = np.diff(t[rows][:, 7])
print("median step", np.median(d))
seg = t[rows][:, 1:] - t[rows][:, :-1]
print("median per-phase (us):", dict(zip(names[1:], np.round(np.median(seg[1:], 0), 2))))

# inside the 64 x 64 factorisation: 10 first panel done, 11 panel 1 brought up to date, 12 panel 1 done, 13 panel 2 done,
# 14 panel 3 brought up to date; 7 = all done
fr = full[rows]
inner = np.stack([fr[:, 10] - fr[:, 6], fr[:, 11] - fr[:, 10], fr[:, 12] - fr[:, 11], fr[:, 13] - fr[:, 12], fr[:, 14] - fr[:, 13],
                  fr[:, 7] - fr[:, 14]], 1)
print("factor phases (us, median): panel0 %.2f | update1 %.2f | panel1 %.2f | update2+panel2 %.2f | update3 %.2f | panel3 + tail %.2f"
      % tuple(np.median(inner[1:], 0)))
print("last column (us, median): reached before the previous factorisation ended by %.2f | first panel pair seen after %.2f | last pair after %.2f | acc_done after %.2f"
      % (np.median((fr[1:, 8] - fr[:-1, 7])), np.median(fr[1:, 15] - fr[1:, 8]), np.median(fr[1:, 5] - fr[1:, 8]), np.median(fr[1:, 1] - fr[1:, 8])))
print("last pair of panels seen relative to the fin3 of the row above: %.2f" % np.median(fr[2:, 5] - fr[1:-1, 3]))
sel = [i for i, r in enumerate(rows) if 4 <= r <= 17]
print("column r-3: own block (r, r-3) seen %.2f, block above (r-1, r-3) seen %.2f us after the factorisation of L_{r-3,r-3} ended; column r-4: %.2f / %.2f after that of L_{r-4,r-4}"
      % (np.median([fr[i, 24] - fr[i - 3, 7] for i in sel]), np.median([fr[i, 25] - fr[i - 3, 7] for i in sel]),
         np.median([fr[i, 26] - fr[i - 4, 7] for i in sel]), np.median([fr[i, 27] - fr[i - 4, 7] for i in sel])))
print("last column reached %.2f us after the factorisation of L_{r-3,r-3} ended" % np.median([fr[i, 8] - fr[i - 3, 7] for i in sel]))
