#!/usr/bin/env python3
"""Idle time between kernels from a rocprofv3 --kernel-trace CSV: for every gap (next kernel starts after everything
before it has ended) charge the gap to the pair (kernel that ended last -> kernel that starts).  Prints the pairs with
the most idle time, restricted to the window between the first and the last launch of `--between` (default: the BA
normal-equation kernel), i.e. the bundle adjustment of the bench."""
import argparse
import csv
import glob
import re
from collections import defaultdict

ap = argparse.ArgumentParser()
ap.add_argument("root")
ap.add_argument("--between", default=None, help="default: ba_normal_eq_kernel (round 4: both block sweeps in one launch), else ba_point_blocks_kernel")
a = ap.parse_args()
rows = []
for f in glob.glob(a.root + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"]).split("(")[0].split("<")[0]
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
rows.sort()
between = a.between or ("ba_normal_eq_kernel" if any(r[2] == "ba_normal_eq_kernel" for r in rows) else "ba_point_blocks_kernel")
idx = [i for i, r in enumerate(rows) if r[2] == between]
lo, hi = idx[len(idx) // 2], idx[-1]          # second half: steady state of the last step
rows = rows[lo:hi + 1]
busy = 0
gaps = defaultdict(lambda: [0, 0])
end, last = rows[0][1], rows[0][2]
t0 = rows[0][0]
for s, e, n in rows[1:]:
    if s > end:
        g = gaps[(last, n)]
        g[0] += 1
        g[1] += s - end
    if e > end:
        end, last = e, n
by_kernel = defaultdict(lambda: [0, 0])
for s_, e_, n_ in rows:
    by_kernel[n_][0] += 1
    by_kernel[n_][1] += e_ - s_
span = end - t0
idle = sum(g[1] for g in gaps.values())
print(f"window {span / 1e6:.2f} ms, idle {idle / 1e6:.2f} ms ({100 * idle / span:.1f} %), {len(rows)} launches")
for (p, n), (c, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{t / 1e6:8.3f} ms  {c:6d} x {t / max(c, 1) / 1e3:7.1f} us   {p[:40]:40s} -> {n[:40]}")
n_it = max(by_kernel.get(between, [1])[0] - 1, 1)
print(f"--- kernel time inside the window ({n_it} iterations, {span / 1e3 / n_it:.0f} us each) ---")
for n_, (c, t) in sorted(by_kernel.items(), key=lambda kv: -kv[1][1])[:30]:
    print(f"{t / 1e6:8.3f} ms  {c:6d} x {t / max(c, 1) / 1e3:7.1f} us  = {t / 1e3 / n_it:7.1f} us/iteration   {n_[:60]}")
