#!/usr/bin/env python3
"""profiles/<round>_pmc_fetch_size.txt + _pmc_write_size.txt (tools/pmc_summary.py output) -> <round>_pmc_traffic.json,
the per-launch HBM traffic table bench.py reads for `roofline.traffic`."""
import json
import sys

fetch, write, out = sys.argv[1], sys.argv[2], sys.argv[3]


def parse(path):
    rows = {}
    for line in open(path):
        if line.startswith("#") or not line.strip():
            continue
        parts = line.split()
        try:
            total, mean, n = float(parts[-1]), float(parts[-2]), int(parts[-3])
        except (ValueError, IndexError):
            continue
        rows[" ".join(parts[:-3])] = (n, mean)
    return rows


f, w = parse(fetch), parse(write)
kern = {}
for name in sorted(set(f) | set(w)):
    if name.startswith(("at::", "rocprim", "__amd", "Cijk", "rocblas")):
        continue
    e = {}
    if name in f:
        e["FETCH_SIZE_KB_per_launch"] = f[name][1]
        e["launches"] = f[name][0]
    if name in w:
        e["WRITE_SIZE_KB_per_launch"] = w[name][1]
        e.setdefault("launches", w[name][0])
    kern[name] = e
json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes on `MM_SCHUR_OVERLAP=0 bench.py --steps 1 "
                   "--warmup 1` (C3: 500x1080p, 4000 kpts; counter collection serialises kernels, so the overlapped build + "
                   "solve runs one after the other there); mean per launch in KB as reported (guide: gfx950 FETCH_SIZE reads "
                   "1/2 of wide 16-B streaming reads; 4/8-byte gathers are uncalibrated -> FETCH is NOT doubled here)",
           "kernels": kern}, open(out, "w"), indent=1)
print(len(kern), "kernels ->", out)
