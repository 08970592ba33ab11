#!/bin/bash
# shader clock INSIDE the matching kernel (s_memtime vs s_memrealtime per workgroup): rebuild bf_match.o with
# -DMM_BF_CLOCK, run the C3-shaped benchmark on one variant, restore the product build.  usage: bf_inkernel_clock.sh [variant]
v=${1:-314}
mkdir -p gpurun_out
cd meatmodeler_amd/csrc || exit 1
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -mllvm -amdgpu-mfma-vgpr-form=1 -DMM_BF_CLOCK -c bf_match.hip -o bf_match.o || exit 1
make > ../../gpurun_out/make_clk.log 2>&1 || { tail ../../gpurun_out/make_clk.log; exit 1; }
cd ../..
BF_ONLY=$v timeout -k 10 200 python - <<'PY' > gpurun_out/bf_inkernel_clock_$v.log 2>&1
import ctypes as C, os, runpy, sys, numpy as np
sys.argv = ["tools/bench_bf.py"]
runpy.run_path("tools/bench_bf.py", run_name="__main__")
from meatmodeler_amd._lib import LIB_PATH
raw = C.CDLL(LIB_PATH)
buf = (C.c_longlong * 8192)()
assert raw.mm_debug_bf_clock(buf) == 0
a = np.array(buf[:], dtype=np.float64).reshape(4096, 2)
a = a[a[:, 1] > 0]
clk = a[:, 0] / a[:, 1] * 0.1
print(f"in-kernel shader clock over {len(a)} workgroups: median {np.median(clk):.3f} GHz, p10 {np.percentile(clk, 10):.3f}, p90 {np.percentile(clk, 90):.3f}; workgroup life median {np.median(a[:, 1]) * 0.01:.1f} us")
PY
echo "clock rc=$?"; tail -3 gpurun_out/bf_inkernel_clock_$v.log
cd meatmodeler_amd/csrc && rm -f bf_match.o && make > ../../gpurun_out/make.log 2>&1 || { tail ../../gpurun_out/make.log; exit 1; }
