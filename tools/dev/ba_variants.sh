#!/bin/bash
# bench.py over compile-time variants of one csrc file: FILE=ba.hip VARIANTS="-DX=1|-DX=2" KERNELS="ba_normal_eq_kernel ba_backsub_points_kernel"
mkdir -p gpurun_out
cd meatmodeler_amd/csrc || exit 1
obj=${FILE%.hip}.o
IFS='|' read -ra VS <<< "${VARIANTS}"
for v in "${VS[@]}"; do
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wall -Wno-unused-result $v -c $FILE -o $obj || exit 1
  make > ../../gpurun_out/make_var.log 2>&1 || { tail ../../gpurun_out/make_var.log; exit 1; }
  ( cd ../.. && python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_var.json 2>/dev/null; KERNELS="$KERNELS" V="$v" python - <<'PY'
import json, os
d = json.loads(open("gpurun_out/bench_var.json").read().strip().splitlines()[-1])
ks = {k["kernel"]: k for k in d["kernels_all_launches_extra_step"]}
print("[%s] ms/iteration %.4f nfev %d | " % (os.environ["V"], d["ba"]["ms_per_iteration"], d["ba"]["nfev"]) +
      " ".join("%s %.1f" % (n, ks[n]["avg_us"]) for n in os.environ["KERNELS"].split() if n in ks))
PY
  )
done
rm -f $obj && make > ../../gpurun_out/make.log 2>&1
