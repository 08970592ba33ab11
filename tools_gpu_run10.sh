#!/bin/bash
mkdir -p gpurun_out
make -C meatmodeler_amd/csrc -j8 > gpurun_out/make.log 2>&1 || { tail gpurun_out/make.log; exit 1; }
for mode in "" "--no-profile"; do
timeout -k 10 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline $mode > gpurun_out/bench_mode.log 2>&1
rc=$?; echo "bench500 [$mode] rc=$rc"
python3 - <<PY
import json
l=[x for x in open('gpurun_out/bench_mode.log') if x.startswith('{')]
d=json.loads(l[-1])
print(d['ms_per_step'], d['stage_ms'], d['problem']['ba_nfev'])
PY
done
