#!/bin/bash
mkdir -p gpurun_out
make -C meatmodeler_amd/csrc -j8 > gpurun_out/make.log 2>&1 || { tail gpurun_out/make.log; exit 1; }
timeout -k 10 900 python -m pytest tests -m gpu -q -s --timeout 300 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
rc=$?
echo "pytest rc=$rc"; tail -4 gpurun_out/pytest_gpu.log; grep "real-match" gpurun_out/pytest_gpu.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pytest timed out - stopping"; exit 1; fi
timeout -k 10 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_full_1.log 2>&1
rc=$?; echo "bench500 rc=$rc"
python3 - <<PY
import json
l=[x for x in open('gpurun_out/bench_full_1.log') if x.startswith('{')]
d=json.loads(l[-1])
print(d['value'], d['ms_per_step'], d['frames_per_s'], d['ba_residuals_per_s']); print(d['stage_ms']); print(d['problem']); print(d['roofline'])
for k in d['kernels_all_launches_extra_step'][:16]: print('  %-28s %8.1f launches %9.3f ms/step %9.2f us avg'%(k['kernel'],k['launches_per_step'],k['ms_per_step'],k['avg_us']))
PY
