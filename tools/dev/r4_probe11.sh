#!/bin/bash
python -m pytest tests -m gpu -x -q > gpurun_out/pytest_r4k.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_r4k.log
python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_r4c.json 2> gpurun_out/bench_r4c.err; python - <<'PY'
import json
j=json.loads([l for l in open('gpurun_out/bench_r4c.json') if l.startswith('{')][-1])
print('ms_per_step', round(j['ms_per_step'],1), 'stage', {k:round(v,2) for k,v in j['stage_ms'].items()}, 'ba', {k:j['ba'][k] for k in ('nfev','iterations','ms_per_iteration')})
for k in j['kernels_all_launches_extra_step'][:16]: print(' ', k['kernel'], round(k['launches_per_step'],1), round(k['avg_us'],1), round(k['ms_per_step'],2))
PY
