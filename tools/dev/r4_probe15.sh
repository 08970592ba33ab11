#!/bin/bash
python tools/bench_schur.py 20 2>&1 | grep "schur alone"
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "schur or damped or adjust_points or library_trf or batched or two_ranks or real_matches" > gpurun_out/pytest_r4n.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_r4n.log
python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_r4d.json 2> gpurun_out/bench_r4d.err; python - <<'PY'
import json
j=json.loads([l for l in open('gpurun_out/bench_r4d.json') if l.startswith('{')][-1])
print('ms_per_step', round(j['ms_per_step'],1), 'ba', {k:j['ba'][k] for k in ('nfev','iterations','ms_per_iteration')})
for k in j['kernels_all_launches_extra_step'][:3]: print(' ', k['kernel'], round(k['launches_per_step'],1), round(k['avg_us'],1))
PY
