#!/bin/bash
python tools/bench_schur.py 20 > gpurun_out/bench_schur_r4_pipe2deep.log 2>&1; grep "schur alone\|serial" gpurun_out/bench_schur_r4_pipe2deep.log
python -m pytest tests -m gpu -x -q -k "schur or damped or batched or adjust_points or library_trf or two_ranks" > gpurun_out/pytest_r4i.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_r4i.log
python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_r4b.json 2> gpurun_out/bench_r4b.err; python - <<'PY'
import json
j=json.loads([l for l in open('gpurun_out/bench_r4b.json') if l.startswith('{')][-1])
print('ms_per_step', round(j['ms_per_step'],1), 'stage', {k:round(v,2) for k,v in j['stage_ms'].items()}, 'ba', {k:j['ba'][k] for k in ('nfev','iterations','ms_per_iteration')})
for k in j['kernels_all_launches_extra_step'][:16]: print(' ', k['kernel'], round(k['launches_per_step'],1), round(k['avg_us'],1), round(k['ms_per_step'],2))
PY
for b in 0 1; do
timeout -k 10 500 python bench.py --frames 2000 --height 2160 --width 3840 --nfeatures 8000 --ba-window 50 --ba-stride 25 \
      --steps 1 --warmup 1 --no-cpu-baseline --no-profile --ba-batched $b > gpurun_out/c5_r4b_b$b.log 2> gpurun_out/c5_r4b_b$b.err
grep "^{" gpurun_out/c5_r4b_b$b.log | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); s=j['sliding_window_ba']
print('C5 batched', s.get('batched'), 'ms', round(s['ms'],1), 'nfev', s['nfev_total'], 'ms/eval', round(s['ms']/s['nfev_total'],4), 'max nfev', max(w[4] for w in s['per_window']), 'step ms', round(j['ms_per_step'],1))
"
done
