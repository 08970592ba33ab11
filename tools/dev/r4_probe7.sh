#!/bin/bash
python -m pytest tests -m gpu -x -q > gpurun_out/pytest_r4g.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/pytest_r4g.log
for b in 1 0; do
MM_BATCH_DEBUG=1 timeout -k 10 500 python bench.py --frames 400 --height 2160 --width 3840 --nfeatures 8000 --ba-window 50 --ba-stride 25 \
      --steps 1 --warmup 1 --no-cpu-baseline --ba-batched $b > gpurun_out/c5p_b$b.log 2> gpurun_out/c5p_b$b.err
grep -v amdgpu.ids gpurun_out/c5p_b$b.err | head -5
grep "^{" gpurun_out/c5p_b$b.log | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); s=j['sliding_window_ba']
print('batched', s.get('batched'), 'ms', round(s['ms'],1), 'nfev', s['nfev_total'], 'windows', s['windows'], 'max nfev', max(w[4] for w in s['per_window']))
for k in j['kernels_all_launches_extra_step'][:22]:
    print('  %-28s %8.1f launches %9.1f us avg %9.2f ms' % (k['kernel'], k['launches_per_step'], k['avg_us'], k['ms_per_step']))
print('  sum of listed ms', round(sum(k['ms_per_step'] for k in j['kernels_all_launches_extra_step']),1))
"
done
