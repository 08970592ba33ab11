#!/bin/bash
# round-4 probe 5: where does a lock-step round go?  C5 shape on 400 frames (15 windows), per-kernel events
for b in 1 0; do
timeout -k 10 500 python bench.py --frames 400 --height 2160 --width 3840 --nfeatures 8000 --ba-window 50 --ba-stride 25 \
      --steps 1 --warmup 1 --no-cpu-baseline --ba-batched $b > gpurun_out/c5p_b$b.log 2> gpurun_out/c5p_b$b.err
grep "^{" gpurun_out/c5p_b$b.log | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); s=j['sliding_window_ba']
print('batched', s.get('batched'), 'ms', round(s['ms'],1), 'nfev', s['nfev_total'], 'windows', s['windows'], 'max nfev', max(w[4] for w in s['per_window']))
tot=0
for k in j['kernels_all_launches_extra_step'][:30]:
    print('  %-28s %8.1f launches %9.1f us avg %9.2f ms' % (k['kernel'], k['launches_per_step'], k['avg_us'], k['ms_per_step']))
print('  sum of listed ms', round(sum(k['ms_per_step'] for k in j['kernels_all_launches_extra_step']),1))
"
done
