#!/bin/bash
MM_BATCH_DEBUG=1 timeout -k 10 300 python bench.py --frames 200 --height 2160 --width 3840 --nfeatures 8000 --ba-window 50 --ba-stride 25 \
      --steps 1 --warmup 0 --no-cpu-baseline --no-profile --ba-batched 1 > gpurun_out/c5q.log 2> gpurun_out/c5q.err
grep -v amdgpu.ids gpurun_out/c5q.err | head -20
grep "^{" gpurun_out/c5q.log | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); s=j['sliding_window_ba']
print('batched', s.get('batched'), 'ms', round(s['ms'],1), 'nfev', s['nfev_total'], 'windows', s['windows'], 'max nfev', max(w[4] for w in s['per_window']))
"
