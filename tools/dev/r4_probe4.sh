#!/bin/bash
# round-4 probe 4: batched lock-step windows -- tests, then the C5 shape with and without it
python -m pytest tests -m gpu -x -q > gpurun_out/pytest_r4h.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_r4h.log
run() {
  name=$1; shift
  timeout -k 10 500 python bench.py --frames 2000 --height 2160 --width 3840 --nfeatures 8000 --ba-window 50 --ba-stride 25 \
      --steps 1 --warmup 1 --no-cpu-baseline --no-profile "$@" > gpurun_out/c5_$name.log 2> gpurun_out/c5_$name.err
  grep "^{" gpurun_out/c5_$name.log | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); s=j['sliding_window_ba']
print('$name', 'batched', s.get('batched'), 'ms', round(s['ms'],1), 'nfev', s['nfev_total'], 'ms/eval', round(s['ms']/s['nfev_total'],4), 'max nfev', max(w[4] for w in s['per_window']), 'step ms', round(j['ms_per_step'],1))
"
}
run batched --ba-batched 1
run streams8 --ba-batched 0 --ba-streams 8
