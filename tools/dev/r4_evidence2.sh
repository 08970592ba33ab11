#!/bin/bash
# round-4 evidence, part 2: the C5 shape on one GPU (8 streams / lock-step batched), a smaller-window case where the lock-step
# solve pays, and the self-launched two-rank rehearsal (gloo, both ranks on this GPU)
c5() {
  name=$1; shift
  timeout -k 10 500 python bench.py --frames 2000 --height 2160 --width 3840 --nfeatures 8000 --ba-window 50 --ba-stride 25 \
      --ba-window-max-nfev 500 --steps 1 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/c5_$name.log 2> gpurun_out/c5_$name.err
  grep "^{" gpurun_out/c5_$name.log | tail -1 > gpurun_out/bench_c5_$name.json
  python - <<PY
import json
j=json.load(open('gpurun_out/bench_c5_$name.json')); s=j['sliding_window_ba']
print('$name', 'batched', s.get('batched'), 'ms', round(s['ms'],1), 'nfev', s['nfev_total'], 'windows', s['windows'], 'max nfev', max(w[4] for w in s['per_window']), 'step ms', round(j['ms_per_step'],1), 'detect', round(j['stage_ms']['detect'],1), 'link', round(j['stage_ms']['link'],2))
PY
}
c5 streams8 --ba-batched 0 --ba-streams 8
c5 batched --ba-batched 1
# small windows (1080p, 2000 key points: ~9 k points per window): launch-latency bound -> the lock-step solve pays
for b in 0 1; do
timeout -k 10 300 python bench.py --frames 1000 --nfeatures 2000 --ba-window 50 --ba-stride 25 --steps 1 --warmup 1 --no-cpu-baseline --no-profile \
    --ba-batched $b > gpurun_out/small_b$b.log 2> gpurun_out/small_b$b.err
grep "^{" gpurun_out/small_b$b.log | tail -1 > gpurun_out/bench_smallwin_b$b.json
python - <<PY
import json
j=json.load(open('gpurun_out/bench_smallwin_b$b.json')); s=j['sliding_window_ba']
print('small windows batched', s.get('batched'), 'ms', round(s['ms'],1), 'nfev', s['nfev_total'], 'windows', s['windows'], 'points/window', round(sum(w[2] for w in s['per_window'])/len(s['per_window'])))
PY
done
MM_DIST_BACKEND=gloo timeout -k 10 400 python bench.py --gpus 2 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/n2.log 2> gpurun_out/n2.err
grep "^{" gpurun_out/n2.log | tail -1 > gpurun_out/bench_n2_gloo.json
python - <<PY
import json
j=json.load(open('gpurun_out/bench_n2_gloo.json'))
print('n_gpus', j['n_gpus'], 'ms_per_step', round(j['ms_per_step'],1), j['ba'])
PY
