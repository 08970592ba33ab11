#!/bin/bash
# round-4 probe 3: sliding-window BA at the C5 shape -- do more hardware queues / streams lift the 8-stream plateau?
run() {  # name, env..., -- bench args
  name=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout -k 10 400 python bench.py --frames 2000 --height 2160 --width 3840 --nfeatures 8000 --ba-window 50 --ba-stride 25 \
      --steps 1 --warmup 1 --no-cpu-baseline --no-profile "$@" > gpurun_out/c5_$name.log 2> gpurun_out/c5_$name.err
  grep "^{" gpurun_out/c5_$name.log | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); s=j['sliding_window_ba']
print('$name', 'streams', s['streams'], 'ms', round(s['ms'],1), 'nfev', s['nfev_total'], 'ms/eval', round(s['ms']/s['nfev_total'],4), 'max nfev', max(w[4] for w in s['per_window']))
"
}
run s8_q4 A=1 -- --ba-streams 8
run s8_q8 GPU_MAX_HW_QUEUES=8 -- --ba-streams 8
run s16_q16 GPU_MAX_HW_QUEUES=16 -- --ba-streams 16
run s32_q32 GPU_MAX_HW_QUEUES=32 -- --ba-streams 32
run s16_q16_nospin GPU_MAX_HW_QUEUES=16 MM_TRF_SPIN=0 -- --ba-streams 16
python bench.py --steps 3 --warmup 1 > gpurun_out/bench_r4a.json 2> gpurun_out/bench_r4a.err; python - <<'PY'
import json
j=json.loads([l for l in open('gpurun_out/bench_r4a.json') if l.startswith('{')][-1])
print('ms_per_step', round(j['ms_per_step'],1), 'stage', {k:round(v,2) for k,v in j['stage_ms'].items()}, 'ba', {k:j['ba'][k] for k in ('nfev','iterations','ms_per_iteration')})
for k in j['kernels_all_launches_extra_step'][:14]: print(k['kernel'], round(k['launches_per_step'],1), round(k['avg_us'],1), round(k['ms_per_step'],2))
PY
