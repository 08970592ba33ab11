#!/bin/bash
for ch in 256 512 1024 4096; do
  MM_SCHUR_CHUNK=$ch python tools/bench_schur.py 20 2>&1 | grep "chunks\|schur alone" | sed "s/^/CH=$ch /"
done
for ch in 256 1024; do
MM_SCHUR_CHUNK=$ch python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_r4_ch$ch.json 2> gpurun_out/bench_r4_ch$ch.err; python - <<PY
import json
j=json.loads([l for l in open('gpurun_out/bench_r4_ch$ch.json') if l.startswith('{')][-1])
print('CH=$ch ms_per_step', round(j['ms_per_step'],1), 'ba', {k:j['ba'][k] for k in ('nfev','iterations','ms_per_iteration')})
for k in j['kernels_all_launches_extra_step'][:4]: print(' ', k['kernel'], round(k['launches_per_step'],1), round(k['avg_us'],1))
PY
done
