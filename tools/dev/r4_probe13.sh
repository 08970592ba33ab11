#!/bin/bash
python -m pytest tests -m gpu -x -q > gpurun_out/pytest_r4l.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_r4l.log
for g in 512 1024 2048; do
MM_VEC_GRID=$g python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_r4_vg$g.json 2> gpurun_out/bench_r4_vg$g.err; python - <<PY
import json
j=json.loads([l for l in open('gpurun_out/bench_r4_vg$g.json') if l.startswith('{')][-1])
print('VEC_GRID=$g ms_per_step', round(j['ms_per_step'],1), 'ba', {k:j['ba'][k] for k in ('nfev','iterations','ms_per_iteration')})
for k in j['kernels_all_launches_extra_step'][:12]:
    if k['kernel'] in ('fused_vec_kernel','ba_normal_eq_kernel','ba_damp_kernel','chol_band_fused_kernel','schur_pairs_kernel','chol_band_bwd_kernel','ba_backsub_points_kernel'): print(' ', k['kernel'], round(k['launches_per_step'],1), round(k['avg_us'],1))
PY
done
