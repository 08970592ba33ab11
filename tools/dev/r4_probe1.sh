#!/bin/bash
# round-4 probe: GPU tests, the pair-kernel variants, where a sliding window's time goes
python -m pytest tests -m gpu -x -q > gpurun_out/pytest_r4d.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_r4d.log
for v in "lean2:MM_SCHUR_OCC=2" "lean3:MM_SCHUR_OCC=3" "ref:MM_SCHUR_PAIRS=ref"; do
  name=${v%%:*}; envs=${v#*:}
  env $envs python tools/bench_schur.py 20 > gpurun_out/bench_schur_r4_$name.log 2>&1
  echo "== $name"; grep "schur alone\|serial" gpurun_out/bench_schur_r4_$name.log
done
timeout -k 10 400 python tools/dev/window_phases.py > gpurun_out/window_phases_r4.log 2>&1; tail -8 gpurun_out/window_phases_r4.log
