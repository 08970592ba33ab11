#!/bin/bash
# round-4 probe 2: pipelined pair kernel (grid multipliers) against the one-chunk-per-wave kernels; bwd kernel; tests
python -m pytest tests -m gpu -x -q > gpurun_out/pytest_r4e.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_r4e.log
for v in "pipe2:MM_SCHUR_GRID_MULT=2" "pipe4:MM_SCHUR_GRID_MULT=4" "pipe8:MM_SCHUR_GRID_MULT=8" "pipe16:MM_SCHUR_GRID_MULT=16" "lean:MM_SCHUR_PAIRS=lean"; do
  name=${v%%:*}; envs=${v#*:}
  env $envs python tools/bench_schur.py 20 > gpurun_out/bench_schur_r4_$name.log 2>&1
  echo "== $name"; grep "schur alone\|serial" gpurun_out/bench_schur_r4_$name.log
done
python tools/bench_chol.py > gpurun_out/bench_chol_r4a.log 2>&1; tail -6 gpurun_out/bench_chol_r4a.log
