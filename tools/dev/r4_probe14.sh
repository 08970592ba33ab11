#!/bin/bash
for v in "lean:A=1" "pipe1:MM_SCHUR_PAIRS=pipe MM_SCHUR_GRID_MULT=1" "pipe2:MM_SCHUR_PAIRS=pipe MM_SCHUR_GRID_MULT=2" "pipe4:MM_SCHUR_PAIRS=pipe MM_SCHUR_GRID_MULT=4"; do
  name=${v%%:*}; envs=${v#*:}
  env $envs python tools/bench_schur.py 20 2>&1 | grep "schur alone" | sed "s/^/$name /"
done
MM_SCHUR_PAIRS=pipe timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "schur or damped or adjust_points or library_trf or batched or two_ranks" > gpurun_out/pytest_r4m.log 2>&1; echo "pytest(pipe) rc=$?"; tail -3 gpurun_out/pytest_r4m.log
