#!/bin/bash
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "chol or adjust_points or library_trf or batched or schur or damped" > gpurun_out/pytest_r4j.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_r4j.log
timeout -k 10 120 python tools/bench_chol.py > gpurun_out/bench_chol_r4b.log 2>&1; tail -1 gpurun_out/bench_chol_r4b.log
timeout -k 10 120 python tools/bench_schur.py 20 2>&1 | grep "schur alone\|serial"
