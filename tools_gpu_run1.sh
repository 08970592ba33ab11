#!/bin/bash
# first GPU contact: parity tests, smoke, a small bench
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
rc=$?
echo "pytest rc=$rc"
tail -5 gpurun_out/pytest_gpu.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pytest timed out - stopping"; exit 1; fi
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
rc=$?; echo "smoke rc=$rc"; tail -3 gpurun_out/smoke.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
timeout -k 10 600 python bench.py --frames 40 --steps 2 --warmup 1 > gpurun_out/bench_small.log 2>&1
rc=$?; echo "bench rc=$rc"; tail -c 3000 gpurun_out/bench_small.log
