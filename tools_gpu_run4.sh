#!/bin/bash
mkdir -p gpurun_out
make -C meatmodeler_amd/csrc -j8 > gpurun_out/make.log 2>&1 || { tail gpurun_out/make.log; exit 1; }
timeout -k 10 900 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
rc=$?
echo "pytest rc=$rc"; tail -8 gpurun_out/pytest_gpu.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pytest timed out - stopping"; exit 1; fi
timeout -k 10 300 python tools/bench_bf.py > gpurun_out/bench_bf.log 2>&1
rc=$?; echo "bench_bf rc=$rc"; cat gpurun_out/bench_bf.log | tail -12
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
timeout -k 10 900 python bench.py --steps 2 --warmup 1 > gpurun_out/bench_full.log 2>&1
rc=$?; echo "bench500 rc=$rc"; tail -c 4500 gpurun_out/bench_full.log
