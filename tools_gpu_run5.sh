#!/bin/bash
mkdir -p gpurun_out
make -C meatmodeler_amd/csrc -j8 > gpurun_out/make.log 2>&1 || { tail gpurun_out/make.log; exit 1; }


timeout -k 10 900 python -m pytest tests -m gpu -q -s --timeout 300 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
rc=$?
echo "pytest rc=$rc"; tail -8 gpurun_out/pytest_gpu.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pytest timed out - stopping"; exit 1; fi
for i in 1 2; do
timeout -k 10 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_full_$i.log 2>&1
rc=$?; echo "bench500 run $i rc=$rc"; tail -c 3300 gpurun_out/bench_full_$i.log | head -c 3000
echo
done
