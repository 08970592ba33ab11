#!/bin/bash
# quick GPU check: selected parity tests (-k "$1"), then a short bench without the CPU leg
mkdir -p gpurun_out
make -C meatmodeler_amd/csrc -j8 > gpurun_out/make.log 2>&1 || { tail gpurun_out/make.log; exit 1; }
timeout -k 10 600 python -m pytest tests -m gpu -q --timeout 240 -p no:cacheprovider -k "${1:-link or pipeline}" > gpurun_out/pytest_quick.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -15 gpurun_out/pytest_quick.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 600 python bench.py --no-cpu-baseline ${2:-} > gpurun_out/bench_quick.log 2>&1
rc=$?; echo "bench rc=$rc"; tail -c 3000 gpurun_out/bench_quick.log
