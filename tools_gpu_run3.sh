#!/bin/bash
mkdir -p gpurun_out
make -C meatmodeler_amd/csrc -j8 > gpurun_out/make.log 2>&1 || { tail gpurun_out/make.log; exit 1; }
timeout -k 10 900 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
rc=$?
echo "pytest rc=$rc"; tail -8 gpurun_out/pytest_gpu.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pytest timed out - stopping"; exit 1; fi
timeout -k 10 900 python bench.py --steps 2 --warmup 1 > gpurun_out/bench_full.log 2>&1
rc=$?; echo "bench500 rc=$rc"; tail -c 5000 gpurun_out/bench_full.log
if [ $rc -ne 0 ]; then exit 1; fi
cd /tmp && export TMPDIR=/tmp
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1
rc=$?; echo "rocprof rc=$rc"; tail -3 $GRAFT_REPO_ROOT/gpurun_out/rocprof.log | cut -c1-300
cd $GRAFT_REPO_ROOT
find gpurun_out/prof -name "*kernel_stats*" | head; find gpurun_out/prof -name "*_kernel_trace.csv" -size +20M -delete
ls -la gpurun_out/prof/* | head
