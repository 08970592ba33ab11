#!/bin/bash
mkdir -p gpurun_out
make -C meatmodeler_amd/csrc -j8 > gpurun_out/make.log 2>&1 || { tail gpurun_out/make.log; exit 1; }
timeout -k 10 900 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
rc=$?
echo "pytest rc=$rc"; tail -5 gpurun_out/pytest_gpu.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pytest timed out - stopping"; exit 1; fi
timeout -k 10 300 python tools/bench_bf.py > gpurun_out/bench_bf.log 2>&1
rc=$?; echo "bench_bf rc=$rc"; tail -8 gpurun_out/bench_bf.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf $ROOT/gpurun_out/pmc_$ctr
  timeout -k 10 600 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $ROOT/gpurun_out/pmc_$ctr -- python3 $ROOT/bench.py --frames 96 --steps 1 --warmup 1 --no-cpu-baseline > $ROOT/gpurun_out/pmc_$ctr.log 2>&1
  rc=$?; echo "pmc $ctr rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
  python3 $ROOT/tools/pmc_summary.py $ROOT/gpurun_out/pmc_$ctr $ctr > $ROOT/gpurun_out/pmc_${ctr}_summary.txt 2>&1
  head -16 $ROOT/gpurun_out/pmc_${ctr}_summary.txt
  find $ROOT/gpurun_out/pmc_$ctr -name "*.csv" -size +8M -delete
done
