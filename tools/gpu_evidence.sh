#!/bin/bash
# end-of-round evidence: parity tests, smoke, bench (default flags), rocprofv3 kernel stats, PMC traffic counters
mkdir -p gpurun_out
make -C meatmodeler_amd/csrc -j8 > gpurun_out/make.log 2>&1 || { tail gpurun_out/make.log; exit 1; }
timeout -k 10 900 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/pytest_gpu.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
rc=$?; echo "smoke rc=$rc"; tail -2 gpurun_out/smoke.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 900 python bench.py > gpurun_out/bench_default.log 2>&1
rc=$?; echo "bench rc=$rc"; tail -c 400 gpurun_out/bench_default.log
if [ $rc -ne 0 ]; then exit 1; fi
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/prof
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile > $ROOT/gpurun_out/rocprof.log 2>&1
rc=$?; echo "rocprof rc=$rc"
find $ROOT/gpurun_out/prof -name "*_kernel_trace.csv" -delete
# (counter collection serialises kernels: the overlapped build + solve runs one after the other there)
export MM_SCHUR_OVERLAP=0
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf $ROOT/gpurun_out/pmc_$ctr
  timeout -k 10 900 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $ROOT/gpurun_out/pmc_$ctr -- python3 $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile > $ROOT/gpurun_out/pmc_$ctr.log 2>&1
  rc=$?; echo "pmc $ctr rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
  python3 $ROOT/tools/pmc_summary.py $ROOT/gpurun_out/pmc_$ctr $ctr > $ROOT/gpurun_out/pmc_${ctr}_summary.txt 2>&1
  find $ROOT/gpurun_out/pmc_$ctr -name "*.csv" -size +2M -delete
done
head -14 $ROOT/gpurun_out/pmc_FETCH_SIZE_summary.txt
unset MM_SCHUR_OVERLAP
cd $ROOT && ./tools/prof_chol.sh > gpurun_out/prof_chol_summary.txt 2>&1; tail -4 gpurun_out/prof_chol_summary.txt | cut -c1-160
cd $ROOT && ./tools/gpu_trace.sh > gpurun_out/trace_summary.txt 2>&1; head -4 gpurun_out/trace_summary.txt
cd $ROOT && ./tools/dev/orb_stats.sh > gpurun_out/orb_kernels.txt 2>&1; tail -8 gpurun_out/orb_kernels.txt
