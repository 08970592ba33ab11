#!/bin/bash
# kernel durations of the banded solve benchmark (rocprofv3 --kernel-trace --stats)
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/prof_chol
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_chol -- python3 $ROOT/tools/bench_chol.py "$@" > $ROOT/gpurun_out/prof_chol.log 2>&1
echo "rc=$?"; tail -2 $ROOT/gpurun_out/prof_chol.log
f=$(find $ROOT/gpurun_out/prof_chol -name "*kernel_stats.csv" | head -1)
head -8 "$f" | cut -c1-200
find $ROOT/gpurun_out/prof_chol -name "*_kernel_trace.csv" -delete
