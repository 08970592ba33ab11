#!/bin/bash
# kernel-trace of one bench step and the idle-gap table of its bundle adjustment
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/trace
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/trace -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile > $ROOT/gpurun_out/trace.log 2>&1
echo "rc=$?"
python3 $ROOT/tools/trace_gaps.py $ROOT/gpurun_out/trace | tee $ROOT/gpurun_out/trace_gaps.txt
find $ROOT/gpurun_out/trace -name "*.csv" -size +1M -delete
