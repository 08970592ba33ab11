#!/bin/bash
# kernel stats of the sliding-window adjustment at BASELINE config 5's window shape (300 frames -> 11 windows)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_c5
timeout -k 10 800 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c5 -- python3 $ROOT/bench.py --frames 300 --height 2160 --width 3840 --nfeatures 8000 --batch 64 --ba-window 50 --ba-stride 25 --steps 1 --warmup 0 --no-cpu-baseline --no-profile "$@" > $ROOT/gpurun_out/c5_prof.json 2> $ROOT/gpurun_out/c5_prof.err
cp $(find /tmp/prof_c5 -name '*kernel_stats.csv' | head -1) $ROOT/gpurun_out/c5_kernel_stats.csv
