#!/bin/bash
# phase trace of the single-launch banded factorisation: rebuild chol.o with -DMM_CHOL_TRACE, run tools/dev/chol_trace.py,
# then restore the product build.  usage: tools/dev/chol_trace.sh [tag]
tag=${1:-run}
mkdir -p gpurun_out
cd meatmodeler_amd/csrc || exit 1
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -DMM_CHOL_TRACE -c chol.hip -o chol.o || exit 1
make > ../../gpurun_out/make_trace.log 2>&1 || { tail ../../gpurun_out/make_trace.log; exit 1; }
cd ../..
timeout -k 10 200 python tools/dev/chol_trace.py > gpurun_out/chol_trace_$tag.log 2>&1
echo "trace rc=$?"; tail -8 gpurun_out/chol_trace_$tag.log
cd meatmodeler_amd/csrc && rm -f chol.o && make > ../../gpurun_out/make.log 2>&1 || { tail ../../gpurun_out/make.log; exit 1; }
