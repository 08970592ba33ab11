#!/bin/bash
# bench_chol over compile-time variants of chol.hip: VARIANTS="-DX=1|-DX=2 -DY=3|..." (| separated)
mkdir -p gpurun_out
cd meatmodeler_amd/csrc || exit 1
IFS='|' read -ra VS <<< "${VARIANTS}"
for v in "${VS[@]}"; do
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics $v -c chol.hip -o chol.o || exit 1
  make > ../../gpurun_out/make_var.log 2>&1 || { tail ../../gpurun_out/make_var.log; exit 1; }
  ( cd ../.. && timeout -k 10 120 python tools/bench_chol.py 3000 528 30 2>&1 | tail -1 | sed "s/^/[$v] /" | cut -c1-150 )
done
rm -f chol.o && make > ../../gpurun_out/make.log 2>&1
