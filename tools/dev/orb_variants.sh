#!/bin/bash
# per-kernel ORB times over compile-time variants of orb.hip: VARIANTS="-DX=1|-DX=2"
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT/meatmodeler_amd/csrc || exit 1
IFS='|' read -ra VS <<< "${VARIANTS}"
for v in "${VS[@]}"; do
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wall -Wno-unused-result -ffp-contract=off $v -c orb.hip -o orb.o || exit 1
  make > $ROOT/gpurun_out/make_var.log 2>&1 || { tail $ROOT/gpurun_out/make_var.log; exit 1; }
  echo "[$v] $(cd $ROOT && bash tools/dev/orb_stats.sh 2>&1 | grep -E 'fast|describe|pyramid' | awk '{printf "%s %s  ", $1, $(NF-1)}')"
done
rm -f orb.o && make > $ROOT/gpurun_out/make.log 2>&1
