#!/bin/bash
# causal profile of the single-launch factorisation: ~1 us of delay injected at one site per build (csrc/chol.hip MM_DELAY):
#   1 chain wave before the last inverse | 7 chain wave before the first panel | 5 helper wave before it publishes L panels
#   2 row head after its last solve stage arrived | 4 the other consumers' last stage | 3 row head before staging its diagonal block
#   6 row head at the start of its last column's products
#   41 / 42 the last solve stage of the d = 2 / d >= 3 owners | 43 / 44 of the d >= 2 owners inside / outside M x M
#   8 d >= 2 owners between their products and their solve | 9 pre-accumulators of M x M before they publish
mkdir -p gpurun_out
cd meatmodeler_amd/csrc || exit 1
for site in ${SITES:-0 1 7 5 2 4 3 6}; do
  if [ $site -eq 0 ]; then extra=""; else extra="-DMM_CHOL_DELAY_SITE=$site"; fi
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics $extra $CHOL_EXTRA -c chol.hip -o chol.o || exit 1
  make > ../../gpurun_out/make_delay.log 2>&1 || { tail ../../gpurun_out/make_delay.log; exit 1; }
  ( cd ../.. && timeout -k 10 120 python tools/bench_chol.py 3000 528 30 2>&1 | tail -1 | sed "s/^/site $site: /" )
done
rm -f chol.o && make > ../../gpurun_out/make.log 2>&1
