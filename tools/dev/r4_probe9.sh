#!/bin/bash
for d in 0 1 2 4 3 7; do
  MM_SCHUR_DEBUG=$d python tools/bench_schur.py 20 2>&1 | grep "schur alone" | sed "s/^/dbg=$d /"
done
