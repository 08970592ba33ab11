#!/bin/bash
# after `gpurun -- ./tools/gpu_evidence.sh`: copy the summaries from gpurun_out/ (scratch) into profiles/ (tracked)
set -e
cd "$(dirname "$0")/.."
R=${1:-r01}
grep "^{" gpurun_out/bench_default.log | tail -1 > profiles/${R}_bench_default.json
# (gpurun merges every call's files into gpurun_out/: take the newest)
newest() { find "$1" -name "$2" -printf '%T@ %p\n' | sort -n | tail -1 | cut -d' ' -f2-; }
cp "$(newest gpurun_out/prof '*kernel_stats.csv')" profiles/${R}_rocprofv3_kernel_stats.csv
cp gpurun_out/pmc_FETCH_SIZE_summary.txt profiles/${R}_pmc_fetch_size.txt
cp gpurun_out/pmc_WRITE_SIZE_summary.txt profiles/${R}_pmc_write_size.txt
python3 tools/pmc_to_json.py profiles/${R}_pmc_fetch_size.txt profiles/${R}_pmc_write_size.txt profiles/${R}_pmc_traffic.json
cp "$(newest gpurun_out/prof_chol '*kernel_stats.csv')" profiles/${R}_chol_kernel_stats.csv
cp gpurun_out/trace_gaps.txt profiles/${R}_trace_gaps.txt
grep -v rocprofv3 gpurun_out/orb_kernels.txt > profiles/${R}_orb_kernels.txt
ls -la profiles | tail -n +2
