run() { tag=$1; shift; env "$@" timeout -k 10 300 python bench.py --frames 400 --nfeatures 8000 --ba-window 50 --ba-stride 25 --steps 1 --warmup 1 --no-cpu-baseline $EXTRA > gpurun_out/win_$tag.log 2>&1; grep "^{" gpurun_out/win_$tag.log | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); s=j['sliding_window_ba']
print('$tag', round(s['ms'],1), s['nfev_total'], round(s['ms']/s['nfev_total'],4), s['streams'])
ks={k['kernel']:k for k in j['kernels_all_launches_extra_step']}
for n in ('chol_band_fused_kernel','chol_band_bwd_kernel','chol_init_kernel','schur_pairs_kernel','chol_panel_kernel','chol_trailing_kernel'):
    if n in ks: print('   ',n, ks[n]['launches_per_step'], round(ks[n]['avg_us'],1))
"; }
run default A=1
run nofused MM_CHOL_FUSED=0
EXTRA="--ba-streams 1" run streams1 A=1
EXTRA="--ba-streams 1" run streams1_nofused MM_CHOL_FUSED=0
