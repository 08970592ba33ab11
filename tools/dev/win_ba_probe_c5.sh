run() { tag=$1; shift; env "$@" timeout -k 10 400 python bench.py --frames 2000 --height 2160 --width 3840 --nfeatures 8000 --ba-window 50 --ba-stride 25 --steps 1 --warmup 1 --no-cpu-baseline $EXTRA > gpurun_out/winc5_$tag.log 2>&1; grep "^{" gpurun_out/winc5_$tag.log | tail -1 > gpurun_out/winc5_$tag.json; python -c "
import json,sys
j=json.load(open('gpurun_out/winc5_$tag.json')); s=j['sliding_window_ba']
print('$tag', round(s['ms'],1), s['nfev_total'], round(s['ms']/s['nfev_total'],4), s['streams'], j['stage_ms'])
"; }
run default A=1
run nofused MM_CHOL_FUSED=0
