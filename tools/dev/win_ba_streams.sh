for s in 8 12 16; do
timeout -k 10 300 python bench.py --frames 400 --nfeatures 8000 --ba-window 50 --ba-stride 25 --steps 1 --warmup 1 --no-cpu-baseline --ba-streams $s > gpurun_out/win_s$s.log 2>&1; grep "^{" gpurun_out/win_s$s.log | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); s=j['sliding_window_ba']
print('streams', s['streams'], round(s['ms'],1), s['nfev_total'], round(s['ms']/s['nfev_total'],4))
"
done
