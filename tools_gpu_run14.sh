#!/bin/bash
# rehearsal of the N > 1 path on ONE GPU: 2 and 3 ranks sharing cuda:0, gloo collectives (RCCL needs one GPU per rank)
mkdir -p gpurun_out
make -C meatmodeler_amd/csrc -j8 > gpurun_out/make.log 2>&1 || { tail gpurun_out/make.log; exit 1; }
export MM_DIST_BACKEND=gloo
timeout -k 10 600 python bench.py --frames 60 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/dist_n1.log 2>&1
echo "n1 rc=$?"; tail -c 600 gpurun_out/dist_n1.log | head -c 600; echo
for n in 2 3; do
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus $n --frames 60 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/dist_n$n.log 2>&1
rc=$?; echo "n$n rc=$rc"; grep -E "Error|error|Traceback" gpurun_out/dist_n$n.log | head -5
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
done
python3 - <<PY
import json
for n in (1,2,3):
    l=[x for x in open('gpurun_out/dist_n%d.log'%n) if x.startswith('{')]
    if not l: print(n,'no json'); continue
    d=json.loads(l[-1]); p=d['problem']
    print(n, d['n_gpus'], round(d['ms_per_step'],1), d['stage_ms'], p['tracks'], p['observations'], p['ba_nfev'], p['ba_cost'])
PY
