#!/bin/bash
# SQ counters of the ORB kernels (one pass per counter group), per kernel: instructions per pixel and busy fractions
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export ORB_F=64 ORB_REPS=2
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rm -rf /tmp/orbpmc_$tag
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/orbpmc_$tag -- python3 $ROOT/tools/bench_orb.py > $ROOT/gpurun_out/orbpmc_$tag.log 2>&1 || exit 1
done
python3 - <<PY > $ROOT/gpurun_out/orb_pmc.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
dur = collections.defaultdict(float)
for d in glob.glob("/tmp/orbpmc_*"):
    seen = set()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if "orb_" not in n: continue
            n = n.split("orb_")[1].split("(")[0]
            acc[n][r["Counter_Name"]] += float(r["Counter_Value"])
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        if "SQ_INSTS_VALU" not in d: continue
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if "orb_" not in n: continue
            n = n.split("orb_")[1].split("(")[0]
            calls[n] += 1
            dur[n] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for n, c in sorted(acc.items(), key=lambda kv: -dur[kv[0]]):
    print(n, "launches", calls[n], "total_ms(under pmc)", round(dur[n] / 1e6, 3))
    for k, v in sorted(c.items()):
        print("    %-24s %.4g" % (k, v))
PY
cat $ROOT/gpurun_out/orb_pmc.txt
