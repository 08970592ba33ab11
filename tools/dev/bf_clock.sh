#!/bin/bash
# effective shader clock during the BF kernel: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / kernel duration
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/bfclk
BF_ONLY=1 timeout -k 10 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS --output-format csv -d $ROOT/gpurun_out/bfclk -- python3 $ROOT/tools/bench_bf.py > $ROOT/gpurun_out/bfclk.log 2>&1
echo rc=$?
python3 - <<PY
import csv, glob, collections
rows = []
for f in glob.glob("$ROOT/gpurun_out/bfclk/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
tr = {}
for f in glob.glob("$ROOT/gpurun_out/bfclk/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        tr[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in rows:
    if "bf_knn2" not in r["Kernel_Name"]: continue
    acc[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
for did, c in list(acc.items())[-4:]:
    dur = tr.get(did, (0, ""))[0]
    print(did, "dur_us", dur / 1e3, {k: v for k, v in c.items()}, "clock GHz", c.get("GRBM_GUI_ACTIVE", 0) / 8 / max(dur, 1))
PY
find $ROOT/gpurun_out/bfclk -name "*.csv" -size +1M -delete
