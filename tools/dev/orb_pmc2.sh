#!/bin/bash
# more SQ counters of the ORB kernels (wait / issue breakdown), per kernel
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export ORB_F=64 ORB_REPS=2
for grp in "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INSTS_SMEM SQ_INSTS_VMEM_WR" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_LDS_ATOMIC_RETURN SQ_IFETCH"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rm -rf /tmp/orbpmc2_$tag
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/orbpmc2_$tag -- python3 $ROOT/tools/bench_orb.py > $ROOT/gpurun_out/orbpmc2_$tag.log 2>&1 || echo "pass $tag failed"
done
python3 - <<PY > $ROOT/gpurun_out/orb_pmc2.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
dur = collections.defaultdict(float)
first = True
for d in sorted(glob.glob("/tmp/orbpmc2_*")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if "orb_" not in n: continue
            n = n.split("orb_")[1].split("(")[0]
            acc[n][r["Counter_Name"]] += float(r["Counter_Value"])
    if first:
        for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                n = r["Kernel_Name"]
                if "orb_" not in n: continue
                n = n.split("orb_")[1].split("(")[0]
                calls[n] += 1
                dur[n] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        first = False
for n, c in sorted(acc.items(), key=lambda kv: -dur[kv[0]]):
    print(n, "launches", calls[n], "total_ms(under pmc)", round(dur[n] / 1e6, 3))
    for k, v in sorted(c.items()):
        print("    %-24s %.4g" % (k, v))
PY
grep -A26 "^pyramid_kernel" $ROOT/gpurun_out/orb_pmc2.txt
