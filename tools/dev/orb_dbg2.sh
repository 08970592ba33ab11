#!/bin/bash
# describe kernel time with stages cut off (MM_ORB_DBG: 32 after the patch load, 64 after moments, 128 after hblur, 256 after vblur)
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export ORB_F=64 ORB_REPS=2
for dbg in 0 256 128 64 32; do
  rm -rf /tmp/orbdbg
  MM_ORB_DBG=$dbg timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/orbdbg -- python3 $ROOT/tools/bench_orb.py > /tmp/orbdbg.log 2>&1
  python3 -c "
import csv,glob
for f in glob.glob('/tmp/orbdbg/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'orb_describe' in r['Name']: print('dbg=$dbg', r['Calls'], 'avg_us', float(r['AverageNs'])/1e3)
"
done
