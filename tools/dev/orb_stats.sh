#!/bin/bash
# per-kernel time of ORB detect + describe on the bench clip (tools/bench_orb.py under rocprofv3 --stats)
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/orbstats
ORB_REPS=3 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/orbstats -- python3 $ROOT/tools/bench_orb.py > /tmp/orbstats.log 2>&1
tail -1 /tmp/orbstats.log
python3 -c "
import csv,glob
for f in glob.glob('/tmp/orbstats/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'orb_' in r['Name']: print('%-22s calls %4s  total %8.3f ms  per clip %7.3f ms' % (r['Name'].split('orb_')[1].split('(')[0], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['TotalDurationNs'])/3e6))
"
