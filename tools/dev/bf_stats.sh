#!/bin/bash
# per-kernel time of the matching stage alone (tools/bench_bf.py, MFMA variant) under rocprofv3 --stats
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/bfstats
BF_ONLY=${BF_ONLY:-200} timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bfstats -- python3 $ROOT/tools/bench_bf.py > /tmp/bfstats.log 2>&1
tail -2 /tmp/bfstats.log
python3 -c "
import csv,glob
for f in glob.glob('/tmp/bfstats/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'bf_' in r['Name']: print('%-28s calls %4s  avg %9.1f us' % (r['Name'].split('bf_')[1].split('(')[0][:28], r['Calls'], float(r['AverageNs'])/1e3))
"
