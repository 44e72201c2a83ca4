#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5rl
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for P in 0 1; do ISC_PAIR_UNROLLS=$P timeout -k 10 300 python3 tools/profile_rl.py 6 2>&1 | grep -o "'ms_per_iter': [0-9.]*" | sed "s/^/pair=$P /"; done
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rl -- python3 tools/profile_rl.py 3 > $OUT/rl.log 2>&1; echo "prof rc=$?"
cp $(ls $OUT/rl/*/*kernel_stats.csv | head -1) $OUT/kernel_stats_rl_B512.csv
rm -rf $OUT/rl
python3 - <<'PY'
import csv,os
rows=list(csv.DictReader(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r5rl/kernel_stats_rl_B512.csv')))
tot=sum(int(r['TotalDurationNs']) for r in rows)
print('total kernel ms over the run', tot/1e6)
for r in rows[:28]:
    print('%7d %9.1f us %8.1f  %s'%(int(r['Calls']), int(r['TotalDurationNs'])/1e3, float(r['AverageNs'])/1e3, r['Name'][:90]))
PY
