#!/bin/bash
# kernel trace of batched beam-5 searches (64 images, 20 steps forced)
set -o pipefail
R=$GRAFT_REPO_ROOT
TAG=${1:-r05_beam64}
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/profile_beam64.py 20 > $R/gpurun_out/${TAG}_plain.log 2>&1
tail -1 $R/gpurun_out/${TAG}_plain.log
ISC_BEAM_GRAPHS=0 python3 $R/tools/profile_beam64.py 20 > $R/gpurun_out/${TAG}_eager.log 2>&1
tail -1 $R/gpurun_out/${TAG}_eager.log
rm -rf $R/gpurun_out/prof_${TAG}
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG} -- python3 $R/tools/profile_beam64.py 10 > $R/gpurun_out/prof_${TAG}.log 2>&1
tail -1 $R/gpurun_out/prof_${TAG}.log
cp $(ls $R/gpurun_out/prof_${TAG}/*/*kernel_stats.csv | head -1) $R/gpurun_out/${TAG}_kernel_stats.csv
rm -rf $R/gpurun_out/prof_${TAG}
python3 - $R/gpurun_out/${TAG}_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print('total ms', sum(int(r['TotalDurationNs']) for r in rows) / 1e6)
for r in rows[:24]:
    print(r['Name'][:64].ljust(64), r['Calls'].rjust(5), '%8.0f us' % (int(r['TotalDurationNs']) / 1e3), '%7.1f' % (float(r['AverageNs']) / 1e3))
PY
