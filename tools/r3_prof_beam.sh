set -e
R=$GRAFT_REPO_ROOT
TAG=${1:-r03_h}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_${TAG}_beam
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_beam -- python3 $R/tools/profile_beam.py 10 > $R/gpurun_out/prof_${TAG}_beam.log 2>&1
tail -1 $R/gpurun_out/prof_${TAG}_beam.log
python3 - $R/gpurun_out/prof_${TAG}_beam <<'PY'
import csv, glob, sys
f = max(glob.glob(sys.argv[1] + '/*/*kernel_stats.csv'), key=lambda p: len(open(p).read()))
rows = list(csv.DictReader(open(f)))
print('total ms', sum(int(r['TotalDurationNs']) for r in rows) / 1e6)
for r in rows[:14]:
    print(r['Name'][:64].ljust(64), r['Calls'].rjust(5), '%8.0f us' % (int(r['TotalDurationNs']) / 1e3), '%7.1f' % (float(r['AverageNs']) / 1e3))
PY
