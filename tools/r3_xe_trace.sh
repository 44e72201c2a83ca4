# kernel-trace statistics of the XE iteration at B=128 only, summarised per iteration
set -e
R=$GRAFT_REPO_ROOT
TAG=${1:-r03_f}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_${TAG}_xe
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_xe -- python3 $R/tools/profile_xe.py 6 128 > $R/gpurun_out/prof_${TAG}_xe.log 2>&1
tail -1 $R/gpurun_out/prof_${TAG}_xe.log
python3 - $R/gpurun_out/prof_${TAG}_xe <<'PY'
import csv, glob, sys
f = max(glob.glob(sys.argv[1] + '/*/*kernel_stats.csv'), key=lambda p: len(open(p).read()))
rows = list(csv.DictReader(open(f)))
it = [int(r['Calls']) for r in rows if r['Name'].startswith('clamp_adam_kernel')][0]
print('iterations', it, 'launches/iter', sum(int(r['Calls']) for r in rows) / it, 'busy ms/iter', sum(int(r['TotalDurationNs']) for r in rows) / it / 1e6)
for r in rows[:28]:
    print(r['Name'][:64].ljust(64), '%7.1f' % (int(r['Calls']) / it), '%8.0f us/iter' % (int(r['TotalDurationNs']) / it / 1e3), '%7.1f' % (float(r['AverageNs']) / 1e3))
PY
