# per-launch durations of isc_beam_select: back to back vs between other kernels (tools/select_lab.py)
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_select_lab
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_select_lab -- python3 $R/tools/select_lab.py > $R/gpurun_out/prof_select_lab.log 2>&1
tail -1 $R/gpurun_out/prof_select_lab.log
python3 - $R/gpurun_out/prof_select_lab <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = [r for r in csv.DictReader(open(f)) if 'beam_select' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows]
print('warm-up      ', ' '.join('%.1f' % x for x in d[:3]))
print('back to back ', ' '.join('%.1f' % x for x in d[3:23]))
print('between      ', ' '.join('%.1f' % x for x in d[23:43]))
print('returns at once, between', ' '.join('%.1f' % x for x in d[43:]))
PY
