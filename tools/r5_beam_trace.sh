#!/bin/bash
# launch-by-launch listing of ONE replayed single-image beam-5 search (prologue + the first steps): start offset, duration,
# gap to the previous kernel's end, name
set -o pipefail
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_beamtrace
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_beamtrace -- python3 $R/tools/profile_beam.py 6 > $R/gpurun_out/prof_beamtrace.log 2>&1
tail -1 $R/gpurun_out/prof_beamtrace.log
python3 - $(ls $R/gpurun_out/prof_beamtrace/*/*kernel_trace.csv | head -1) <<'PY' > $R/gpurun_out/r05_beam_single_image_trace.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# searches start with the input staging launch (copy_multi_kernel); take the last complete one
starts = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('copy_multi_kernel')]
a, b = starts[-2], starts[-1]
t0 = int(rows[a]['Start_Timestamp'])
prev_end = t0
for i in range(a, b):
    r = rows[i]
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print('%4d %9.1f %7.1f %7.1f  %s' % (i - a, (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r['Kernel_Name'][:90]))
    prev_end = e
print('search: %d launches, %.1f us from the first start to the last end' % (b - a, (prev_end - t0) / 1e3))
PY
head -45 $R/gpurun_out/r05_beam_single_image_trace.txt
tail -3 $R/gpurun_out/r05_beam_single_image_trace.txt
rm -rf $R/gpurun_out/prof_beamtrace
