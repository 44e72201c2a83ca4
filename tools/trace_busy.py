#!/usr/bin/env python3
"""Device occupancy picture of a kernel trace between optimizer launches: per iteration the wall time, the union of kernel
intervals (device busy), per-queue busy time and the idle gaps longer than 50 us (offset, length, the kernels on either side).
    python tools/trace_busy.py <dir with *_kernel_trace.csv> [iterations from the end, default 3]"""
import csv, glob, sys, collections
f = max(glob.glob(sys.argv[1] + '/*/*kernel_trace.csv'), key=lambda p: len(open(p).read()))
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
name = lambda r: r['Kernel_Name']
adam = [i for i, r in enumerate(rows) if name(r).startswith('clamp_adam_kernel')]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
qkey = 'Queue_Id' if 'Queue_Id' in rows[0] else 'Stream_Id'
for k in range(-n, 0):
    seg = rows[adam[k - 1] + 1: adam[k] + 1]
    t0, t1 = int(seg[0]['Start_Timestamp']), int(seg[-1]['End_Timestamp'])
    iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), name(r)[:48]) for r in seg)
    busy, end, gaps, last = 0, t0, [], iv[0][2]
    for s, e, nm in iv:
        if s > end:
            if s - end > 50000:
                gaps.append((end - t0, s - end, last, nm))
            busy += e - s
            end, last = e, nm
        elif e > end:
            busy += e - end
            end, last = e, nm
    per = collections.Counter()
    for r in seg:
        per[r[qkey]] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    print('iteration %d: wall %.2f ms, device busy %.2f ms, launches %d, per queue %s' % (
        k, (t1 - t0) / 1e6, busy / 1e6, len(seg), {q: round(v / 1e6, 2) for q, v in per.items()}))
    for off, ln, a, b in gaps:
        print('    idle %7.1f us at %8.1f us   after %-48s before %s' % (ln / 1e3, off / 1e3, a, b))
    # per queue: runs of launches separated by > 200 us (offset of the run, its length, launches, busy inside, first kernel)
    byq = collections.defaultdict(list)
    for r in seg:
        byq[r[qkey]].append((int(r['Start_Timestamp']), int(r['End_Timestamp']), name(r)[:40]))
    for q, lst in sorted(byq.items()):
        lst.sort()
        run = [lst[0]]
        def flush(run):
            print('    queue %s run at %8.1f us  length %8.1f us  launches %4d  busy %8.1f us  %s ... %s' % (
                q, (run[0][0] - t0) / 1e3, (run[-1][1] - run[0][0]) / 1e3, len(run), sum(e - s for s, e, _ in run) / 1e3,
                run[0][2], run[-1][2]))
        for x in lst[1:]:
            if x[0] - run[-1][1] > 200000:
                flush(run); run = []
            run.append(x)
        flush(run)
