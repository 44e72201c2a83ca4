#!/usr/bin/env python3
"""Per-iteration picture of a kernel trace of graph-served XE iterations: wall between optimizer launches, busy time per
queue, the longest kernels.  python tools/xe_graph_trace_summary.py <dir with *_kernel_trace.csv>"""
import csv, glob, sys, collections
f = max(glob.glob(sys.argv[1] + '/*/*kernel_trace.csv'), key=lambda p: len(open(p).read()))
rows = list(csv.DictReader(open(f)))
name = lambda r: r['Kernel_Name']
adam = [i for i, r in enumerate(rows) if name(r).startswith('clamp_adam_kernel')]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
adam = [i for i, r in enumerate(rows) if name(r).startswith('clamp_adam_kernel')]
# the last 4 iterations: replays
seg = rows[adam[-5] + 1: adam[-1] + 1]
n = 4
t0, t1 = int(seg[0]['Start_Timestamp']), int(seg[-1]['End_Timestamp'])
print('launches/iter', len(seg) / n, 'wall/iter ms', (t1 - t0) / n / 1e6)
busy = collections.Counter(); cnt = collections.Counter()
qkey = 'Queue_Id' if 'Queue_Id' in seg[0] else 'Stream_Id'
for r in seg:
    busy[r[qkey]] += int(r['End_Timestamp']) - int(r['Start_Timestamp']); cnt[r[qkey]] += 1
for q in busy: print('queue', q, 'launches/iter', cnt[q] / n, 'busy ms/iter', busy[q] / n / 1e6)
per = collections.defaultdict(lambda: [0, 0])
for r in seg:
    k = (r[qkey], name(r)[:60]); per[k][0] += 1; per[k][1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
for (q, k), (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1])[:34]:
    print(q, k.ljust(60), '%6.1f' % (c / n), '%7.0f us/iter' % (t / n / 1e3), '%6.1f' % (t / c / 1e3))

# the last iteration launch by launch: offset from its first launch, duration, gap to the previous end, grid, kernel
if len(sys.argv) > 2:
    it = rows[adam[-2] + 1: adam[-1] + 1]
    t_first = int(it[0]['Start_Timestamp'])
    prev_end = t_first
    with open(sys.argv[2], 'w') as fo:
        for i, r in enumerate(it):
            st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
            grid = r.get('Grid_Size', r.get('Grid_Size_X', '?'))
            wg = r.get('Workgroup_Size', r.get('Workgroup_Size_X', '?'))
            fo.write('%4d %9.1f %8.1f %7.1f %9s %5s %s\n' % (i, (st - t_first) / 1e3, (en - st) / 1e3, (st - prev_end) / 1e3,
                                                          grid, wg, name(r)[:70]))
            prev_end = max(prev_end, en)
