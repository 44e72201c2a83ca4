#!/usr/bin/env python3
"""Split-f16 GEMM ops in a rocprofv3 kernel trace: an op is the operand-split kernel (h3_split_kernel) followed by its
GEMM (gemm_h3_kernel<EPI> / gemm_h3x_kernel<EPI> / gemm_h3m_kernel).  bench.py times the op with HIP events around the C-ABI call; this script
pairs the two dispatches in the trace so that the rocprof durations can be set beside that figure.

    python3 tools/h3_op_breakdown.py <dir with *_kernel_trace.csv> > profiles/<round>_h3_op_breakdown.json
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def main():
    files = glob.glob(os.path.join(sys.argv[1], '**', '*kernel_trace.csv'), recursive=True)
    if not files:
        raise SystemExit('no *kernel_trace.csv under %s' % sys.argv[1])
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), re.sub(r'\(.*$', '', r['Kernel_Name']).strip(),
                         int(r.get('Grid_Size_X') or r.get('Grid_Size') or 0)))
    rows.sort()
    ops = defaultdict(lambda: dict(n=0, gemm_ns=0, split_ns=0, gap_ns=0, with_split=0))
    for i, (s, e, name, grid) in enumerate(rows):
        if 'gemm_h3' not in name:
            continue
        key = '%s grid=%d' % (name, grid)
        o = ops[key]
        o['n'] += 1
        o['gemm_ns'] += e - s
        if i and rows[i - 1][2].startswith('h3_split_kernel'):
            ps, pe = rows[i - 1][0], rows[i - 1][1]
            o['split_ns'] += pe - ps
            o['gap_ns'] += max(0, s - pe)
            o['with_split'] += 1
    out = {}
    for k, o in sorted(ops.items()):
        n = o['n']
        out[k] = dict(launches=n, gemm_avg_us=round(o['gemm_ns'] / n / 1e3, 2),
                      split_avg_us=round(o['split_ns'] / n / 1e3, 2), gap_avg_us=round(o['gap_ns'] / n / 1e3, 2),
                      op_avg_us=round((o['gemm_ns'] + o['split_ns'] + o['gap_ns']) / n / 1e3, 2),
                      launches_preceded_by_split=o['with_split'])
    print(json.dumps(dict(_note=__doc__.strip().split('\n\n')[0].replace('\n', ' '), ops=out), indent=1))


if __name__ == '__main__':
    main()
