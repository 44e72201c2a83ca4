#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 PMC passes of bench.py.

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d <fetch_dir> -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d <write_dir> -- python3 bench.py ... (same command)
    python3 tools/pmc_summary.py <fetch_dir> <write_dir> <batch_per_gpu> > profiles/<round>_pmc_summary_B<batch>.json

Values are KB per launch averaged over the launches of a kernel symbol.  traffic_bytes = 2 * FETCH_SIZE * 1024
+ WRITE_SIZE * 1024: gfx950 tallies the 128-B read requests of 16-B-per-lane loads at 64 B
(MI355X_MICROARCH.md, HBM section) - the attention scan, whose algorithmic bytes are known, calibrates it.
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def per_kernel(directory, counter):
    files = glob.glob(os.path.join(directory, '**', '*counter_collection.csv'), recursive=True)
    if not files:
        raise SystemExit('no *counter_collection.csv under %s' % directory)
    per_dispatch = defaultdict(float)
    names = {}
    for f in files:
        for row in csv.DictReader(open(f)):
            if row.get('Counter_Name') != counter:
                continue
            key = (f, row.get('Dispatch_Id') or row.get('Correlation_Id'))
            per_dispatch[key] += float(row['Counter_Value'])      # rows split by XCD / instance: sum them
            names[key] = row['Kernel_Name']
    tot, cnt = defaultdict(float), defaultdict(int)
    for key, v in per_dispatch.items():
        n = re.sub(r'\(.*$', '', names[key]).strip()              # drop the argument list
        tot[n] += v
        cnt[n] += 1
    return {n: (tot[n] / cnt[n], cnt[n]) for n in tot}


def main():
    fetch_dir, write_dir, batch = sys.argv[1], sys.argv[2], int(sys.argv[3])
    fetch, write = per_kernel(fetch_dir, 'FETCH_SIZE'), per_kernel(write_dir, 'WRITE_SIZE')
    kernels = {}
    for n, (f_kb, launches) in sorted(fetch.items()):
        if n.startswith('void at::') or 'elementwise' in n or n.startswith('__amd'):
            continue                                              # torch's own fill / copy kernels
        w_kb = write.get(n, (0.0, 0))[0]
        kernels[n] = dict(launches=launches, fetch_size_kb_raw=round(f_kb, 1), write_size_kb=round(w_kb, 1),
                          traffic_bytes=int(2 * f_kb * 1024 + w_kb * 1024))
    print(json.dumps(dict(_note=__doc__.strip().split('\n\n')[-1].replace('\n', ' '), kernels=kernels,
                          batch_per_gpu=batch), indent=1))


if __name__ == '__main__':
    main()
