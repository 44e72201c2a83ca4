#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 PMC passes of bench.py.

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d <fetch_dir> -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d <write_dir> -- python3 bench.py ... (same command)
    python3 tools/pmc_summary.py <fetch_dir> <write_dir> <batch_per_gpu> [<mfma_dir> <clk_dir> <stats_dir>] > profiles/<round>_pmc_summary_B<batch>.json

Optional matrix-pipe passes (tools/profile_round.sh): <mfma_dir> holds SQ_VALU_MFMA_BUSY_CYCLES (cycles in which a SIMD's
matrix pipe executes: 32 per v_mfma_f32_32x32x16_f16, 64 per v_mfma_f32_32x32x2_f32), SQ_INSTS_VALU_MFMA_MOPS_F16 / _F32,
SQ_WAVE_CYCLES / SQ_WAIT_* (quad-cycles); <clk_dir> GRBM_GUI_ACTIVE (summed over the 8 XCDs: / 8 = the kernel's own
cycles, MI355X_MICROARCH.md "DVFS give-back") and the L2's TCC_HIT / TCC_MISS; <stats_dir> the kernel-trace stats
(average duration).  Derived per kernel: mfma_util = MFMA_BUSY / (1024 SIMDs x GUI_ACTIVE / 8) - the share of the
kernel's cycles, at the clock the chip actually held, in which a matrix pipe was executing; clock_ghz = GUI_ACTIVE / 8 /
duration; l2_hit = HIT / (HIT + MISS).

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


def durations(stats_dir):
    out = {}
    for f in glob.glob(os.path.join(stats_dir, '**', '*kernel_stats.csv'), recursive=True):
        for row in csv.DictReader(open(f)):
            out[re.sub(r'\(.*$', '', row['Name']).strip()] = float(row['AverageNs'])
    return out


def main():
    fetch_dir, write_dir, batch = sys.argv[1], sys.argv[2], int(sys.argv[3])
    fetch, write = per_kernel(fetch_dir, 'FETCH_SIZE'), per_kernel(write_dir, 'WRITE_SIZE')
    extra = {}
    if len(sys.argv) >= 7:
        mfma_dir, clk_dir, stats_dir = sys.argv[4:7]
        for d, names in ((mfma_dir, ('SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_BUSY_CYCLES', 'SQ_WAVE_CYCLES', 'SQ_WAIT_INST_ANY',
                                     'SQ_WAIT_ANY', 'SQ_INSTS_VALU_MFMA_MOPS_F16', 'SQ_INSTS_VALU_MFMA_MOPS_F32')),
                         (clk_dir, ('GRBM_GUI_ACTIVE', 'TCC_HIT_sum', 'TCC_MISS_sum'))):
            for c in names:
                try:
                    extra[c] = per_kernel(d, c)
                except SystemExit:
                    extra[c] = {}
        extra['_dur'] = durations(stats_dir)
    kernels = {}
    for n, (f_kb, launches) in sorted(fetch.items()):
        if n.startswith('void at::') or 'elementwise' in n or n.startswith('__amd'):
            continue                                              # torch's own fill / copy kernels
        w_kb = write.get(n, (0.0, 0))[0]
        kernels[n] = dict(launches=launches, fetch_size_kb_raw=round(f_kb, 1), write_size_kb=round(w_kb, 1),
                          traffic_bytes=int(2 * f_kb * 1024 + w_kb * 1024))
        if extra:
            k = kernels[n]
            g = lambda c: extra.get(c, {}).get(n, (None, 0))[0]
            for c, key in (('SQ_VALU_MFMA_BUSY_CYCLES', 'mfma_busy_cycles'), ('SQ_BUSY_CYCLES', 'sq_busy_cycles'),
                           ('SQ_WAVE_CYCLES', 'wave_quad_cycles'), ('SQ_WAIT_INST_ANY', 'wait_inst_quad_cycles'),
                           ('SQ_WAIT_ANY', 'wait_any_quad_cycles'), ('SQ_INSTS_VALU_MFMA_MOPS_F16', 'mfma_mops_f16'),
                           ('SQ_INSTS_VALU_MFMA_MOPS_F32', 'mfma_mops_f32'), ('GRBM_GUI_ACTIVE', 'gui_active_sum_xcd')):
                v = g(c)
                if v is not None:
                    k[key] = int(v)
            dur = extra['_dur'].get(n)
            if dur:
                k['avg_us'] = round(dur / 1e3, 2)
            gui = g('GRBM_GUI_ACTIVE')
            if gui and k.get('mfma_busy_cycles') is not None:
                k['mfma_util'] = round(k['mfma_busy_cycles'] / (1024.0 * gui / 8.0), 4)
            if gui and dur:
                k['clock_ghz'] = round(gui / 8.0 / dur, 3)
            hit, miss = g('TCC_HIT_sum'), g('TCC_MISS_sum')
            if hit is not None and miss is not None and hit + miss > 0:
                k['l2_hit'] = round(hit / (hit + miss), 4)
    print(json.dumps(dict(_note=__doc__.strip().split('\n\n')[-1].replace('\n', ' '), kernels=kernels,
                          batch_per_gpu=batch), indent=1))


if __name__ == '__main__':
    main()
