"""Per-kernel register / LDS / scratch table of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage, compile only).

    python tools/kernel_resources.py insenticap_model_amd/csrc/rows.hip [name-filter]
"""
import re
import subprocess
import sys

FLAGS = ['--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17']


def table(src):
    err = subprocess.run(['/opt/rocm/bin/hipcc'] + FLAGS + ['-Rpass-analysis=kernel-resource-usage', '-c', src, '-o', '/dev/null'],
                         capture_output=True, text=True).stderr
    cur, rows = None, []
    for line in err.splitlines():
        m = re.search(r'remark: (.*?)\s*\[-Rpass', line)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith('Function Name:'):
            cur = {'name': t.split(':', 1)[1].strip()}
            rows.append(cur)
        elif cur is not None:
            k, _, v = t.partition(':')
            cur[k.strip()] = v.strip()
    names = subprocess.run(['c++filt'], input='\n'.join(r['name'] for r in rows), capture_output=True, text=True).stdout.split('\n')
    for r, n in zip(rows, names):
        r['name'] = n
    return rows


if __name__ == '__main__':
    flt = sys.argv[2] if len(sys.argv) > 2 else ''
    for r in table(sys.argv[1]):
        if flt in r['name']:
            print('%-60s vgpr %4s agpr %3s sgpr %3s scratch %4s occ %2s lds %6s' % (
                r['name'][:60], r.get('VGPRs'), r.get('AGPRs'), r.get('TotalSGPRs'), r.get('ScratchSize [bytes/lane]'),
                r.get('Occupancy [waves/SIMD]'), r.get('LDS Size [bytes/block]')))
