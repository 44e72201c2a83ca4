# default bench line (all extras) -> gpurun_out/quick_bench.json, summary on stdout
set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 800 python bench.py "$@" > gpurun_out/quick_bench.json 2> gpurun_out/quick_bench.err || { tail -20 gpurun_out/quick_bench.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open('gpurun_out/quick_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['config']['batch_per_gpu'])
for k in d['roofline_kernels']:
    print(' ', k['kernel'], k['avg_us'], k['frac'], k.get('traffic'))
e = d.get('extra', {})
print({k: e[k] for k in ('batch_sweep',) if k in e})
for k in ('xe_train', 'xe_train_strong', 'rl_iteration'):
    if k in e: print(k, e[k]['ms_per_iter'])
if 'beam5' in e: print('beam', e['beam5'].get('per_image_p50_ms'))
if 'greedy_small_batches' in e: print(e['greedy_small_batches'])
PY
