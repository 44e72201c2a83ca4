set -x
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests/ -q -m gpu > gpurun_out/d_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/d_tests.log
tail -15 gpurun_out/d_tests.log
timeout -k 10 600 python bench.py --steps 10 --warmup 3 > gpurun_out/d_bench.json 2> gpurun_out/d_bench.err
echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/d_bench.json'))
print(d['value'], d['ms_per_step'])
for e in [d['roofline']]+d['roofline_kernels']:
    print('%-44s %8.1f us  frac %.3f  %s' % (e['kernel'][:44], e['avg_us'], e['frac'], e['phase']))
x=d['extra']
print({k: x[k] for k in ('greedy_small_batches','xe_train','xe_train_strong','xe_train_by_batch') if k in x})
print(x.get('beam5'))
print(x.get('rl_iteration'))
PY
