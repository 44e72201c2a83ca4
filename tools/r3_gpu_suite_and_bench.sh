set -x
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests/ -q -m gpu > gpurun_out/suite_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/suite_tests.log
tail -8 gpurun_out/suite_tests.log
timeout -k 10 600 python bench.py > gpurun_out/suite_bench.json 2> gpurun_out/suite_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/suite_bench.json'))
print(d['value'], d['ms_per_step'])
for e in [d['roofline']]+d['roofline_kernels'][:8]:
    print('%-44s %8.1f us  frac %.3f  %s  %s' % (e['kernel'][:44], e['avg_us'], e['frac'], e['per_rollout_ms'], e['per_rollout_ms_events']))
x=d['extra']
print(x['beam5'])
print(x['greedy_small_batches'])
print(d['cpu_baseline'])
PY
