set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_train_graph.py tests/test_gpu_dp.py -x -q -m gpu > gpurun_out/graph_tests.log 2>&1 || { tail -60 gpurun_out/graph_tests.log; exit 1; }
tail -3 gpurun_out/graph_tests.log
bash tools/r3_quick_bench.sh --no-cpu-baseline --no-kernel-timing --steps 4
python - <<'PY'
import json
d = json.loads(open('gpurun_out/quick_bench.json').read().strip().splitlines()[-1])
for k in ('xe_train', 'xe_train_strong', 'xe_train_by_batch'): print(k, d['extra'][k])
PY
