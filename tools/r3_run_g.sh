cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
for thr in 224 160; do
  echo "== ISC_LAB_H3X_MIN_TILES=$thr"
  ISC_LAB_H3X_MIN_TILES=$thr timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline > gpurun_out/g_bench_$thr.json 2> gpurun_out/g_bench_$thr.err
  python - <<PY
import json
d=json.load(open('gpurun_out/g_bench_$thr.json'))
print(d['value'], d['ms_per_step'])
for e in [d['roofline']]+d['roofline_kernels'][:12]:
    print('%-44s %8.1f us  frac %.3f' % (e['kernel'][:44], e['avg_us'], e['frac']))
PY
done
timeout -k 10 600 python -m pytest tests/test_gpu_h3.py tests/test_gpu_parity.py tests/test_gpu_bench_config.py -q -m gpu 2>&1 | tail -3
