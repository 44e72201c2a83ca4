cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_determinism.py -q -m gpu -x > gpurun_out/k_tests.log 2>&1
echo "first rc=$?"; tail -4 gpurun_out/k_tests.log
timeout -k 10 300 python - <<'PY'
import bench, torch, json
bench.load_product()
from insenticap_model_amd import Captioner, synth
dev=torch.device('cuda:0')
cap = Captioner(synth.make_idx2word(bench.V), synth.SENTIMENT_CATEGORIES, synth.DEFAULT_SETTINGS)
cap.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_weights(bench.V, synth.DEFAULT_SETTINGS).items()})
cap.to(dev).eval()
print(json.dumps(bench.bench_small_batches(cap, dev)))
inputs,_=bench.device_inputs(64, 100, dev)
b=bench.bench_beam(cap, inputs)
print({k:b[k] for k in ('per_image_p50_ms','per_step_p50_us','full_search_p50_ms')}, b['eager'])
PY
