set -e
cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing > gpurun_out/torchrun1.json 2> gpurun_out/torchrun1.err || { tail -30 gpurun_out/torchrun1.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open('gpurun_out/torchrun1.json').read().strip().splitlines()[-1])
print(d['value'], d['n_gpus'], d['config'].get('collective_backend'))
e = d['extra']
for k in ('xe_train', 'xe_train_strong', 'grad_allreduce', 'rl_iteration'):
    print(k, e.get(k))
PY
