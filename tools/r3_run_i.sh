cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
python bench.py --gpus 2 > gpurun_out/i_gpus2.log 2>&1; echo "gpus2 rc=$?"; tail -1 gpurun_out/i_gpus2.log
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29613 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/i_torchrun1.json 2> gpurun_out/i_torchrun1.err
echo "torchrun rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/i_torchrun1.json'))
print(d['value'], d['n_gpus'], d['scaling'], d['config']['ranks'], d['config']['collective_backend'])
x=d['extra']
for k in ('xe_train','xe_train_strong','grad_allreduce','rl_iteration'):
    print(k, json.dumps(x.get(k))[:400])
PY
