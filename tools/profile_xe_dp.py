#!/usr/bin/env python3
"""The eager merged XE step under a ONE-RANK RCCL group (the data-parallel code path on a one-GPU box): ms per iteration
with the bucketed exchange, with the buckets' collectives skipped, and with one flat all-reduce; under rocprofv3
--kernel-trace the last iterations are the bucketed form.
    python tools/profile_xe_dp.py [iterations [batch]] [--json]
--json: the last line of stdout is one JSON object (bench.py's `xe_exchange_one_rank` job reads it)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29655')
import torch
import torch.distributed as dist
import bench
bench.load_product()
from insenticap_model_amd import Captioner, synth, dp
from insenticap_model_amd.train import xe_train_step

dev = torch.device('cuda:0')
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
as_json = '--json' in sys.argv
argv = [a for a in sys.argv[1:] if a != '--json']
iters = int(argv[0]) if len(argv) > 0 else 20
B = int(argv[1]) if len(argv) > 1 else 128
V, R, T = bench.V, bench.R, bench.T
cap = Captioner(synth.make_idx2word(V), synth.SENTIMENT_CATEGORIES, synth.DEFAULT_SETTINGS)
cap.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_weights(V, synth.DEFAULT_SETTINGS).items()})
cap.to(dev).train()
cap.pair_unrolls = True
optim, xc, dc = cap.get_optim_criterion(4e-4)
arena = dp.GradArena(cap.parameters())
d = synth.make_inputs(B, V, synth.DEFAULT_SETTINGS, regions=R, seq_len=T, seed=500)
s = synth.make_inputs(80, V, synth.DEFAULT_SETTINGS, regions=R, seq_len=T, seed=600)
tt = lambda x: torch.from_numpy(x).to(dev)
fact = (None, tt(d['fc_feats']), tt(d['att_feats']), (tt(d['captions']), d['lengths']), tt(d['cpt_words']))
scs = ((tt(s['captions']), s['lengths']), tt(s['cpt_words']), tt(s['senti_words']), tt(s['senti_labels']))
labels = tt(d['senti_labels'])


def run(n, **kw):
    for _ in range(3):
        xe_train_step(cap, optim, xc, dc, fact, labels, scs, 0.0, 0.1, arena=arena, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        xe_train_step(cap, optim, xc, dc, fact, labels, scs, 0.0, 0.1, arena=arena, **kw)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


flat = run(iters, bucketed=False)
buck = run(iters)
sink = cap.__dict__['_dp_sink']
sink.exchange = False
dry = run(iters)
sink.exchange = True
buck2 = run(iters)
print('ms per iteration: flat %.3f  bucketed %.3f / %.3f  buckets without their collectives %.3f' % (flat, buck, buck2, dry))
if as_json:
    best = min(buck, buck2)
    print(json.dumps(dict(ranks=1, backend=dist.get_backend(), batch_per_gpu=B, seq2seq_rows_per_gpu=80, iters=iters,
                          bucketed_ms_per_iter=round(best, 3), flat_ms_per_iter=round(flat, 3),
                          no_exchange_ms_per_iter=round(dry, 3), exposed_allreduce_ms=round(best - dry, 3),
                          exposed_allreduce_flat_ms=round(flat - dry, 3), buckets=len(sink.buckets) if hasattr(sink, 'buckets') else None,
                          all_reduce_mb=round(arena.nbytes / 1e6, 2))))
dist.destroy_process_group()
