#!/usr/bin/env python3
"""Soak 3, under a ONE-RANK RCCL group (the data-parallel code paths on a one-GPU box): eager bucketed XE steps, graph-served XE
steps with the flat exchange, the RL iteration with Detector.enable_data_parallel - hundreds of each in one process.
    python tools/soak_dp.py [iterations]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29657')
import torch
import torch.distributed as dist
import bench
bench.load_product()
from insenticap_model_amd import Detector, synth, dp, ops
from insenticap_model_amd.train import xe_train_step
from insenticap_model_amd.train_graph import XETrainGraph

dev = torch.device('cuda:0')
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
V, T, R = bench.V, bench.T, bench.R
st = dict(synth.DEFAULT_SETTINGS, **synth.HELPER_SETTINGS)
det = Detector(synth.make_idx2word(V), T, synth.SENTIMENT_CATEGORIES, {'cap_lr': 4e-5}, st)
det.captioner.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_weights(V, st, seed=0).items()})
det.to(dev)
det.enable_data_parallel()
cap, arena = det.captioner, det.dp_arena
tt = torch.from_numpy
B = 128
d = synth.make_inputs(B, V, st, regions=R, seq_len=T, seed=500)
s = synth.make_inputs(80, V, st, regions=R, seq_len=T, seed=600)
fact = (None, tt(d['fc_feats']).to(dev), tt(d['att_feats']).to(dev), (tt(d['captions']).to(dev), d['lengths']), tt(d['cpt_words']).to(dev))
scs = ((tt(s['captions']).to(dev), s['lengths']), tt(s['cpt_words']).to(dev), tt(s['senti_words']).to(dev), tt(s['senti_labels']).to(dev))
labels = tt(d['senti_labels']).to(dev)
c0 = dp.COLLECTIVES
t0 = time.perf_counter()
for i in range(0 if os.environ.get('SKIP_EAGER') else n):
    l = xe_train_step(cap, det.cap_optim, det.cap_xe_crit, det.cap_da_crit, fact, labels, scs, 0.25, 0.1, arena=arena)
torch.cuda.synchronize()
if not os.environ.get('SKIP_EAGER'): print('eager bucketed: %d steps, %.2f ms each, %d collectives, loss %.4f' % (n, (time.perf_counter() - t0) / n * 1e3, dp.COLLECTIVES - c0, float(l['all_loss'])), flush=True)
cap.cpt_feats = cap.fc_feats = None
g = XETrainGraph(cap, det.cap_optim, det.cap_xe_crit, det.cap_da_crit, grad_clip=0.1, arena=arena, warmup=2)
t0 = time.perf_counter()
for i in range(0 if os.environ.get('SKIP_GRAPH') else n):
    l = g.step(fact, labels, scs, 0.25)
torch.cuda.synchronize()
if not os.environ.get('SKIP_GRAPH'): print('graph + flat exchange: %d steps, %.2f ms each, %d replays, loss %.4f' % (n, (time.perf_counter() - t0) / n * 1e3, g.replays, float(l['all_loss'])), flush=True)
batches, split = synth.make_rl_batches(2, 64, V, st, grid=(6, 6), seq_len=T, seed=90)
det.set_ciderd_scorer(split)
facts = [[(b[0], tt(b[1]).to(dev), tt(b[2]).to(dev), (tt(b[3][0]).to(dev), b[3][1]), tt(b[4]).to(dev), tt(b[5]).to(dev), b[6])] for b in batches]
t0 = time.perf_counter()
m = max(20, n // 3)
for i in range(m):
    out = det((facts[i % 2], [scs]), 'fact', True)
    assert all(v == v for v in out.values()), out
torch.cuda.synchronize()
print('RL under the group: %d iterations, %.2f ms each, %d replays' % (m, (time.perf_counter() - t0) / m * 1e3, det._rl_graph.replays), flush=True)
mem = torch.cuda.memory_allocated() / 1e6
for i in range(30):
    xe_train_step(cap, det.cap_optim, det.cap_xe_crit, det.cap_da_crit, fact, labels, scs, 0.0, 0.1, arena=arena)
    det((facts[i % 2], [scs]), 'fact', True)
torch.cuda.synchronize()
print('alternating: allocated %.1f -> %.1f MB' % (mem, torch.cuda.memory_allocated() / 1e6))
ops.check_numerics('soak_dp')
dist.destroy_process_group()
print('soak_dp ok')
