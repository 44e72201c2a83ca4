import os, sys, time
sys.path.insert(0, '.')
import torch
import bench
bench.load_product()
from insenticap_model_amd import Detector, synth, rewards
from insenticap_model_amd.train import xe_train_step
dev = torch.device('cuda:0')
V, T, R = bench.V, bench.T, bench.R
st = dict(synth.DEFAULT_SETTINGS, **synth.HELPER_SETTINGS)
tt = torch.from_numpy
det = Detector(synth.make_idx2word(V), T, synth.SENTIMENT_CATEGORIES, {'cap_lr': 4e-5}, st)
det.captioner.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_weights(V, st, seed=0).items()})
det.to(dev)
cap = det.captioner
B = 64
batches, split = synth.make_rl_batches(1, B, V, st, grid=(6, 6), seq_len=T, seed=90)
det.set_ciderd_scorer(split)
b = batches[0]
facts = [(b[0], tt(b[1]).to(dev), tt(b[2]).to(dev), (tt(b[3][0]).to(dev), b[3][1]), tt(b[4]).to(dev), tt(b[5]).to(dev), b[6])]
s = synth.make_inputs(80, V, st, regions=R, seq_len=T, seed=91)
scs = [((tt(s['captions']).to(dev), s['lengths']), tt(s['cpt_words']).to(dev), tt(s['senti_words']).to(dev), tt(s['senti_labels']).to(dev))]
d = synth.make_inputs(128, V, st, regions=R, seq_len=T, seed=500)
fact = (None, tt(d['fc_feats']).to(dev), tt(d['att_feats']).to(dev), (tt(d['captions']).to(dev), d['lengths']), tt(d['cpt_words']).to(dev))
labels = tt(d['senti_labels']).to(dev)
ct = [0.0]
orig = rewards.self_critical_scores
def timed(*a, **k):
    t0 = time.perf_counter(); r = orig(*a, **k); ct[0] += time.perf_counter() - t0; return r
rewards.self_critical_scores = timed
def time_rl(tag, n=40):
    for i in range(6):
        out = det((facts, scs), 'fact', True)
    torch.cuda.synchronize(); ct[0] = 0.0; t0 = time.perf_counter()
    for i in range(n):
        out = det((facts, scs), 'fact', True)
    torch.cuda.synchronize()
    lens = det._rl_graph._geoms[next(iter(det._rl_graph._geoms))].lens_d.float().mean().item()
    print('%s: %.2f ms per RL iteration, CIDEr %.2f ms, mean sampled length %.1f, rewards %s' % (
        tag, (time.perf_counter() - t0) / n * 1e3, ct[0] / n * 1e3, lens, {k: round(v, 4) for k, v in out.items() if 'reward' in k}), flush=True)
time_rl('random weights')
for i in range(400):
    xe_train_step(cap, det.cap_optim, det.cap_xe_crit, det.cap_da_crit, fact, labels, scs[0], 0.25, 0.1)
cap.cpt_feats = cap.fc_feats = None
time_rl('after 400 XE steps')
