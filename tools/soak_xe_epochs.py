#!/usr/bin/env python3
"""Soak 6: XE epochs the way train_xe.py runs them - graph-served training steps, then a validation pass in eval mode under
no_grad, the scheduled-sampling probability and the learning rate changing between epochs, a checkpoint written each epoch.
    python tools/soak_xe_epochs.py [epochs]"""
import io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
bench.load_product()
from insenticap_model_amd import Captioner, synth, ops
from insenticap_model_amd.train_graph import XETrainGraph

dev = torch.device('cuda:0')
epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 6
V, R, T = bench.V, bench.R, bench.T
st = synth.DEFAULT_SETTINGS
cap = Captioner(synth.make_idx2word(V), synth.SENTIMENT_CATEGORIES, st)
cap.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_weights(V, st).items()})
cap.to(dev)
optim, xc, dc = cap.get_optim_criterion(4e-4)
tt = lambda x: torch.from_numpy(x).to(dev)
def batch(seed):
    d = synth.make_inputs(128, V, st, regions=R, seq_len=T, seed=seed)
    s = synth.make_inputs(80, V, st, regions=R, seq_len=T, seed=seed + 1)
    return ((None, tt(d['fc_feats']), tt(d['att_feats']), (tt(d['captions']), d['lengths']), tt(d['cpt_words'])),
            tt(d['senti_labels']), ((tt(s['captions']), s['lengths']), tt(s['cpt_words']), tt(s['senti_words']), tt(s['senti_labels'])))
train = [batch(10 + 2 * i) for i in range(6)]
val = [batch(500 + 2 * i) for i in range(3)]
g = XETrainGraph(cap, optim, xc, dc, grad_clip=0.1, warmup=2)
t0 = time.perf_counter()
for ep in range(epochs):
    ss = min(0.25, 0.05 * ep)                         # train_xe.py:209-212
    for grp in optim.param_groups:
        grp['lr'] = 4e-4 * (0.8 ** ep)
    cap.train()
    for it in range(24):
        out = g.step(*train[it % 6], ss)
    cap.eval()
    with torch.no_grad():
        vl = 0.0
        for fact, labels, scs in val:
            pred = cap(fact[1], fact[2], fact[4], fact[3][0], labels, 0.0, mode='xe')
            vl += float(xc(pred, fact[3][0][:, 1:], fact[3][1]))
    buf = io.BytesIO()
    torch.save({'model': cap.state_dict(), 'optimizer': optim.state_dict()}, buf)
    torch.cuda.synchronize()
    print('epoch %d  ss %.2f  train %.4f  val %.4f  %.1f s  captures %d replays %d eager %d  allocated %.0f MB' % (
        ep, ss, float(out['all_loss']), vl / 3, time.perf_counter() - t0, g.captures, g.replays, g.eager_steps,
        torch.cuda.memory_allocated() / 1e6), flush=True)
ops.check_numerics('soak_xe_epochs')
assert g.captures <= epochs and g.replays >= epochs * 24 - 3 * epochs - 3
print('soak_xe_epochs ok')
