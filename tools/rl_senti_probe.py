#!/usr/bin/env python3
"""The OTHER half of the reference's RL epochs (train_rl.py:232-235): `detector((senti_data, scs_data), 'senti', True)` - images
with sentiment labels, no ground-truth captions: sampled + greedy roll-out, classifier reward only, seq2seq unroll, backward.
    python tools/rl_senti_probe.py [B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
bench.load_product()
from insenticap_model_amd import Detector, synth
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
V, T, R = bench.V, bench.T, bench.R
st = dict(synth.DEFAULT_SETTINGS, **synth.HELPER_SETTINGS)
det = Detector(synth.make_idx2word(V), T, synth.SENTIMENT_CATEGORIES, {'cap_lr': 4e-5}, st)
det.captioner.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_weights(V, st, seed=0).items()})
det.to(dev)
tt = torch.from_numpy
batches, split = synth.make_rl_batches(1, B, V, st, grid=(6, 6), seq_len=T, seed=90)
det.set_ciderd_scorer(split)
b = batches[0]
labels = torch.randint(0, len(synth.SENTIMENT_CATEGORIES), (B,))
senti = [(b[0], tt(b[1]).to(dev), tt(b[2]).to(dev), tt(b[4]).to(dev), tt(b[5]).to(dev), labels.to(dev))]
fact = [(b[0], tt(b[1]).to(dev), tt(b[2]).to(dev), (tt(b[3][0]).to(dev), b[3][1]), tt(b[4]).to(dev), tt(b[5]).to(dev), b[6])]
s = synth.make_inputs(80, V, st, regions=R, seq_len=T, seed=91)
scs = [((tt(s['captions']).to(dev), s['lengths']), tt(s['cpt_words']).to(dev), tt(s['senti_words']).to(dev), tt(s['senti_labels']).to(dev))]
for kind, data in (('senti', senti), ('fact', fact)):
    for i in range(5):
        out = det((data, scs), kind, True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(20):
        out = det((data, scs), kind, True)
    torch.cuda.synchronize()
    print("Detector.forward(data, '%s', True), B=%d: %.1f ms per iteration  %s" % (kind, B, (time.perf_counter() - t0) / 20 * 1e3, {k: round(v, 4) for k, v in out.items()}), flush=True)
with torch.no_grad():
    for kind, data in (('fact', fact), ('senti', senti)):
        for i in range(3):
            out = det((data,), kind, False)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(10):
            out = det((data,), kind, False)
        torch.cuda.synchronize()
        print("Detector.forward(data, '%s', False) [validation], B=%d: %.1f ms per iteration  %s" % (kind, B, (time.perf_counter() - t0) / 10 * 1e3, {k: round(v, 4) for k, v in out.items()}), flush=True)
