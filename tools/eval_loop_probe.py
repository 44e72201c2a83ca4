#!/usr/bin/env python3
"""The evaluation loop of the reference (train_rl.py / eval: one `detector.sample(fc, att, sentis)` per image - sentiment
detector, then beam search): per-image latency of the pieces.   python tools/eval_loop_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
bench.load_product()
from insenticap_model_amd import Detector, synth
dev = torch.device('cuda:0')
V, T = bench.V, bench.T
st = dict(synth.DEFAULT_SETTINGS, **synth.HELPER_SETTINGS)
det = Detector(synth.make_idx2word(V), T, synth.SENTIMENT_CATEGORIES, {'cap_lr': 4e-5}, st)
det.captioner.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_weights(V, st, seed=0).items()})
det.to(dev).eval()
ins, _ = bench.device_inputs(64, 5, dev)
fc, att, _, sw, lab = ins
att6 = att.reshape(64, 6, 6, 2048)
def timeit(fn, n=64):
    for i in range(8): fn(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): fn(i % 64)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
with torch.no_grad():
    a = timeit(lambda i: det.sample(fc[i], att6[i], sw[i], 5, 1))
    b = timeit(lambda i: det.captioner.sample(fc[i], att[i], sw[i], lab[i:i + 1], 5, 1, T))
    c = timeit(lambda i: det.senti_detector.sample(att6[i:i + 1], det.senti_threshold))
print('Detector.sample (sentiment detector + beam 5): %.3f ms per image; beam search alone %.3f; sentiment detector alone %.3f' % (a, b, c))
