"""Diagnostic (round 3): XE iteration at B=1024 - HIP gradients and the fp32 oracle's, both against the fp64 oracle."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from insenticap_model_amd import synth, ops
import test_gpu_train_sizes as M
from oracle import captioner_oracle as O

V, st, B = 10000, synth.DEFAULT_SETTINGS, int(os.environ.get('B', 1024))
w = synth.make_weights(V, st, seed=0)
d = synth.make_inputs(B, V, st, regions=36, seq_len=20, seed=1024)
s2s = synth.make_inputs(80, V, st, regions=36, seq_len=20, seed=1025)
res = {}
for mode in (1, 0):
    ops.set_h3_mode(mode)
    res['hip_h3mode%d' % mode] = M._hip_iteration(w, V, st, d, s2s)[1]
ops.set_h3_mode(1)
res['oracle32'] = M._oracle_iteration(w, V, d, s2s)[1]
orig = O.to_params
O.to_params = lambda ww, dtype=torch.float32, requires_grad=False: orig(ww, torch.float64, requires_grad)
t64 = lambda a: a
import types
# fp64 inputs: make_inputs arrays are float32 -> cast feature arrays
d64 = {k: (v.astype(np.float64) if getattr(v, 'dtype', None) == np.float32 else v) for k, v in d.items()}
s64 = {k: (v.astype(np.float64) if getattr(v, 'dtype', None) == np.float32 else v) for k, v in s2s.items()}
g64 = M._oracle_iteration(w, V, d64, s64)[1]
O.to_params = orig
print('%-45s %10s | %s' % ('tensor', 'max|g|', '  '.join('%-14s' % k for k in res)))
for k, ref in g64.items():
    sc = np.abs(ref).max()
    print('%-45s %10.3e | %s' % (k, sc, '  '.join('%-14.2e' % (np.abs(res[n][k] - ref).max() / (sc + 1e-30)) for n in res)))
