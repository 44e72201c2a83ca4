#!/usr/bin/env python3
"""Which Python call sites of one eager XE iteration (B=128 + 80) launch torch kernels: a TorchDispatchMode tally."""
import os, sys, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
bench.load_product()
from torch.utils._python_dispatch import TorchDispatchMode
from insenticap_model_amd import Captioner, synth
from insenticap_model_amd.train import xe_train_step
dev = torch.device('cuda:0')
V, R, T = bench.V, bench.R, bench.T
cap = Captioner(synth.make_idx2word(V), synth.SENTIMENT_CATEGORIES, synth.DEFAULT_SETTINGS)
cap.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_weights(V, synth.DEFAULT_SETTINGS).items()})
cap.to(dev).train()
optim, xc, dc = cap.get_optim_criterion(4e-4)
d = synth.make_inputs(128, V, synth.DEFAULT_SETTINGS, regions=R, seq_len=T, seed=500)
s = synth.make_inputs(80, V, synth.DEFAULT_SETTINGS, regions=R, seq_len=T, seed=600)
tt = lambda x: torch.from_numpy(x).to(dev)
fact = (None, tt(d['fc_feats']), tt(d['att_feats']), (tt(d['captions']), d['lengths']), tt(d['cpt_words']))
scs = ((tt(s['captions']), s['lengths']), tt(s['cpt_words']), tt(s['senti_words']), tt(s['senti_labels']))
lab = tt(d['senti_labels'])
for _ in range(2):
    xe_train_step(cap, optim, xc, dc, fact, lab, scs, 0.0, 0.1)
torch.cuda.synchronize()
tally = collections.Counter()
SKIP = ('aten.view', 'aten.select', 'aten.slice', 'aten.unbind', 'aten.detach', 'aten._unsafe_view', 'aten.t.', 'aten.as_strided',
        'aten.reshape', 'aten.unsqueeze', 'aten.squeeze', 'aten.expand', 'aten.permute', 'aten.transpose', 'aten.alias', 'aten.empty',
        'aten.split', 'aten.narrow', 'aten._to_copy.default_nokernel', 'aten.lift_fresh', 'aten.is_pinned', 'aten._pin_memory')


class Tally(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(SKIP):
            site = 'unknown'
            for fr in reversed(traceback.extract_stack(limit=14)):
                if 'insenticap_model_amd' in fr.filename:
                    site = '%s:%d' % (os.path.basename(fr.filename), fr.lineno)
                    break
            tally[(name, site)] += 1
        return func(*args, **(kwargs or {}))

with Tally():
    xe_train_step(cap, optim, xc, dc, fact, lab, scs, 0.0, 0.1)
torch.cuda.synchronize()
print('torch ops seen on the main thread (autograd backward runs on its own thread: not tallied here):', sum(tally.values()))
for (name, site), n in tally.most_common(60):
    print('%4d  %-46s %s' % (n, name, site))
