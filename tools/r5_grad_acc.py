#!/usr/bin/env python3
"""Gradient accuracy of one XE iteration (B = 256 + 80 and 128 + 40, V = 10k): merged chain / two chains on the
split-f16 engine against the exact-fp32 engine (isc_set_h3_mode(0), two chains).  Prints per-tensor max |err| / gmax."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, torch
from insenticap_model_amd import ops, dp
import test_gpu_dp as T

cfg = T.FULL
def grads(pair, mode, lo, hi):
    os.environ['ISC_PAIR_UNROLLS'] = '1' if pair else '0'
    ops.set_h3_mode(mode)
    cap = T._make(cfg)
    arena = dp.GradArena(cap.parameters())
    optim, xc, dc = cap.get_optim_criterion(4e-4)
    fact, labels, scs = T._batches(lo, hi, cfg)
    from insenticap_model_amd.train import xe_forward_backward
    dv = lambda x: x.to('cuda:0')
    _, fc, att, (caps, lengths), cpts = fact[:5]
    (s_caps, s_len), s_cpts, s_sentis, s_labels = scs
    fc, att, caps, cpts, labels, s_caps, s_cpts, s_sentis, s_labels = map(dv, (fc, att, caps, cpts, labels, s_caps, s_cpts, s_sentis, s_labels))
    xe_forward_backward(cap, optim, xc, dc, (fc, att, caps, lengths, cpts), labels, (s_caps, s_len, s_cpts, s_sentis, s_labels), 0.0, arena, None, False)
    torch.cuda.synchronize()
    out = {}
    off = 0
    for k, q in cap.named_parameters():
        out[k] = arena.flat[off:off + q.numel()].detach().cpu().double().numpy().copy(); off += q.numel()
    ops.set_h3_mode(1)
    return out
for lo, hi in ((0, cfg['B']), (0, cfg['B'] // 2)):
    ref = grads(False, 0, lo, hi)
    a = grads(True, 1, lo, hi)
    b = grads(False, 1, lo, hi)
    print('rows', lo, hi)
    for k in ref:
        gm = np.abs(ref[k]).max()
        if gm == 0: continue
        ea, eb = np.abs(a[k] - ref[k]).max() / gm, np.abs(b[k] - ref[k]).max() / gm
        flag = ' <<<' if max(ea, eb) > 1e-4 else ''
        print('%-45s gmax %.3e  merged %.2e  two %.2e%s' % (k, gm, ea, eb, flag))
