#!/usr/bin/env python3
"""Soak 5 (serving): greedy roll-outs and beam searches of changing geometry from their HIP graphs, with a weight change
now and then; every result is compared with the same call served eagerly (a stale graph or table would show), memory must
stay bounded.   python tools/soak_serve.py [rounds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
bench.load_product()
from insenticap_model_amd import Captioner, synth, ops

dev = torch.device('cuda:0')
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
V, T = bench.V, bench.T
cap = Captioner(synth.make_idx2word(V), synth.SENTIMENT_CATEGORIES, synth.DEFAULT_SETTINGS)
cap.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_weights(V, synth.DEFAULT_SETTINGS).items()})
cap.to(dev).eval()
twin = Captioner(synth.make_idx2word(V), synth.SENTIMENT_CATEGORIES, synth.DEFAULT_SETTINGS)
twin.load_state_dict(cap.state_dict())
twin.to(dev).eval()
twin.enable_rollout_graphs(False)
twin.enable_beam_graphs(False)
big, _ = bench.device_inputs(256, 11, dev)
mem = []
t0 = time.perf_counter()
with torch.no_grad():
    for r in range(rounds):
        for B in (1, 4, 16, 100, 128, 200, 256):
            ins = [x[:B].clone() + (0.001 * r if x.is_floating_point() else 0) for x in big]
            a = cap(*ins, T, 1, mode='rl')
            b = twin(*ins, T, 1, mode='rl')
            assert torch.equal(a[0], b[0]) and torch.equal(a[2], b[2]), ('roll-out tokens', r, B)
            assert float((a[1] - b[1]).abs().max()) < 1e-5, ('roll-out log-probs', r, B)
        for n_img, beam in ((1, 5), (1, 3), (2, 3), (1, 5)):
            i0 = (r * 3) % 200
            args = (big[0][i0:i0 + n_img], big[1][i0:i0 + n_img], big[3][i0:i0 + n_img], big[4][i0:i0 + n_img], beam, 1, T)
            ca, sa, _ = cap.sample_batch(*args)
            cb, sb, _ = twin.sample_batch(*args)
            assert ca == cb, ('beam captions', r, n_img, beam)
            assert max(abs(x - y) for p, q in zip(sa, sb) for x, y in zip(p, q)) < 1e-5
        if r % 7 == 3:                       # new weights: graphs and tables of the old ones must not be served
            with torch.no_grad():
                for q in cap.parameters():
                    q.mul_(1.0 + 1e-3)
            twin.load_state_dict(cap.state_dict())
        torch.cuda.synchronize()
        mem.append(torch.cuda.memory_allocated() / 1e6)
        if r % 5 == 0:
            print('round %d  %.1f s  allocated %.1f MB  roll-out graphs %d' % (r, time.perf_counter() - t0, mem[-1],
                  len(cap.__dict__.get('_rollout_graphs') or {})), flush=True)
ops.check_numerics('soak_serve')
assert mem[-1] <= max(mem[:8]) * 1.10 + 64, mem
print('soak_serve ok: %d rounds' % rounds)
