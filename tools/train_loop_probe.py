#!/usr/bin/env python3
"""A training epoch the way a trainer runs it (train_xe.py:132-192): the package's loaders over synthetic images with captions
of DIFFERENT lengths, DevicePrefetcher, one XE iteration per batch - eager step, graph-served step with the reference's tight
padding (every longest-caption length its own geometry), graph-served step with captions padded to one width.
    python tools/train_loop_probe.py [images]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
bench.load_product()
from insenticap_model_amd import Captioner, data, synth
from insenticap_model_amd.train import xe_train_step
from insenticap_model_amd.train_graph import XETrainGraph

dev = torch.device('cuda:0')
n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
V, R, T = bench.V, bench.R, bench.T
st = synth.DEFAULT_SETTINGS
rng = np.random.default_rng(7)
fns = ['img%05d' % i for i in range(n_img)]
fc = {fn: rng.standard_normal(2048, dtype=np.float32) * 0.5 for fn in fns}
att = {fn: rng.standard_normal((6, 6, 2048), dtype=np.float32) * 0.5 for fn in fns}
def caption():
    n = int(rng.integers(6, T))          # words between <SOS> and <EOS>: 6 .. T - 1
    return [1] + rng.integers(4, V, size=n).tolist() + [2]
caps = {fn: [caption() for _ in range(4)] for fn in fns}           # 4 captions per image: 32 images = 128 rows
cpts = {fn: rng.integers(4, V, size=5).tolist() for fn in fns}
scs_rows = [(caption(), rng.integers(4, V, size=5).tolist(), rng.integers(4, V, size=10).tolist(), int(rng.integers(0, 3)))
            for _ in range(80 * (n_img // 32))]


def epoch(mode):
    torch.manual_seed(0)
    cap = Captioner(synth.make_idx2word(V), synth.SENTIMENT_CATEGORIES, st)
    cap.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_weights(V, st).items()})
    cap.to(dev).train()
    optim, xc, dc = cap.get_optim_criterion(4e-4)
    width = 'full' if mode == 'graph, fixed width' else None
    res = 'resident' in mode
    cl = data.get_caption_dataloader(DFC if res else fc, DATT if res else att, caps, cpts, 0, T, 5, 32, num_workers=0, shuffle=True,
                                     caption_width=width, dedup='dedup' in mode)
    sl = data.get_senti_corpus_with_sentis_dataloader(scs_rows, 0, T, 5, 10, 80, num_workers=0, shuffle=True,
                                                      caption_width=width)
    g = XETrainGraph(cap, optim, xc, dc, grad_clip=0.1, warmup=2) if not mode.startswith('eager') else None
    import warnings
    n, widths = 0, set()
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for ep in range(2):
            for fact, scs in zip(data.DevicePrefetcher(cl, dev), data.DevicePrefetcher(sl, dev)):
                fns_, fc_, att_, (caps_, lengths), cpts_ = fact
                labels = torch.zeros(fc_.shape[0], dtype=torch.int64, device=dev)
                widths.add((caps_.shape[1], scs[0][0].shape[1]))
                if g is None:
                    out = xe_train_step(cap, optim, xc, dc, fact, labels, scs, 0.0, 0.1)
                else:
                    out = g.step(fact, labels, scs, 0.0)
                n += 1
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
    print('%-42s %3d iterations, %6.2f ms each (loader + H2D + step), %2d caption-width pairs%s, last loss %.3f' % (
        mode, n, el / n * 1e3, len(widths), '' if g is None else ', %d captures, %d replays, %d eager' % (
            g.captures, g.replays, g.eager_steps), float(out['all_loss'])), flush=True)


DFC = data.DeviceFeatureStore.from_arrays(fns, [fc[f] for f in fns], dev)
DATT = data.DeviceFeatureStore.from_arrays(fns, [att[f] for f in fns], dev)
for m in ('eager', 'graph, tight', 'graph, tight, dedup', 'graph, features resident on the device', 'eager, features resident on the device'):
    epoch(m)
