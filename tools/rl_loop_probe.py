#!/usr/bin/env python3
"""An RL epoch the way train_rl.py runs it (train_rl.py:232-242): `detector((fact_loader, scs_loader), 'fact', True)` over the
package's loaders - plain and wrapped in DevicePrefetcher - with synthetic images, B images per iteration.
    python tools/rl_loop_probe.py [B [images]]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
bench.load_product()
from insenticap_model_amd import Detector, data, synth

dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
n_img = int(sys.argv[2]) if len(sys.argv) > 2 else 4 * B
V, T = bench.V, bench.T
st = dict(synth.DEFAULT_SETTINGS, **synth.HELPER_SETTINGS)
rng = np.random.default_rng(7)
fns = ['img%05d' % i for i in range(n_img)]
fc = {fn: rng.standard_normal(2048, dtype=np.float32) * 0.5 for fn in fns}
att = {fn: rng.standard_normal((6, 6, 2048), dtype=np.float32) * 0.5 for fn in fns}
def caption():
    return [1] + rng.integers(4, V, size=int(rng.integers(6, T))).tolist() + [2]
caps = {fn: [caption() for _ in range(5)] for fn in fns}
cpts = {fn: rng.integers(4, V, size=5).tolist() for fn in fns}
sentis = {fn: rng.integers(4, V, size=10).tolist() for fn in fns}
scs_rows = [(caption(), rng.integers(4, V, size=5).tolist(), rng.integers(4, V, size=10).tolist(), int(rng.integers(0, 3)))
            for _ in range(80 * 8)]
det = Detector(synth.make_idx2word(V), T, synth.SENTIMENT_CATEGORIES, {'cap_lr': 4e-5}, st)
det.captioner.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_weights(V, st, seed=0).items()})
det.to(dev)
det.set_ciderd_scorer({'train': caps})
import warnings
warnings.simplefilter('ignore')
workers = int(os.environ.get('WORKERS', '0'))
dfc = data.DeviceFeatureStore.from_arrays(fns, [fc[f] for f in fns], dev)
datt = data.DeviceFeatureStore.from_arrays(fns, [att[f] for f in fns], dev)
for mode in ('plain loaders', 'DevicePrefetcher', 'features resident on the device', 'features resident on the device + DevicePrefetcher'):
    width = 'full' if 'fixed' in mode else None
    res = 'resident' in mode
    fl = data.get_rl_fact_dataloader(dfc if res else fc, datt if res else att, caps, cpts, sentis, 0, T, 5, 10, B, num_workers=0 if res else workers, shuffle=True, caption_width=width)
    sl = data.get_senti_corpus_with_sentis_dataloader(scs_rows, 0, T, 5, 10, 80, num_workers=0, shuffle=True, caption_width=width)
    if 'Prefetcher' in mode:
        fl, sl = data.DevicePrefetcher(fl, dev), data.DevicePrefetcher(sl, dev)
    for ep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = det((fl, sl), 'fact', True)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
    g = det._rl_graph
    print('%-40s epoch of %d iterations: %.1f ms per iteration; graph: %d captures, %d replays, %d eager so far' % (
        mode, len(fl), el / len(fl) * 1e3, g.captures, g.replays, g.eager_steps), flush=True)
