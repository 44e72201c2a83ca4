#!/usr/bin/env python3
"""Where a batch's host time goes: dataset items, collate (one row per caption: an image's features repeated), pinning,
the copy to the device - for the package's caption loader at 32 images x 4 captions, 6 x 6 x 2048 regions.
    python tools/loader_probe.py [workers]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from insenticap_model_amd import data

workers = int(sys.argv[1]) if len(sys.argv) > 1 else 0
dedup = len(sys.argv) > 2 and sys.argv[2] == 'dedup'
rng = np.random.default_rng(7)
n_img, T, V = 512, 20, 10000
fns = ['img%05d' % i for i in range(n_img)]
fc = {fn: rng.standard_normal(2048, dtype=np.float32) for fn in fns}
att = {fn: rng.standard_normal((6, 6, 2048), dtype=np.float32) for fn in fns}
caps = {fn: [[1] + rng.integers(4, V, size=int(rng.integers(6, T))).tolist() + [2] for _ in range(4)] for fn in fns}
cpts = {fn: rng.integers(4, V, size=5).tolist() for fn in fns}
kw = dict(dedup=True) if dedup else {}
cl = data.get_caption_dataloader(fc, att, caps, cpts, 0, T, 5, 32, num_workers=workers, shuffle=True, **kw)
t0 = time.perf_counter(); n = 0
for b in cl:
    n += 1
t_load = (time.perf_counter() - t0) / n
print('workers=%d%s: loader alone %.2f ms per batch' % (workers, ' dedup' if dedup else '', t_load * 1e3))
if torch.cuda.is_available():
    dev = torch.device('cuda:0')
    for b in data.DevicePrefetcher(cl, dev):
        pass
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 0
    for b in data.DevicePrefetcher(cl, dev):
        n += 1
    torch.cuda.synchronize()
    print('            loader + DevicePrefetcher %.2f ms per batch; att on device %s' % ((time.perf_counter() - t0) / n * 1e3, tuple(b[2].shape)))
