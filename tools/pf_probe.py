import os, sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from insenticap_model_amd import data
rng = np.random.default_rng(7)
n_img, T, V = 256, 20, 10000
fns = ['img%05d' % i for i in range(n_img)]
fc = {fn: rng.standard_normal(2048, dtype=np.float32) for fn in fns}
att = {fn: rng.standard_normal((6, 6, 2048), dtype=np.float32) for fn in fns}
caps = {fn: [[1] + rng.integers(4, V, size=int(rng.integers(6, T))).tolist() + [2] for _ in range(4)] for fn in fns}
cpts = {fn: rng.integers(4, V, size=5).tolist() for fn in fns}
cl = data.get_caption_dataloader(fc, att, caps, cpts, 0, T, 5, 32, num_workers=0, shuffle=True)
dev = torch.device('cuda:0')
torch.zeros(1, device=dev)
pf = data.DevicePrefetcher(cl, dev)
for rep in range(2):
    it = iter(cl); tl = ts = 0.0; n = 0
    while True:
        t0 = time.perf_counter()
        try:
            b = next(it)
        except StopIteration:
            break
        t1 = time.perf_counter()
        d = pf._stage(b)
        t2 = time.perf_counter()
        tl += t1 - t0; ts += t2 - t1; n += 1
    torch.cuda.synchronize()
    print('rep %d: next(it) %.2f ms, _stage %.2f ms per batch (threads %d)' % (rep, tl / n * 1e3, ts / n * 1e3, torch.get_num_threads()))
