import os, sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import bench
bench.load_product()
from insenticap_model_amd import Detector, data, synth
dev = torch.device('cuda:0')
B, n_img = 512, 4096
V, T = bench.V, bench.T
st = dict(synth.DEFAULT_SETTINGS, **synth.HELPER_SETTINGS)
rng = np.random.default_rng(7)
fns = ['img%05d' % i for i in range(n_img)]
fc = {fn: rng.standard_normal(2048, dtype=np.float32) * 0.5 for fn in fns}
att = {fn: rng.standard_normal((6, 6, 2048), dtype=np.float32) * 0.5 for fn in fns}
cap_ = lambda: [1] + rng.integers(4, V, size=int(rng.integers(6, T))).tolist() + [2]
caps = {fn: [cap_() for _ in range(5)] for fn in fns}
cpts = {fn: rng.integers(4, V, size=5).tolist() for fn in fns}
sentis = {fn: rng.integers(4, V, size=10).tolist() for fn in fns}
dfc = data.DeviceFeatureStore.from_arrays(fns, [fc[f] for f in fns], dev)
datt = data.DeviceFeatureStore.from_arrays(fns, [att[f] for f in fns], dev)
fl = data.get_rl_fact_dataloader(dfc, datt, caps, cpts, sentis, 0, T, 5, 10, B, shuffle=True)
for wrap in (False, True, False, True):
    src = data.DevicePrefetcher(fl, dev) if wrap else fl
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 0
    for b in src:
        x = b[1].to(dev) if isinstance(b[1], data.RowGather) else b[1]
        y = b[2].to(dev) if isinstance(b[2], data.RowGather) else b[2]
        n += 1
    torch.cuda.synchronize()
    print('prefetcher=%s: %.2f ms per batch (loader only)' % (wrap, (time.perf_counter() - t0) / n * 1e3))

# ---- the same loaders under the RL iteration: where does the host wait?
class Timed:
    def __init__(self, src): self.src, self.t, self.n = src, 0.0, 0
    def __len__(self): return len(self.src)
    def __iter__(self):
        it = iter(self.src)
        while True:
            t0 = time.perf_counter()
            try:
                b = next(it)
            except StopIteration:
                return
            self.t += time.perf_counter() - t0; self.n += 1
            yield b
scs_rows = [(cap_(), rng.integers(4, V, size=5).tolist(), rng.integers(4, V, size=10).tolist(), int(rng.integers(0, 3))) for _ in range(640)]
det = Detector(synth.make_idx2word(V), T, synth.SENTIMENT_CATEGORIES, {'cap_lr': 4e-5}, st)
det.captioner.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_weights(V, st, seed=0).items()})
det.to(dev)
det.set_ciderd_scorer({'train': caps})
import warnings; warnings.simplefilter('ignore')
sl = data.get_senti_corpus_with_sentis_dataloader(scs_rows, 0, T, 5, 10, 80, shuffle=True)
import gc
if os.environ.get('NOGC'): gc.disable()
if os.environ.get('FREEZE'): gc.collect(); gc.freeze()
for wrap in (False, True, False, True):
    a = Timed(data.DevicePrefetcher(fl, dev) if wrap else fl)
    b = Timed(data.DevicePrefetcher(sl, dev) if wrap else sl)
    for ep in range(2):
        a.t = b.t = 0.0; a.n = b.n = 0
        torch.cuda.synchronize(); t0 = time.perf_counter()
        det((a, b), 'fact', True)
        torch.cuda.synchronize(); el = time.perf_counter() - t0
    print('prefetcher=%s: %.1f ms per iteration; in next(fact loader) %.1f ms, in next(scs loader) %.1f ms per iteration' % (
        wrap, el / a.n * 1e3, a.t / a.n * 1e3, b.t / max(b.n, 1) * 1e3), flush=True)
