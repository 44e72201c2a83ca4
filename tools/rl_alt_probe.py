import os, sys, time
sys.path.insert(0, '.')
import torch
import bench
bench.load_product()
from insenticap_model_amd import Detector, synth
dev = torch.device('cuda:0')
V, T, R = bench.V, bench.T, bench.R
st = dict(synth.DEFAULT_SETTINGS, **synth.HELPER_SETTINGS)
tt = torch.from_numpy
def run(nb, B=64, n=60):
    det = Detector(synth.make_idx2word(V), T, synth.SENTIMENT_CATEGORIES, {'cap_lr': 4e-5}, st)
    det.captioner.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_weights(V, st, seed=0).items()})
    det.to(dev)
    batches, split = synth.make_rl_batches(nb, B, V, st, grid=(6, 6), seq_len=T, seed=90)
    det.set_ciderd_scorer(split)
    facts = [[(b[0], tt(b[1]).to(dev), tt(b[2]).to(dev), (tt(b[3][0]).to(dev), b[3][1]), tt(b[4]).to(dev), tt(b[5]).to(dev), b[6])] for b in batches]
    s = synth.make_inputs(80, V, st, regions=R, seq_len=T, seed=91)
    scs = [((tt(s['captions']).to(dev), s['lengths']), tt(s['cpt_words']).to(dev), tt(s['senti_words']).to(dev), tt(s['senti_labels']).to(dev))]
    print('caption widths', [f[0][3][0].shape[1] for f in facts])
    for i in range(8):
        det((facts[i % nb], scs), 'fact', True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        det((facts[i % nb], scs), 'fact', True)
    torch.cuda.synchronize()
    g = det._rl_graph
    print('%d alternating batch(es): %.2f ms per iteration; captures %d replays %d eager %d geometries %d' % (
        nb, (time.perf_counter() - t0) / n * 1e3, g.captures, g.replays, g.eager_steps, len(g._geoms)), flush=True)
run(1); run(2); run(3)
