import sys, time, torch
sys.path.insert(0, '.')
import bench
bench.load_product()
from insenticap_model_amd import Captioner, synth
dev = torch.device('cuda:0')
V, R, T = bench.V, bench.R, bench.T
st = synth.DEFAULT_SETTINGS
cap = Captioner(synth.make_idx2word(V), synth.SENTIMENT_CATEGORIES, st)
cap.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_weights(V, st).items()})
cap.to(dev).eval()
_, xc, dc = cap.get_optim_criterion(4e-4)
tt = lambda x: torch.from_numpy(x).to(dev)
for B in (128, 512):
    d = synth.make_inputs(B, V, st, regions=R, seq_len=T, seed=500)
    fc, att, caps, cpts, lab = tt(d['fc_feats']), tt(d['att_feats']), tt(d['captions']), tt(d['cpt_words']), tt(d['senti_labels'])
    def val():
        with torch.no_grad():
            pred = cap(fc, att, cpts, caps, lab, 0.0, mode='xe')
            return xc(pred, caps[:, 1:], d['lengths'])
    for _ in range(3): val()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): l = val()
    torch.cuda.synchronize()
    print('validation forward_xe + XE loss, B=%d: %.2f ms per batch (loss %.4f)' % (B, (time.perf_counter() - t0) / 20 * 1e3, float(l)))
