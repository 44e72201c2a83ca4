import sys, time, torch
sys.path.insert(0, '.')
import bench
bench.load_product()
from insenticap_model_amd import Captioner, synth
dev = torch.device('cuda:0')
cap = Captioner(synth.make_idx2word(bench.V), synth.SENTIMENT_CATEGORIES, synth.DEFAULT_SETTINGS)
cap.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_weights(bench.V, synth.DEFAULT_SETTINGS).items()})
cap.to(dev).eval()
for B in (4, 128):
    inputs, _ = bench.device_inputs(B, 700 + B, dev)
    pool = [[x.clone() for x in inputs] for _ in range(24)]
    with torch.no_grad():
        for chk in (True, False):
            cap.numerics_checks = chk
            for i in range(4):
                cap(*pool[i], bench.T, 1, mode='rl')
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(20):
                cap(*pool[4 + i], bench.T, 1, mode='rl')          # a NEW tensor object every call: a serving loop
            torch.cuda.synchronize()
            t_new = (time.perf_counter() - t0) / 20
            t0 = time.perf_counter()
            for i in range(20):
                cap(*pool[0], bench.T, 1, mode='rl')
            torch.cuda.synchronize()
            t_same = (time.perf_counter() - t0) / 20
            print('B=%d numerics_checks=%s: new tensors per call %.3f ms, same tensors %.3f ms' % (B, chk, t_new * 1e3, t_same * 1e3))
