"""Phase stamps of every kernel of a few-row beam step (library built with -DROWS_STAMP=1: tools/_lab/stamp, see
tools/r4_stamp_build.sh).  ISC_HIP_LIB=tools/_lab/stamp/libinsenticap_hip_stamp.so python tools/stamp_step.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from insenticap_model_amd import Captioner, _lib, ops, synth  # noqa: E402

REGION = 131072
NAMES = {0: 'att-LSTM', 1: 'projections', 2: 'gated scan', 3: 'lang-LSTM', 4: 'classifier', 5: 'select'}
SLOTS = {0: ['start', 'pre-B0', 'issued/DMA', 'staged(B1)', 'partials', 'after B2', 'end'],
         3: ['start', 'pre-B0', 'issued/DMA', 'staged(B1)', 'partials', 'after B2', 'end'],
         2: ['start', 'loads issued', 'scores', 'after B', 'sums', 'combined', 'end'],
         4: ['start', 'B1', 'all issued', 'partials', 'after B2', '-', 'end'],
         5: ['start', 'loads issued', 'rounds start', 'merged', 'scored', 'after barrier 2', 'after barrier 3', 'end']}


def main():
    lib = _lib.load()
    for name in ('isc_rows_set_stamp', 'isc_pw_set_stamp'):
        getattr(lib, name).restype = C.c_int
        getattr(lib, name).argtypes = [C.c_void_p]
    dev = torch.device('cuda:0')
    V, st, T = 10000, synth.DEFAULT_SETTINGS, 20
    cap = Captioner(synth.make_idx2word(V), synth.SENTIMENT_CATEGORIES, st)
    w = synth.make_weights(V, st, seed=0)
    w['classifier.bias'][2] = -1e4
    cap.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
    cap.to(dev).eval()
    cap.enable_beam_graphs(False)
    d = synth.make_inputs(1, V, st, regions=36, seq_len=T, seed=11)
    t = lambda k: torch.from_numpy(np.asarray(d[k])).to(dev)
    fc, att, sw, lab = t('fc_feats'), t('att_feats'), t('senti_words'), t('senti_labels')
    for _ in range(3):
        cap.sample(fc[0], att[0], sw[0], lab[0:1], 5, 1, T)
    stamp = torch.zeros(6 * REGION, dtype=torch.int64, device=dev)
    assert lib.isc_rows_set_stamp(stamp.data_ptr()) == 0 and lib.isc_pw_set_stamp(stamp.data_ptr()) == 0
    cap.sample(fc[0], att[0], sw[0], lab[0:1], 5, 1, T)
    torch.cuda.synchronize()
    hs = stamp.cpu().numpy().reshape(6, 1024, 16, 8)
    for kid in (0, 1, 2, 3, 4, 5):
        r = hs[kid]
        if not r[..., 0].any() and kid != 1:
            continue
        if kid == 1:
            continue
        t0 = r[..., 0][r[..., 0] > 0].min()
        print('%s (x10 ns from the launch\'s first wave start; median / max over waves that stamped)' % NAMES[kid])
        for s, nm in enumerate(SLOTS[kid]):
            v = r[..., s][r[..., s] > 0] - t0
            if v.size:
                print('   %-14s median %6d  max %6d  (n=%d)' % (nm, np.median(v), v.max(), v.size))
        if kid == 5:
            continue
        c = r[..., 7][r[..., 7] > 0]
        if c.size:
            print('   shader cycles between CLK0 and CLK1: median %d  max %d' % (np.median(c), c.max()))


if __name__ == '__main__':
    main()
