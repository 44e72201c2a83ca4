"""Timing of the few-row decode paths (beam-5 search of one image, 4-caption greedy roll-out), few-row kernels on / off,
from HIP graphs and eager.  python tools/rows_lab.py [--reps 40] [--profile]  (--profile: graphs off, few reps: for rocprofv3)"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from insenticap_model_amd import Captioner, ops, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=40)
    ap.add_argument('--profile', action='store_true')
    ap.add_argument('--nt', type=int, default=-1, help='isc_set_rows_nt mode (-1: the library default)')
    ap.add_argument('--graphs-only', action='store_true')
    ap.add_argument('--rows-only', action='store_true')
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    V, st, T = 10000, synth.DEFAULT_SETTINGS, 20
    cap = Captioner(synth.make_idx2word(V), synth.SENTIMENT_CATEGORIES, st)
    w = synth.make_weights(V, st, seed=0)
    w['classifier.bias'][2] = -1e4            # never <EOS>: every search / roll-out runs its T steps
    cap.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
    cap.to(dev).eval()
    d = synth.make_inputs(8, V, st, regions=36, seq_len=T, seed=11)
    t = lambda k: torch.from_numpy(np.asarray(d[k])).to(dev)
    fc, att, cpt, sw, lab = t('fc_feats'), t('att_feats'), t('cpt_words'), t('senti_words'), t('senti_labels')
    if a.nt >= 0:
        ops._lib.load().isc_set_rows_nt(a.nt)

    def timed(fn, reps):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        lat = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            lat.append((time.perf_counter() - t0) * 1e3)
        return float(np.median(lat)), float(np.min(lat))

    out = {}
    for rows_on in ((True,) if a.rows_only else (True, False)):
        cap.rows_step = rows_on
        for graphs in ((True,) if a.graphs_only else (False,) if a.profile else (True, False)):
            cap.enable_beam_graphs(graphs)
            cap.enable_rollout_graphs(graphs)
            beam = lambda: cap.sample(fc[0], att[0], sw[0], lab[0:1], 5, 1, T)
            with torch.no_grad():
                greedy = lambda: cap(fc[:4], att[:4], cpt[:4], sw[:4], lab[:4], T, 1, mode='rl')
                key = ('rows' if rows_on else 'general') + ('+graphs' if graphs else '+eager')
                p50, best = timed(beam, 5 if (a.profile or a.graphs_only) else a.reps)
                steps = cap.last_beam_steps
                g50, gbest = timed(greedy, 5 if (a.profile or a.graphs_only) else a.reps)
            out[key] = dict(beam5_p50_ms=round(p50, 4), beam5_min_ms=round(best, 4), steps=steps,
                            us_per_step=round(1e3 * p50 / steps, 2), greedy4_p50_ms=round(g50, 4), greedy4_min_ms=round(gbest, 4))
            print(key, out[key], flush=True)


if __name__ == '__main__':
    main()
