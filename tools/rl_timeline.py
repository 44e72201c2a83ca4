#!/usr/bin/env python3
"""Host / device timeline of one graph-served RL iteration (BASELINE configs[4]) WITHOUT a profiler: host time stamps around
the phases of Detector.forward / RLTrainGraph.step and device events behind every graph replay.
    python tools/rl_timeline.py [B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
bench.load_product()
from insenticap_model_amd import rewards, train_graph, detector as dmod
import insenticap_model_amd.train_graph as tg

dev = torch.device('cuda:0')
LOG, EV = [], []
T0 = [0.0]
on = [False]


def mark(label):
    if on[0]:
        LOG.append((label, (time.perf_counter() - T0[0]) * 1e3))


def ev(label):
    if on[0]:
        e = torch.cuda.Event(enable_timing=True)
        e.record(torch.cuda.current_stream())
        EV.append((label, e))


orig_replay = torch.cuda.CUDAGraph.replay
count = [0]


def replay(self):
    mark('replay %d >' % count[0])
    orig_replay(self)
    ev('graph %d done' % count[0])
    mark('replay %d <' % count[0])
    count[0] += 1


torch.cuda.CUDAGraph.replay = replay


def wrap(mod, name, label):
    f = getattr(mod, name)

    def g(*a, **k):
        mark(label + ' >')
        r = f(*a, **k)
        mark(label + ' <')
        if label == 'cls reward':
            ev('cls reward done')
        return r
    setattr(mod, name, g)


wrap(rewards, 'self_critical_scores', 'cider half')
wrap(rewards, 'get_cls_reward', 'cls reward')
orig_stage = tg.RLTrainGraph._stage


def stage(self, geo, t):
    mark('stage >')
    r = orig_stage(self, geo, t)
    mark('stage <')
    return r


tg.RLTrainGraph._stage = stage
orig_step = tg.RLTrainGraph.step


def step(self, *a, **k):
    mark('graph.step >')
    r = orig_step(self, *a, **k)
    mark('graph.step <')
    return r


tg.RLTrainGraph.step = step

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
# the body of bench.bench_rl, with one traced iteration at the end
from insenticap_model_amd import Detector, synth
V, T, R = bench.V, bench.T, bench.R
st = dict(synth.DEFAULT_SETTINGS, **synth.HELPER_SETTINGS)
det = Detector(synth.make_idx2word(V), T, synth.SENTIMENT_CATEGORIES, {'cap_lr': 4e-5}, st)
det.captioner.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_weights(V, st, seed=0).items()})
det.to(dev)
batches, split = synth.make_rl_batches(1, B, V, st, grid=(6, 6), seq_len=T, seed=90)
det.set_ciderd_scorer(split)
tt = torch.from_numpy
b = batches[0]
fns = b[0]
fact = [(fns, tt(b[1]).to(dev), tt(b[2]).to(dev), (tt(b[3][0]).to(dev), b[3][1]), tt(b[4]).to(dev), tt(b[5]).to(dev),
         {fn: b[6][fn] for fn in fns})]
s = synth.make_inputs(80, V, st, regions=R, seq_len=T, seed=91)
scs = [((tt(s['captions']).to(dev), s['lengths']), tt(s['cpt_words']).to(dev), tt(s['senti_words']).to(dev),
        tt(s['senti_labels']).to(dev))]
for _ in range(6):
    det((fact, scs), 'fact', True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    det((fact, scs), 'fact', True)
torch.cuda.synchronize()
print('ms per iteration (10): %.2f' % ((time.perf_counter() - t0) * 100))
for rep in range(2):
    del LOG[:], EV[:]
    count[0] = 0
    torch.cuda.synchronize()
    on[0] = True
    T0[0] = time.perf_counter()
    e0 = torch.cuda.Event(enable_timing=True)
    e0.record(torch.cuda.current_stream())
    mark('forward >')
    det((fact, scs), 'fact', True)
    mark('forward <')
    torch.cuda.synchronize()
    mark('synchronized')
    on[0] = False
    print('--- host (ms from the start of Detector.forward)')
    for lab, t in LOG:
        print('  %8.2f  %s' % (t, lab))
    print('--- device (ms from the start, events on the stream of the replay)')
    for lab, e in EV:
        print('  %8.2f  %s' % (e0.elapsed_time(e), lab))
