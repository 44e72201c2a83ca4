"""include/insenticap_hip.h promises that entry points may be called from several host threads on distinct streams.
Two threads, two streams, two captioners with different weights, B=1024 greedy roll-outs (split-f16 classifier,
weights scope, step plans, split-K workspace - everything that used to be process-wide state is per stream): each
thread's results must be bit-identical to the same roll-out run alone.  pytest -m gpu."""
import threading

import numpy as np
import pytest
import torch

from insenticap_model_amd import Captioner, ops, synth

pytestmark = pytest.mark.gpu
KEYS = ('fc_feats', 'att_feats', 'cpt_words', 'senti_words', 'senti_labels')


def _make(seed, dev):
    st, V = synth.DEFAULT_SETTINGS, 10000
    cap = Captioner(synth.make_idx2word(V), synth.SENTIMENT_CATEGORIES, st)
    cap.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_weights(V, st, seed=seed).items()})
    return cap.to(dev).eval()


def test_two_host_threads_on_two_streams_do_not_disturb_each_other():
    dev = torch.device('cuda:0')
    B, Tn, reps = 1024, 20, 3
    caps = [_make(0, dev), _make(7, dev)]
    ins = []
    for i in range(2):
        d = synth.make_inputs(B, 10000, synth.DEFAULT_SETTINGS, regions=36, seq_len=Tn, seed=900 + i)
        ins.append([torch.from_numpy(np.asarray(d[k])).to(dev) for k in KEYS])
    alone = []
    for i in range(2):                                    # reference runs, one after the other on the default stream
        with torch.no_grad():
            alone.append([x.cpu() for x in caps[i](*ins[i], Tn, 1, mode='rl')])
    torch.cuda.synchronize()
    h3_before = ops._lib.load().isc_h3_launches()
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    out, errors = [None, None], []
    barrier = threading.Barrier(2)

    def worker(i):
        try:
            torch.cuda.set_device(dev)
            with torch.cuda.stream(streams[i]), torch.no_grad():
                for x in ins[i]:
                    x.record_stream(streams[i])
                barrier.wait()
                res = []
                for _ in range(reps):                     # enqueue-only loops: the two threads interleave their calls
                    res.append(caps[i](*ins[i], Tn, 1, mode='rl'))
                streams[i].synchronize()
                out[i] = [[x.cpu() for x in r] for r in res]
        except Exception as e:                            # surfaced in the main thread
            errors.append((i, repr(e)))
            try:
                barrier.abort()
            except Exception:
                pass
    torch.cuda.current_stream(dev).synchronize()
    for s in streams:
        s.wait_stream(torch.cuda.current_stream(dev))
    threads = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert ops._lib.load().isc_h3_launches() - h3_before >= 2 * reps * Tn      # the split-f16 path was in play
    for i in range(2):
        for r in out[i]:
            for got, ref in zip(r, alone[i]):
                assert torch.equal(got, ref), i
    assert not torch.equal(alone[0][0], alone[1][0])      # the two captioners really decode different things
