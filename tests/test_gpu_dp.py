"""DP parity on the HIP path (BASELINE.json configs[3] in miniature): two ranks (sharing the one
GPU of the test box, gloo backend) each run xe_train_step on half the batch; the parameters after
k steps must equal a single-process run on the whole batch. Run with pytest -m gpu."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from insenticap_model_amd import Captioner, dp, synth
from insenticap_model_amd.train import xe_train_step

pytestmark = pytest.mark.gpu

V, ST, R, TLEN, STEPS = 64, synth.TINY_SETTINGS, 6, 8, 3


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _batches(lo, hi):
    d = synth.make_inputs(8, V, ST, regions=R, seq_len=TLEN, seed=31)
    s = synth.make_inputs(4, V, ST, regions=R, seq_len=TLEN, seed=32)
    t = torch.from_numpy
    fact = (None, t(d['fc_feats'][lo:hi]), t(d['att_feats'][lo:hi]),
            (t(d['captions'][lo:hi]), d['lengths'][lo:hi]), t(d['cpt_words'][lo:hi]))
    labels = t(d['senti_labels'][lo:hi])
    s_lo, s_hi = (0, 4) if (lo, hi) == (0, 8) else ((0, 2) if lo == 0 else (2, 4))
    scs = ((t(s['captions'][s_lo:s_hi]), s['lengths'][s_lo:s_hi]), t(s['cpt_words'][s_lo:s_hi]),
           t(s['senti_words'][s_lo:s_hi]), t(s['senti_labels'][s_lo:s_hi]))
    return fact, labels, scs


def _make():
    cap = Captioner(synth.make_idx2word(V), synth.SENTIMENT_CATEGORIES, ST)
    cap.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_weights(V, ST, seed=9).items()})
    cap.to('cuda:0').eval()            # eval-mode dropout => deterministic, gradients still flow
    return cap


def _run(cap, lo, hi, arena, steps):
    optim, xe_crit, da_crit = cap.get_optim_criterion(4e-4)
    fact, labels, scs = _batches(lo, hi)
    losses, first_grad = [], None
    for i in range(steps):
        out = xe_train_step(cap, optim, xe_crit, da_crit, fact, labels, scs, 0.0, 0.1, arena=arena)
        losses.append(float(out['all_loss']))
        if i == 0:
            first_grad = arena.flat.detach().cpu().numpy().copy()     # reduced + clamped gradient of step 1
    return losses, first_grad


def _worker(rank, world, port, results):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK='0')
    dp.init_from_env('gloo')
    cap = _make()
    dp.broadcast_parameters(cap)
    arena = dp.GradArena(cap.parameters())
    lo, hi = dp.shard(8, rank, world)
    losses, g1 = _run(cap, lo, hi, arena, STEPS)
    torch.cuda.synchronize()
    results[rank] = ({k: v.detach().cpu().numpy() for k, v in cap.state_dict().items()}, losses, g1)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_two_rank_training_matches_single_process():
    mgr = mp.Manager()
    results = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), results), nprocs=2, join=True)
    cap = _make()
    arena = dp.GradArena(cap.parameters())
    ref_losses, ref_g1 = _run(cap, 0, 8, arena, STEPS)
    ref = {k: v.detach().cpu().numpy() for k, v in cap.state_dict().items()}
    (p0, l0, g0), (p1, l1, g1) = results[0], results[1]
    np.testing.assert_allclose(l0, ref_losses, rtol=2e-5)      # the loss trajectory depends on the updates
    np.testing.assert_allclose(l1, l0, rtol=1e-6)
    # all-reduced gradient of the two half-batches == gradient of the whole batch
    np.testing.assert_array_equal(g0, g1)
    np.testing.assert_allclose(g0, ref_g1, atol=1e-4 * np.abs(ref_g1).max())
    off = 0
    for k, v in ref.items():          # and tensor by tensor, relative to each tensor's own scale
        a, b = g0[off:off + v.size], ref_g1[off:off + v.size]
        np.testing.assert_allclose(a, b, atol=2e-4 * np.abs(b).max() + 1e-7, err_msg=k)
        off += v.size
    for k in ref:
        np.testing.assert_array_equal(p0[k], p1[k], err_msg=k)          # ranks stay in lock-step
        # Adam normalises by |g|: elements whose gradient is ~eps-sized can move by up to lr per step in
        # either direction depending on rounding order, everything else must agree tightly
        diff = np.abs(p0[k] - ref[k])
        assert diff.max() <= STEPS * 4e-4 * 1.05, k
    assert arena.nbytes == sum(v.size for v in ref.values()) * 4
