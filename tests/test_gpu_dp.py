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
TINY = dict(V=V, st='tiny', R=R, T=TLEN, B=8, S=4, steps=STEPS, wseed=9)
# BASELINE configs[3] per-rank shape: 128 captions per rank (+ 40 of the 80 seq2seq rows), V = 10k, T = 20, 36 x 2048
FULL = dict(V=10000, st='default', R=36, T=20, B=256, S=80, steps=2, wseed=0)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _settings(cfg):
    return synth.TINY_SETTINGS if cfg['st'] == 'tiny' else synth.DEFAULT_SETTINGS


def _batches(lo, hi, cfg=TINY):
    st = _settings(cfg)
    d = synth.make_inputs(cfg['B'], cfg['V'], st, regions=cfg['R'], seq_len=cfg['T'], seed=31)
    s = synth.make_inputs(cfg['S'], cfg['V'], st, regions=cfg['R'], seq_len=cfg['T'], seed=32)
    t = torch.from_numpy
    fact = (None, t(d['fc_feats'][lo:hi]), t(d['att_feats'][lo:hi]),
            (t(d['captions'][lo:hi]), d['lengths'][lo:hi]), t(d['cpt_words'][lo:hi]))
    labels = t(d['senti_labels'][lo:hi])
    half = cfg['S'] // 2
    s_lo, s_hi = (0, cfg['S']) if (lo, hi) == (0, cfg['B']) else ((0, half) if lo == 0 else (half, cfg['S']))
    scs = ((t(s['captions'][s_lo:s_hi]), s['lengths'][s_lo:s_hi]), t(s['cpt_words'][s_lo:s_hi]),
           t(s['senti_words'][s_lo:s_hi]), t(s['senti_labels'][s_lo:s_hi]))
    return fact, labels, scs


def _make(cfg=TINY):
    st = _settings(cfg)
    cap = Captioner(synth.make_idx2word(cfg['V']), synth.SENTIMENT_CATEGORIES, st)
    cap.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_weights(cfg['V'], st, seed=cfg['wseed']).items()})
    cap.to('cuda:0').eval()            # eval-mode dropout => deterministic, gradients still flow
    return cap


def _run(cap, lo, hi, arena, steps, cfg=TINY, bucketed=True):
    optim, xe_crit, da_crit = cap.get_optim_criterion(4e-4)
    fact, labels, scs = _batches(lo, hi, cfg)
    losses, first_grad = [], None
    for i in range(steps):
        out = xe_train_step(cap, optim, xe_crit, da_crit, fact, labels, scs, 0.0, 0.1, arena=arena, bucketed=bucketed)
        losses.append(float(out['all_loss']))
        if i == 0:
            first_grad = arena.flat.detach().cpu().numpy().copy()     # reduced + clamped gradient of step 1
    return losses, first_grad


def _worker(rank, world, port, results, cfg=TINY, bucketed=True):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK='0')
    dp.init_from_env('gloo')
    cap = _make(cfg)
    dp.broadcast_parameters(cap)
    arena = dp.GradArena(cap.parameters())
    lo, hi = dp.shard(cfg['B'], rank, world)
    losses, g1 = _run(cap, lo, hi, arena, cfg['steps'], cfg, bucketed)
    if rank == 0:                                          # (per step: counts + 4 buckets + losses, or counts + arena + losses)
        assert dp.COLLECTIVES == cfg['steps'] * (6 if bucketed else 3), dp.COLLECTIVES
    torch.cuda.synchronize()
    params = {k: v.detach().cpu().numpy() for k, v in cap.state_dict().items()}
    big = cfg['V'] >= 1000                                 # (spawn pickles cfg: compare by value, not identity)
    if big and rank != 0:                                  # (88 MB per copy through the manager: rank 0's is enough)
        params = {k: float(np.abs(v).sum()) for k, v in params.items()}
    results[rank] = (params, losses, g1 if (not big or rank == 0) else float(np.abs(g1).sum()), arena.nbytes)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_two_rank_training_at_the_config_size_matches_single_process():
    """BASELINE configs[3] per-rank shape: two ranks x (128 captions + 40 seq2seq rows) at V = 10k, T = 20, 36 x 2048
    features - the 88 MB gradient arena under a real two-rank all-reduce (gloo: both ranks share this box's one GPU) -
    against one process on all 256 + 80 rows: losses, the reduced gradient, lock-step parameters after two steps."""
    cfg = FULL
    mgr = mp.get_context('spawn').Manager()
    results = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), results, cfg), nprocs=2, join=True)
    cap = _make(cfg)
    arena = dp.GradArena(cap.parameters())
    ref_losses, ref_g1 = _run(cap, 0, cfg['B'], arena, cfg['steps'], cfg)
    ref = {k: v.detach().cpu().numpy() for k, v in cap.state_dict().items()}
    (p0, l0, g0, nb0), (p1, l1, g1_sum, nb1) = results[0], results[1]
    # the arena of the reference architecture at V = 10k: 88.25 MB of gradients, every view on a 256-byte boundary
    assert nb0 == nb1 and 4 * 22063379 <= nb0 <= 4 * (22063379 + 40 * 64)
    np.testing.assert_allclose(l0, ref_losses, rtol=3e-5)
    np.testing.assert_allclose(l1, l0, rtol=1e-6)
    assert float(np.abs(g0).sum()) == g1_sum                # bit-identical reduced gradient on both ranks
    for (k, v), off in zip(ref.items(), arena.offsets):
        a, b = g0[off:off + v.size], ref_g1[off:off + v.size]
        np.testing.assert_allclose(a, b, atol=2e-4 * np.abs(b).max() + 1e-7, err_msg=k)
        assert float(np.abs(p0[k]).sum()) == p1[k], k      # ranks stay in lock-step
        assert np.abs(p0[k] - ref[k]).max() <= cfg['steps'] * 4e-4 * 1.05, k


def test_two_rank_training_matches_single_process():
    mgr = mp.get_context('spawn').Manager()      # (never fork a process that has initialised the GPU)
    results = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), results), nprocs=2, join=True)
    cap = _make()
    arena = dp.GradArena(cap.parameters())
    ref_losses, ref_g1 = _run(cap, 0, 8, arena, STEPS)
    ref = {k: v.detach().cpu().numpy() for k, v in cap.state_dict().items()}
    (p0, l0, g0, _), (p1, l1, g1, _) = results[0], results[1]
    np.testing.assert_allclose(l0, ref_losses, rtol=2e-5)      # the loss trajectory depends on the updates
    np.testing.assert_allclose(l1, l0, rtol=1e-6)
    # all-reduced gradient of the two half-batches == gradient of the whole batch
    np.testing.assert_array_equal(g0, g1)
    np.testing.assert_allclose(g0, ref_g1, atol=1e-4 * np.abs(ref_g1).max())
    for (k, v), off in zip(ref.items(), arena.offsets):          # and tensor by tensor, relative to each tensor's own scale
        a, b = g0[off:off + v.size], ref_g1[off:off + v.size]
        np.testing.assert_allclose(a, b, atol=2e-4 * np.abs(b).max() + 1e-7, err_msg=k)
    for k in ref:
        np.testing.assert_array_equal(p0[k], p1[k], err_msg=k)          # ranks stay in lock-step
        # Adam normalises by |g|: elements whose gradient is ~eps-sized can move by up to lr per step in
        # either direction depending on rounding order, everything else must agree tightly
        diff = np.abs(p0[k] - ref[k])
        assert diff.max() <= STEPS * 4e-4 * 1.05, k
    assert arena.nbytes >= sum(v.size for v in ref.values()) * 4


def test_bucketed_exchange_equals_the_flat_all_reduce_bit_for_bit():
    """dp.GradSink - four buckets reduced from inside the merged backward, clamp + Adam per bucket behind each reduction -
    against ONE flat all-reduce after the backward (two ranks, gloo, both on this box's GPU): the reduced first-step
    gradient and the parameters after three steps are the same bits, on both ranks."""
    mgr = mp.get_context('spawn').Manager()
    out = {}
    for bucketed in (True, False):
        results = mgr.dict()
        mp.spawn(_worker, args=(2, _free_port(), results, TINY, bucketed), nprocs=2, join=True)
        out[bucketed] = (results[0], results[1])
    for rank in (0, 1):
        (pa, la, ga, _), (pb, lb, gb, _) = out[True][rank], out[False][rank]
        assert la == lb
        np.testing.assert_array_equal(ga, gb)
        for k in pa:
            np.testing.assert_array_equal(pa[k], pb[k], err_msg=k)


def _graph_worker(rank, world, port, results):
    """Two ranks, XE iterations through train_graph.XETrainGraph: two eager iterations on the graph's streams, the capture
    (with the process group live: collectives BETWEEN the two graphs), replays."""
    from insenticap_model_amd.train_graph import XETrainGraph
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK='0')
    dp.init_from_env('gloo')
    cfg = TINY
    cap = _make(cfg)
    dp.broadcast_parameters(cap)
    arena = dp.GradArena(cap.parameters())
    optim, xe_crit, da_crit = cap.get_optim_criterion(4e-4)
    lo, hi = dp.shard(cfg['B'], rank, world)
    fact, labels, scs = _batches(lo, hi, cfg)
    graph = XETrainGraph(cap, optim, xe_crit, da_crit, grad_clip=0.1, arena=arena, warmup=2)
    losses = [float(graph.step(fact, labels, scs, 0.0)['all_loss']) for _ in range(6)]
    torch.cuda.synchronize()
    results[rank] = ({k: v.detach().cpu().numpy() for k, v in cap.state_dict().items()}, losses, graph.replays)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_two_rank_training_graph_with_a_live_process_group_matches_eager():
    """HIP-graph capture next to a multi-rank group (gloo, both ranks on this box's GPU): the forward+backward graph,
    the gradient / normaliser / loss all-reduces between the graphs, the clamp+Adam graph - six iterations, of which at
    least three are replays on every rank - land on the parameters of six single-process eager iterations over the
    whole batch, and the two ranks stay bit-identical."""
    mgr = mp.get_context('spawn').Manager()
    results = mgr.dict()
    mp.spawn(_graph_worker, args=(2, _free_port(), results), nprocs=2, join=True)
    cap = _make()
    arena = dp.GradArena(cap.parameters())
    ref_losses, _ = _run(cap, 0, TINY['B'], arena, 6)
    ref = {k: v.detach().cpu().numpy() for k, v in cap.state_dict().items()}
    (p0, l0, r0), (p1, l1, r1) = results[0], results[1]
    assert r0 >= 3 and r1 >= 3, (r0, r1)
    np.testing.assert_allclose(l0, ref_losses, rtol=5e-5)
    np.testing.assert_allclose(l1, l0, rtol=1e-6)
    for k in ref:
        np.testing.assert_array_equal(p0[k], p1[k], err_msg=k)
        assert np.abs(p0[k] - ref[k]).max() <= 6 * 4e-4 * 1.05, k


# ----------------------------------------------------------------------------- RL step under DP (BASELINE configs[4])
RL_B, RL_S2S = 8, 4


def _rl_data(lo, hi, s_lo, s_hi):
    """Rows [lo, hi) of ONE fact batch (rl_fact layout) and rows [s_lo, s_hi) of one seq2seq batch."""
    st = dict(ST, **synth.HELPER_SETTINGS)
    batches, split = synth.make_rl_batches(1, RL_B, V, st, seq_len=TLEN, seed=70)
    b = batches[0]
    t = torch.from_numpy
    fns = b[0][lo:hi]
    lens = b[3][1][lo:hi]                 # like the collates, a shard's caption tensor ends at ITS longest caption
    item = (fns, t(b[1][lo:hi]), t(b[2][lo:hi]), (t(b[3][0][lo:hi, :max(lens) + 1].copy()), lens), t(b[4][lo:hi]),
            t(b[5][lo:hi]), {fn: b[6][fn] for fn in fns})
    s = synth.make_inputs(RL_S2S, V, ST, regions=R, seq_len=TLEN, seed=72)
    s_lens = s['lengths'][s_lo:s_hi]
    scs = ((t(s['captions'][s_lo:s_hi, :max(s_lens) + 1].copy()), s_lens), t(s['cpt_words'][s_lo:s_hi]),
           t(s['senti_words'][s_lo:s_hi]), t(s['senti_labels'][s_lo:s_hi]))
    draws = np.random.default_rng(71).integers(2, V, size=(RL_B, TLEN), dtype=np.int64)[lo:hi]
    return item, scs, split, draws


def _make_detector():
    from insenticap_model_amd.detector import Detector
    from test_detector import load_helper
    st = dict(ST, **synth.HELPER_SETTINGS)
    st['dropout_p'] = 0.0                  # train mode without dropout: deterministic given the replayed draws
    det = Detector(synth.make_idx2word(V), TLEN, synth.SENTIMENT_CATEGORIES, {'cap_lr': 4e-4}, st)
    det.captioner.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_weights(V, ST, seed=9).items()})
    load_helper(det.senti_detector, 51)
    load_helper(det.sent_senti_cls, 52)
    det.xe_ss_prob = det.seq2seq_ss_prob = 0.0          # scheduled sampling draws on the device: off for parity
    return det.to('cuda:0')


def _run_rl(det, lo, hi, s_lo, s_hi, steps):
    item, scs, split, draws = _rl_data(lo, hi, s_lo, s_hi)
    det.set_ciderd_scorer(split)                        # document frequencies over ALL images, on every rank
    orig = det.captioner.forward_rl
    dev_draws = torch.from_numpy(draws).to('cuda:0')    # once, kept alive: the third iteration is captured into HIP graphs

    def replay_rl(*a, **k):
        if not k.get('sample_max', 1):
            k['_replay'] = dev_draws
        return orig(*a, **k)
    det.captioner.forward_rl = replay_rl
    out, first_grad = [], None
    for i in range(steps):
        out.append(det(([item], [scs]), 'fact', True))
        if i == 0:
            first_grad = det.dp_arena.flat.detach().cpu().numpy().copy()
    return out, first_grad


def _rl_worker(rank, world, port, results):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK='0')
    dp.init_from_env('gloo')
    det = _make_detector()
    det.enable_data_parallel()
    lo, hi = dp.shard(RL_B, rank, world)
    s_lo, s_hi = dp.shard(RL_S2S, rank, world)
    losses, g1 = _run_rl(det, lo, hi, s_lo, s_hi, STEPS)
    torch.cuda.synchronize()
    results[rank] = ({k: v.detach().cpu().numpy() for k, v in det.captioner.state_dict().items()}, losses, g1,
                     det.dp_arena.collectives)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_two_rank_rl_step_matches_single_process():
    """Detector.forward(training=True) under DP: two ranks on half the fact batch and half the seq2seq batch each
    (different token counts and mask sums per rank), multinomial draws replayed, against ONE process on the whole
    batches: the loss dictionaries, the all-reduced gradient and the parameters after 3 steps."""
    mgr = mp.get_context('spawn').Manager()      # (never fork a process that has initialised the GPU)
    results = mgr.dict()
    mp.spawn(_rl_worker, args=(2, _free_port(), results), nprocs=2, join=True)
    det = _make_detector()
    det.enable_data_parallel(broadcast=False)           # no process group here: world 1, flat arena for the compare
    ref_losses, ref_g1 = _run_rl(det, 0, RL_B, 0, RL_S2S, STEPS)
    ref = {k: v.detach().cpu().numpy() for k, v in det.captioner.state_dict().items()}
    (p0, l0, g0, c0), (p1, l1, g1, c1) = results[0], results[1]
    assert c0 == c1 == STEPS                            # one arena all-reduce per iteration
    for a, b, r in zip(l0, l1, ref_losses):
        assert set(a) == set(r) == {'da_loss', 'fact_reward', 'cls_reward', 'all_rewards', 'cap_loss', 'xe_loss',
                                    'seq2seq_loss'}
        for k in r:
            np.testing.assert_allclose(a[k], b[k], rtol=1e-6, atol=1e-7, err_msg=k)        # ranks report global values
            np.testing.assert_allclose(a[k], r[k], rtol=5e-5, atol=2e-6, err_msg=k)
    np.testing.assert_array_equal(g0, g1)
    np.testing.assert_allclose(g0, ref_g1, atol=1e-4 * np.abs(ref_g1).max())
    for (k, v), off in zip(ref.items(), det.dp_arena.offsets):
        a, b = g0[off:off + v.size], ref_g1[off:off + v.size]
        np.testing.assert_allclose(a, b, atol=2e-4 * np.abs(b).max() + 1e-7, err_msg=k)
    for k in ref:
        np.testing.assert_array_equal(p0[k], p1[k], err_msg=k)
        assert np.abs(p0[k] - ref[k]).max() <= STEPS * 4e-4 * 1.05, k


def test_rccl_all_reduce_in_a_fresh_process():
    """The gradient exchange on the backend the 8-GPU runs use: a fresh child process (nothing touched the GPU before
    its init_process_group) builds a ONE-rank "nccl" group - RCCL on ROCm - and runs two xe_train_steps whose arena
    all-reduce goes through it (tests/_rccl_child.py)."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, NCCL_DEBUG='INFO', NCCL_DEBUG_SUBSYS='INIT,COLL', HSA_ENABLE_IPC_MODE_LEGACY='0')
    child = os.path.join(os.path.dirname(os.path.abspath(__file__)), '_rccl_child.py')
    r = subprocess.run([sys.executable, child, str(_free_port())], env=env, capture_output=True, text=True, timeout=420)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    line = [x for x in r.stdout.splitlines() if x.startswith('RCCL_CHILD ')]
    assert line, r.stdout[-3000:]
    info = json.loads(line[-1][len('RCCL_CHILD '):])
    assert info['backend'] == 'nccl' and info['rccl_mapped']
    assert info['collectives'] == 2 * 4 + 1 and info['identity']    # 2 training steps x 4 buckets + the probe
    # per step: normaliser counts + four gradient buckets + loss statistics, all on device tensors through RCCL
    assert info['all_reduces'] == 2 * 6 + 1
    # ... and the same under train_graph.XETrainGraph: 2 eager warm-up steps + 3 replays, three collectives each, the
    # parameters of an eager twin bit for bit
    assert info['graph'] == dict(replays=3, eager=2, all_reduces=5 * 3, arena_collectives=5, equal=True), info['graph']
    assert info['arena_bytes'] >= 4 * sum(int(np.prod(s)) for s in synth.param_shapes(V, ST).values())
    assert info['moved'] >= 30 and all(np.isfinite(x) for x in info['losses'])
    log = r.stdout + r.stderr
    assert 'NCCL INFO' in log or 'RCCL' in log, log[-2000:]        # the backend's own log saw the communicator
