"""World-size-2 gloo tests (CPU) of the data-parallel machinery in insenticap_model_amd/dp.py:
flat gradient arena + one all-reduce, token-count-weighted loss normalisation, parameter
broadcast, inference sharding. The N>1 HIP path uses the same code with backend nccl (= RCCL)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from insenticap_model_amd import dp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


class Toy(torch.nn.Module):
    """Tiny stand-in with the same loss structure as XECriterion: a masked token mean."""

    def __init__(self):
        super().__init__()
        g = torch.Generator().manual_seed(0)
        self.w = torch.nn.Parameter(torch.randn(7, 5, generator=g))
        self.b = torch.nn.Parameter(torch.randn(5, generator=g))
        self.unused = torch.nn.Parameter(torch.ones(3))

    def token_nll(self, x, y, mask):
        logp = torch.log_softmax(x @ self.w + self.b, dim=-1)
        nll = -logp.gather(1, y.unsqueeze(1)).squeeze(1) * mask
        return nll.sum() / mask.sum(), mask.sum()


def _data():
    g = torch.Generator().manual_seed(1)
    x = torch.randn(12, 7, generator=g)
    y = torch.randint(0, 5, (12,), generator=g)
    mask = torch.tensor([1, 1, 1, 0, 1, 1, 1, 1, 0, 0, 0, 1], dtype=torch.float32)   # unequal token counts
    return x, y, mask


def _worker(rank, world, port, results):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    r, w, _ = dp.init_from_env('gloo')
    assert (r, w) == (rank, world) and dp.world_size() == world
    torch.manual_seed(100 + rank)                      # ranks start from different weights ...
    model = Toy()
    with torch.no_grad():
        model.w.add_(torch.randn_like(model.w))
    dp.broadcast_parameters(model, src=0)              # ... until the broadcast
    arena = dp.GradArena(model.parameters())
    assert arena.flat.numel() == 3 * dp.GradArena.ALIGN and arena.offsets == [0, 64, 128]    # (256-byte aligned views)
    x, y, mask = _data()
    lo, hi = dp.shard(12, rank, world)
    arena.zero_()
    loss, n_local = model.token_nll(x[lo:hi], y[lo:hi], mask[lo:hi])
    scaled, n_global = dp.dp_token_mean(loss, n_local)
    scaled.backward()
    assert model.w.grad.data_ptr() == arena.flat.data_ptr()         # autograd accumulated in place
    arena.all_reduce()
    total = scaled.detach().clone()
    dist.all_reduce(total)
    results[rank] = dict(w=model.w.detach().numpy().copy(), gw=model.w.grad.numpy().copy(),
                         gb=model.b.grad.numpy().copy(), gu=model.unused.grad.numpy().copy(),
                         loss=float(total), n=float(n_global), shard=(lo, hi))
    dist.barrier()
    dist.destroy_process_group()


def test_dp_two_ranks_match_single_process():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    results = mgr.dict()
    mp.spawn(_worker, args=(world, port, results), nprocs=world, join=True)
    r0, r1 = results[0], results[1]
    # same weights after broadcast, identical reduced gradients on both ranks
    np.testing.assert_array_equal(r0['w'], r1['w'])
    np.testing.assert_allclose(r0['gw'], r1['gw'], atol=0)
    assert r0['shard'] == (0, 6) and r1['shard'] == (6, 12)
    # single-process reference on the full batch with rank 0's weights
    model = Toy()
    with torch.no_grad():
        model.w.copy_(torch.from_numpy(r0['w']))
    x, y, mask = _data()
    loss, n = model.token_nll(x, y, mask)
    loss.backward()
    assert r0['n'] == float(n) == 8.0
    np.testing.assert_allclose(r0['loss'], float(loss.detach()), rtol=1e-6)
    np.testing.assert_allclose(r0['gw'], model.w.grad.numpy(), atol=1e-6)
    np.testing.assert_allclose(r0['gb'], model.b.grad.numpy(), atol=1e-6)
    assert (r0['gu'] == 0).all()          # parameters without a gradient stay zero in the arena


def test_shard_covers_everything_once():
    for n in (0, 1, 7, 64, 1000):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                lo, hi = dp.shard(n, r, world)
                seen.extend(range(lo, hi))
            assert seen == list(range(n))


def test_single_process_helpers_are_identity():
    t = torch.tensor(3.0, requires_grad=True)
    s, n = dp.dp_token_mean(t * 2, 5.0)
    assert float(n) == 5.0 and float(s.detach()) == 6.0
    assert float(dp.dp_batch_mean(t).detach()) == 3.0


def _guard_worker(rank, world, port, results):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    dp.init_from_env('gloo')
    out = {}
    # the group really is gloo; pretend it is RCCL: every collective of the package must then refuse a CPU tensor
    real = dist.get_backend
    dist.get_backend = lambda group=None: 'nccl'
    try:
        for name, fn in (('global_count', lambda: dp.global_count(5.0)),
                         ('global_counts', lambda: dp.global_counts([1.0, 2.0], 'cpu')),
                         ('global_counts_async', lambda: dp.global_counts([1.0, 2.0], 'cpu', asynchronous=True)),
                         ('all_reduce_async', lambda: dp.all_reduce_async_(torch.ones(3))),
                         ('dp_token_mean', lambda: dp.dp_token_mean(torch.tensor(1.0), 3.0)),
                         ('arena', lambda: dp.GradArena(Toy().parameters()).all_reduce()),
                         ('same', lambda: dp.assert_same_across_ranks(3, 'cpu')),
                         ('broadcast', lambda: dp.broadcast_parameters(Toy()))):
            try:
                fn()
                out[name] = 'no error'
            except TypeError as e:
                out[name] = str(e)
    finally:
        dist.get_backend = real
    # and with the true backend the same calls go through (and count)
    before = dp.COLLECTIVES
    n = dp.global_count(5.0 + rank)
    loc, glob = dp.global_counts([1.0, torch.tensor(2.0 * (rank + 1))], 'cpu')
    # the asynchronous forms (the eager data-parallel step's normaliser counts and loss statistics): same sums, read after wait()
    loc_a, glob_a, work = dp.global_counts([1.0, torch.tensor(2.0 * (rank + 1))], 'cpu', asynchronous=True)
    vec, work2 = dp.all_reduce_async_(torch.tensor([1.0 + rank, 10.0]))
    work.wait()
    work2.wait()
    out['async'] = (loc_a.tolist(), glob_a.tolist(), vec.tolist())
    dp.assert_same_across_ranks(7, 'cpu')
    try:
        dp.assert_same_across_ranks(7 + rank, 'cpu', what='trip count')
        mismatch = 'no error'
    except RuntimeError as e:
        mismatch = str(e)
    results[rank] = dict(guard=out, n=float(n), glob=glob.tolist(), loc=loc.tolist(), issued=dp.COLLECTIVES - before,
                         mismatch=mismatch)
    dist.barrier()
    dist.destroy_process_group()


def test_collectives_refuse_cpu_tensors_under_rccl_and_mismatched_ranks_raise():
    """Round-2 defect: `global_count` all-reduced a CPU tensor; gloo accepted it, RCCL would not have.  With the
    backend name mocked to "nccl" every collective entry of dp.py must raise on a CPU tensor (so a count built on the
    host can never reach RCCL), and unequal trip counts across ranks raise instead of hanging."""
    world, port = 2, _free_port()
    results = mp.Manager().dict()
    mp.spawn(_guard_worker, args=(world, port, results), nprocs=world, join=True)
    for r in (0, 1):
        g = results[r]['guard']
        assert g.pop('async') == ([1.0, 2.0 * (r + 1)], [2.0, 6.0], [3.0, 20.0])
        assert set(g) == {'global_count', 'global_counts', 'global_counts_async', 'all_reduce_async', 'dp_token_mean',
                          'arena', 'same', 'broadcast'}
        for k, msg in g.items():
            assert 'nccl' in msg, (k, msg)
        assert results[r]['n'] == 11.0
        assert results[r]['glob'] == [2.0, 6.0] and results[r]['loc'] == [1.0, 2.0 * (r + 1)]
        assert results[r]['issued'] == 6
        assert 'trip count differs across ranks' in results[r]['mismatch']


def test_shard_drop_last_gives_equal_shards():
    for n in (0, 7, 64, 1001):
        for world in (1, 2, 8):
            spans = [dp.shard(n, r, world, drop_last=True) for r in range(world)]
            assert len({hi - lo for lo, hi in spans}) == 1
            assert spans[0][0] == 0 and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert spans[-1][1] == (n // world) * world
