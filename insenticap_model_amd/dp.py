"""Data-parallel training of the captioner over the GPUs of one node (one process per GPU,
torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU for tests).

The reference is single-device (opts.py:98-99); DP is a capability of this build
(SURVEY 8(e)).  Design for xGMI (point-to-point links, no switch):
  * all 40 gradient tensors live in one flat fp32 arena (22,063,379 elements = 88.25 MB at V=10k), `p.grad` are views
    into it, so autograd accumulates in place and the exchange needs no flatten / unflatten copies: ONE all-reduce of
    the arena after the backward (GradArena.all_reduce; the graph-served steps), or - the eager merged step - four
    buckets whose all-reduces start from inside the backward as each bucket's last dW is enqueued (GradSink), with
    clamp + Adam per bucket behind each reduction.  The small collectives around them (normaliser counts, the loss
    statistics) are asynchronous: the compute stream waits for each only where its result is read.
  * the reduction is a SUM; each rank pre-scales its loss by local_tokens / global_tokens so the
    result equals the single-process loss exactly even when ranks hold different token counts
    (XECriterion / RewardCriterion divide by the *global* mask sum: captioner.py:438, utils.py:175).
  * the elementwise clamp (train_xe.py:19-23) is nonlinear, so it runs after the reduction, fused
    into the Adam launch (optim.FusedClampAdam).
Inference (greedy / beam) shards images across ranks with no collective at all.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise the default process group from torchrun's environment. Returns (rank, world, local)."""
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        kw = {}
        if backend == 'nccl':
            torch.cuda.set_device(local)
            kw['device_id'] = torch.device('cuda', local)
        dist.init_process_group(backend, **kw)
    return rank, world, local


COLLECTIVES = 0          # all-reduces issued by this process through all_reduce_ (tests assert on it)


def world_size(group=None):
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def distributed(group=None):
    """True when a process group exists - ALSO a one-rank group.  The training steps take their data-parallel branches
    (count all-reduce, loss shares, arena all-reduce, statistics all-reduce) whenever this holds, so a one-rank RCCL
    group on a one-GPU box drives exactly the code an 8-rank run does (shares are then 1.0: same numbers)."""
    return dist.is_available() and dist.is_initialized()


def all_reduce_(t, group=None, op=None):
    """The ONE place a collective of this package is issued.  RCCL ("nccl") only moves device tensors: a CPU tensor
    handed to it raises here with the caller's name on the stack instead of "No backend type associated with device
    type cpu" from deep inside c10d (round 2: `global_count` built its count on the host; gloo hid it in every test)."""
    if not distributed(group):
        return t
    if dist.get_backend(group) == 'nccl' and t.device.type != 'cuda':
        raise TypeError('collective on a %s tensor with backend nccl (RCCL): build it on the rank\'s GPU' % t.device.type)
    dist.all_reduce(t, op=dist.ReduceOp.SUM if op is None else op, group=group)
    global COLLECTIVES
    COLLECTIVES += 1
    return t


def all_reduce_async_(t, group=None):
    """all_reduce_ that does not make the current stream wait: returns (t, work); `work.wait()` orders the CURRENT stream
    behind the reduction (the host never blocks under nccl).  work is None without a process group."""
    if not distributed(group):
        return t, None
    if dist.get_backend(group) == 'nccl' and t.device.type != 'cuda':
        raise TypeError('collective on a %s tensor with backend nccl (RCCL): build it on the rank\'s GPU' % t.device.type)
    work = dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=True)
    global COLLECTIVES
    COLLECTIVES += 1
    return t, work


class GradArena:
    """Flat gradient storage: `p.grad` of every parameter is a view into one contiguous buffer.  Every view starts on a
    256-byte boundary (64 floats; 88.26 MB instead of 88.25 at V = 10k): the backward's dW contractions may then write
    a gradient straight into its view (GradSink), which the kernels' 16-byte stores require."""
    ALIGN = 64          # floats

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        ref = self.params[0]
        self.offsets, n = [], 0
        for p in self.params:
            self.offsets.append(n)
            n += (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.flat = torch.zeros(n, dtype=ref.dtype, device=ref.device)
        self.collectives = 0
        for p, off in zip(self.params, self.offsets):
            p.grad = self.flat[off:off + p.numel()].view_as(p)

    def zero_(self):
        """Replaces optimizer.zero_grad(): keeps the views alive (set_to_none would detach them)."""
        self.flat.zero_()
        for p in self.params:
            if p.grad is None or p.grad.data_ptr() < self.flat.data_ptr() or \
                    p.grad.data_ptr() >= self.flat.data_ptr() + self.flat.numel() * self.flat.element_size():
                raise RuntimeError('a .grad was replaced; call zero_() instead of zero_grad(set_to_none=True)')

    def all_reduce(self, group=None):
        """Sum the gradients of all ranks: one collective over the whole arena.  Issued whenever a process group
        exists - also a one-rank group, where it is a no-op arithmetically but still goes through the backend
        (tests/test_gpu_dp.py drives RCCL that way on a one-GPU box)."""
        if distributed(group):
            all_reduce_(self.flat, group)
            self.collectives += 1
        return self.flat

    @property
    def nbytes(self):
        return self.flat.numel() * self.flat.element_size()


class GradSink:
    """Bucketed gradient exchange overlapped with the backward pass (train_xe.py:190-192: the exchange sits between
    `loss.backward()` and `clip_gradient`; the reference is single-device).

    Weights are shared across time steps, so no gradient is final before the reverse sweep ends - but the backward's dW
    contractions AFTER the sweep (and the classifier's, which needs no sweep at all) are ~1.3 ms of a 4.9 ms iteration at
    B = 128 + 80, against ~0.5-1 ms for an 88 MB all-reduce over xGMI.  The merged backward of an iteration
    (autograd_pair._pair_backward) therefore writes each gradient straight into its arena view, finishes the
    parameters bucket by bucket - a bucket = a contiguous run of parameters in arena order - and calls `ready(k)` as
    soon as bucket k's last dW is enqueued: its all-reduce starts on the backend's stream (async_op) while the next
    bucket computes.  Order of completion (BUCKETS below): classifier (before the sweep starts: hidden behind the whole
    sweep), lang-LSTM + attention, att-LSTM + region / word projections, embeddings + fc.  `finish(optim, clip)` then
    waits bucket by bucket and runs clamp + Adam on each as its reduction lands, so the update of bucket k overlaps the
    reduction of bucket k + 1; only the last bucket's reduction (25 MB) is exposed.

    The sums are the same elementwise sums as the flat all-reduce's: parameters after a step are bit-identical
    (tests/test_gpu_dp.py)."""
    # (first parameter-name prefix of each bucket, in ARENA order; the backward completes them last to first)
    STARTS = ('word_embed.', 'att_embed.', 'attention.', 'classifier.')

    def __init__(self, module, arena, group=None, exchange=True):
        self.arena, self.group, self.exchange = arena, group, exchange
        names = [n for n, q in module.named_parameters() if q.requires_grad]
        assert len(names) == len(arena.params)
        self.views = {n: q.grad for n, q in zip(names, arena.params)}
        starts = [i for i, n in enumerate(names) if any(n.startswith(s) and (i == 0 or not names[i - 1].startswith(s))
                                                         for s in self.STARTS)]
        if not starts or starts[0] != 0:
            starts = [0] + starts
        self.buckets = []
        for b, lo in enumerate(starts):
            hi = starts[b + 1] if b + 1 < len(starts) else len(names)
            f_lo = arena.offsets[lo]
            f_hi = arena.offsets[hi] if hi < len(names) else arena.flat.numel()
            self.buckets.append(dict(names=names[lo:hi], params=arena.params[lo:hi], flat=arena.flat[f_lo:f_hi],
                                     work=None, launched=False))
        self.bucket_of = {n: b for b, bk in enumerate(self.buckets) for n in bk['names']}
        self.order = []                 # buckets in the order the backward completed them
        self.collectives = 0

    def out(self, name):
        """The tensor the backward writes parameter `name`'s gradient into."""
        return self.views[name]

    def begin(self):
        self.order = []
        for bk in self.buckets:
            bk['work'], bk['launched'] = None, False

    def ready(self, b, unscale=None):
        """Every gradient of bucket b has been written (by launches enqueued on the current stream).  `unscale`: device
        scalar 1/S of the backward's power-of-two gradient scale, applied to the whole bucket in one launch first."""
        bk = self.buckets[b]
        if bk['launched']:
            return
        if unscale is not None:
            bk['flat'].mul_(unscale)
        bk['launched'] = True
        self.order.append(b)
        if self.exchange and distributed(self.group):
            bk['work'] = dist.all_reduce(bk['flat'], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            global COLLECTIVES
            COLLECTIVES += 1
            self.collectives += 1
            self.arena.collectives += 1

    def wait(self, b):
        w = self.buckets[b]['work']
        if w is not None:
            w.wait()                    # the current stream waits for the backend's stream; the host does not block (nccl)
            self.buckets[b]['work'] = None

    def finish(self, optim, grad_clip):
        """Clamp + Adam bucket by bucket, each behind its own reduction (train_xe.py:191-192)."""
        missing = [b for b in range(len(self.buckets)) if not self.buckets[b]['launched']]
        for b in missing:               # (a backward that did not go through the sink for these: reduce them now)
            self.ready(b)
        first = True
        for i, b in enumerate(self.order):
            self.wait(b)
            optim.step_params(self.buckets[b]['params'], grad_clip, first=first, last=i == len(self.order) - 1)
            first = False


def global_counts(local_counts, device, group=None, asynchronous=False):
    """Sum over ranks of a short list of per-rank normalisers (token counts, mask sums, row counts): ONE small
    all-reduce.  `local_counts`: python numbers and / or 0-dim tensors; the vector is built on `device` - the rank's
    GPU under RCCL.  Returns (local [n], global [n]) float32 tensors on `device`; `asynchronous`: (local, global, work) -
    the current stream does not wait for the reduction until `work.wait()` (work is None without a group)."""
    device = torch.device(device)
    # python numbers travel in ONE pinned buffer by a non-blocking copy: a tensor built from a python list on the device is
    # a pageable H2D copy, i.e. a host wait for everything queued on the stream - once per iteration it serialised the
    # host's enqueueing with the device (XE iteration from graphs under a one-rank group: 6.0 ms against 4.7 without)
    nums = [float(c) for c in local_counts if not torch.is_tensor(c)]
    host = None
    if nums and device.type == 'cuda':
        host = torch.tensor(nums, dtype=torch.float32).pin_memory().to(device, non_blocking=True)
    elif nums:
        host = torch.tensor(nums, dtype=torch.float32, device=device)
    parts, k = [], 0
    for c in local_counts:
        if torch.is_tensor(c):
            parts.append(c.detach().to(device=device, dtype=torch.float32).reshape(1))
        else:
            parts.append(host[k:k + 1])
            k += 1
    local = torch.cat(parts) if len(parts) > 1 else parts[0].clone()
    if asynchronous:
        glob, work = all_reduce_async_(local.clone(), group)
        return local, glob, work
    return local, all_reduce_(local.clone(), group)


def global_count(local_count, group=None, device=None):
    """All-reduce ONE count (python number or 0-dim tensor).  A python number needs `device` (the rank's GPU under
    RCCL; defaults to the CPU, which only gloo accepts - `all_reduce_` says so loudly)."""
    if device is None:
        device = local_count.device if torch.is_tensor(local_count) else 'cpu'
    return global_counts([local_count], device, group)[1][0]


def dp_token_mean(local_mean_loss, local_tokens, group=None):
    """Rescales a per-rank token-mean loss so that SUM-reduced gradients equal those of the global
    token mean: loss_r * n_r / sum_r n_r.  Returns (scaled loss for backward, global token count).  The count lives
    on the loss's device."""
    n = global_count(local_tokens, group, device=local_mean_loss.device)
    lt = local_tokens.to(local_mean_loss.device) if torch.is_tensor(local_tokens) else float(local_tokens)
    return local_mean_loss * (lt / n), n


def dp_batch_mean(local_mean_loss, group=None):
    """Same for a plain batch mean with equal per-rank batch sizes (the domain-align MSE)."""
    return local_mean_loss / world_size(group)


def assert_same_across_ranks(value, device, group=None, what='value'):
    """Every rank must pass the same integer (loop trip counts, key-set digests): a mismatch would otherwise show as a
    hang inside RCCL.  One 2-int all-reduce (MIN of v and of -v)."""
    if not distributed(group):
        return
    t = torch.tensor([int(value), -int(value)], dtype=torch.int64, device=device)
    all_reduce_(t, group, op=dist.ReduceOp.MIN)
    lo, hi = int(t[0]), -int(t[1])
    if lo != hi:
        raise RuntimeError('%s differs across ranks (min %d, max %d, here %d): shard the loaders to equal lengths '
                           '(dp.shard with drop_last)' % (what, lo, hi, int(value)))


def broadcast_parameters(module, src=0, group=None):
    """Make every rank start from rank `src`'s weights (one flat broadcast)."""
    if world_size(group) == 1:
        return
    ps = [p.data for p in module.parameters()]
    flat = torch.cat([p.reshape(-1) for p in ps])
    if dist.get_backend(group) == 'nccl' and flat.device.type != 'cuda':
        raise TypeError('broadcast of CPU parameters with backend nccl (RCCL): move the module to the GPU first')
    dist.broadcast(flat, src=src, group=group)
    off = 0
    for p in ps:
        p.copy_(flat[off:off + p.numel()].view_as(p))
        off += p.numel()
    # written through `.data`: the parameters' version counters did not move - tell the caches keyed on the weight
    # VALUES (f16 weight planes, token tables, captured graphs) that they are stale
    from . import ops
    ops.WEIGHT_EPOCH += 1


def shard(n_items, rank, world, drop_last=False):
    """Contiguous shard [lo, hi) of n_items for this rank (inference: images are independent).  `drop_last`: equal
    shards of n_items // world (training: every rank must run the same number of iterations)."""
    if drop_last:
        per = n_items // world
        return rank * per, (rank + 1) * per
    per = (n_items + world - 1) // world
    lo = min(rank * per, n_items)
    return lo, min(lo + per, n_items)
