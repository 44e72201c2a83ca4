"""The XE training iteration (train_xe.py:149-192) served from HIP graphs.

At the reference's batch sizes an iteration is ~670 launches of 4-40 us each: on MI355X the device finishes them as
fast as the host can enqueue them (7.1 ms of kernel time against 7.8 ms of host time per iteration at B = 128 + 80
seq2seq rows), so whatever a kernel saves is not seen until the host is out of the loop.  `XETrainGraph.step` runs the
same phases as `train.xe_train_step` - (1) both unrolls, losses, backward; (2) the data-parallel exchange;
(3) clamp + Adam + re-split of the weight planes - with (1) and (3) captured once per input geometry and replayed (one
graph without a process group, two with the exchange between them):

    graph = XETrainGraph(captioner, optim, xe_crit, da_crit, grad_clip=0.1)
    for batch ...:  losses = graph.step(fact_batch, senti_labels, scs_batch, ss_prob)

What makes the capture sound:
  * private streams: the capture stream and the seq2seq side stream belong to this object, so everything this library
    keeps per stream (split-K workspace, f16 weight-plane buffer, weights-scope slot) is private to its graphs;
  * two long branches: the seq2seq unroll - forward AND backward - forks off on the side stream before the XE unroll
    starts and joins after it.  Its gradients go to tensors of their own (p.grad is swapped out around its backward)
    and are added in one multi-tensor launch after the join: the same two-operand sums autograd's accumulation forms
    in the eager step, bit for bit.  (The eager step forks the seq2seq FORWARD only after the XE forward is queued and
    lets autograd interleave the backward sweeps: replayed as a graph that hid 1.1 of the branch's 2.9 ms; the two
    full-length branches hide nearly all of it - 5.8 -> 4.9 ms per iteration at B = 128 + 80.  As two separately
    launched graphs on the two streams the same work once died with a GPU memory fault on a second replay that could
    not be reproduced or attributed; one graph with two branches ran 1500 iterations in a row and every test.)
  * static inputs: every batch is copied into fixed device buffers first (caption lengths included: no host copy sits
    inside a graph); a new geometry (batch size, caption length, ss_prob, train / eval mode) gets its own graph after
    `warmup` eager iterations on the same streams;
  * step-dependent scalars: Adam's lr and bias corrections live in three device floats rewritten before each replay
    (isc_clamp_adam_hyper), computed exactly as the eager launch derives them, so graph and eager steps are
    bit-identical; the optimizer's `state[...]['step']` counters advance per replay (state_dict layout unchanged);
  * weight planes: the captured iteration ends with the refresh of its own streams' planes, so each replay leaves the
    planes the next one starts from.  If anything else changed the weights in between (load_state_dict, an eager step,
    another optimizer) - seen from `Captioner._weights_key()` - the iteration runs eagerly once (rebuilding the planes)
    and, because a rebuilt scope may lay its planes out anew, every graph is captured again afterwards;
  * no collective inside a graph: under a process group phase (2) runs between the two graphs on the same stream;
  * gradient accumulation on the graph's streams: autograd runs a parameter's AccumulateGrad node on the stream that
    was current when the node was CREATED, and the node lives as long as any graph piece that reaches it -
    `captioner.cpt_feats` / `.fc_feats` of an earlier eager iteration do (they are outputs of the decode node, whose
    inputs are all parameters).  `step` drops those two attributes first, so the nodes are created anew under this
    object's stream; a caller that keeps other graph-attached results of an eager iteration alive (`pred`, a loss
    that was not detached) must drop them before the first `step`, or the capture fails with
    hipErrorStreamCaptureUnjoined / ...Implicit (work on a stream outside the capture).
"""
import collections
import weakref

import torch

from . import dp, ops
from .optim import FusedClampAdam
from .train import dp_shares, loss_dict, xe_forward_backward, xe_update


class _Geometry:
    """Static buffers + the captured graphs of one input geometry."""

    def __init__(self):
        self.inputs = None          # dict name -> static device tensor
        self.eager_runs = 0
        self.g_iter = self.g_up = None
        self.keep = None            # tensors the graphs read across each other (gradient lists, losses)
        self.vec = None             # static [xe, da, seq2seq] losses (local values)
        self.layout = None          # weights-scope cold-begin count the graphs were captured under


class XETrainGraph:
    def __init__(self, captioner, optim, xe_crit, da_crit, grad_clip=0.1, arena=None, group=None, warmup=2,
                 max_geometries=4):
        if not isinstance(optim, FusedClampAdam):
            raise TypeError('XETrainGraph needs the fused optimizer (Captioner.get_optim_criterion)')
        if len(optim.param_groups) != 1:
            raise ValueError('one parameter group expected (captioner.py:422)')
        self.cap, self.optim, self.xe_crit, self.da_crit = captioner, optim, xe_crit, da_crit
        self.grad_clip, self.arena, self.group, self.warmup = grad_clip, arena, group, max(1, int(warmup))
        self.device = next(captioner.parameters()).device
        ops.require_device(*captioner.parameters())
        # (default priority on purpose: on high-priority streams the same replays took 18.3 instead of 5.8 ms)
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.stream = ops.private_stream(self.device)
        self.side = ops.private_stream(self.device)
        self._scope_keys = ((idx, self.stream.cuda_stream), (idx, self.side.cuda_stream))
        self._handles = (self.stream.cuda_stream, self.side.cuda_stream)
        self.hyper = torch.zeros(3, dtype=torch.float32, device=self.device)
        self._params = [q for q in captioner.parameters() if q.requires_grad]
        self.shares = torch.ones(3, dtype=torch.float32, device=self.device)
        self._geoms = collections.OrderedDict()
        self._max_geoms = max_geometries
        self._valid_key = None       # Captioner._weights_key() after the last step this object ran
        self.replays = self.eager_steps = self.captures = 0
        # (ops.private_stream: no other owner in this package holds these two.)  The state the library keeps per stream
        # is dropped with this object, so a later owner of the same handles starts clean (and 2 x 320 MB of workspace
        # do not outlive the graphs that used them).
        self._finalizer = weakref.finalize(self, XETrainGraph._release, self._scope_keys)

    @staticmethod
    def _release(keys):
        for index, handle in keys:
            try:
                ops.release_stream_state(index, handle)
            except Exception:           # interpreter shutdown: the library may be gone
                pass

    def close(self):
        """Drop the graphs and this object's per-stream state now (also happens when the object is collected)."""
        self._geoms.clear()
        torch.cuda.synchronize(self.device)
        self._finalizer()

    # ------------------------------------------------------------------ helpers
    def _dist(self):
        return dp.distributed(self.group)

    @staticmethod
    def _as_list(lengths):
        return [int(x) for x in (lengths.tolist() if isinstance(lengths, torch.Tensor) else lengths)]

    def _signature(self, t, ss_prob):
        return (tuple((k, tuple(v.shape), v.dtype) for k, v in sorted(t.items())), float(ss_prob),
                bool(self.cap.training), self._dist())

    def _stage(self, geo, t):
        if geo.inputs is None:
            geo.inputs = {k: torch.empty(v.shape, dtype=v.dtype, device=self.device) for k, v in t.items()}
        for k, v in t.items():
            if not v.is_cuda and not v.is_pinned():
                v = v.pin_memory()
            geo.inputs[k].copy_(v, non_blocking=True)

    def _phase_args(self, geo):
        i = geo.inputs
        fact = (i['fc'], i['att'], i['caps'], i['len'], i['cpts'])
        scs = (i['s_caps'], i['s_len'], i['s_cpts'], i['s_sentis'], i['s_labels']) if 's_caps' in i else None
        return fact, i['labels'], scs

    # ---- the phases of an iteration (run eagerly while a geometry warms up, captured afterwards) ----------------
    def _phase_xe(self, geo, ss_prob):
        """XE unroll + domain-align loss, forward and backward, on self.stream; gradients land in p.grad (the arena's
        views under DP).  Returns the detached [xe, da, 0] losses."""
        fact, labels, _ = self._phase_args(geo)
        self.cap.cpt_feats = self.cap.fc_feats = None
        return xe_forward_backward(self.cap, self.optim, self.xe_crit, self.da_crit, fact, labels, None, ss_prob,
                                   self.arena, self.shares if self._dist() else None, False, None)

    def _phase_s2s(self, geo, ss_prob):
        """seq2seq unroll, forward and backward, on self.side - concurrently with _phase_xe, so its gradients go to
        tensors of their own (p.grad is swapped out around the backward) and are added afterwards (_phase_add): the
        same two-operand sums autograd's accumulation forms in the eager step.  Returns (loss, gradient list)."""
        from .train import _xe_loss
        _, _, scs = self._phase_args(geo)
        s_caps, s_len, s_cpts, s_sentis, s_labels = scs
        params = self._params
        self.cap.cpt_feats = self.cap.fc_feats = None          # the accumulate nodes are re-made under THIS stream
        pred2 = self.cap(s_caps, s_cpts, s_sentis, s_labels, ss_prob, mode='seq2seq')
        loss = _xe_loss(self.xe_crit, pred2, s_caps[:, 1:], s_len)
        if self._dist():
            loss = loss * self.shares[1]
        main = [q.grad for q in params]
        for q in params:
            q.grad = None
        try:
            loss.backward()
            own = [q.grad for q in params]
        finally:
            for q, g in zip(params, main):
                q.grad = g
        self.cap.cpt_feats = None
        return loss.detach(), own

    def _phase_add(self, geo, vec_xe, s2s):
        """p.grad += the seq2seq unroll's gradients (one multi-tensor launch); the loss vector."""
        if s2s is None:
            return vec_xe
        loss, own = s2s
        dst, src = [], []
        for q, g in zip(self._params, own):
            if g is None:
                continue
            if q.grad is None:
                q.grad = g                                      # a parameter only the seq2seq unroll reaches
            else:
                dst.append(q.grad)
                src.append(g)
        if dst:
            torch._foreach_add_(dst, src)
        return torch.stack([vec_xe[0], vec_xe[1], loss])

    def _exchange(self, vec):
        if self.arena is not None:
            self.arena.all_reduce(self.group)
        if self._dist():
            vec = dp.all_reduce_(vec.clone(), self.group)
        return vec

    def _set_hyper(self, step_no):
        g = self.optim.param_groups[0]
        b1, b2 = g['betas']
        h = torch.tensor(ops.adam_hyper(g['lr'], b1, b2, step_no), dtype=torch.float32).pin_memory()
        self.hyper.copy_(h, non_blocking=True)

    def _states(self):
        return [self.optim.state[q] for q in self.optim.param_groups[0]['params'] if q in self.optim.state and
                len(self.optim.state[q])]

    def _capture(self, geo, ss_prob):
        """The graph(s) of this geometry, scopes warm: the seq2seq unroll as a branch on self.side around the XE unroll
        on self.stream, then add + update (a second graph under a process group: the exchange sits in front of the
        update).  Capturing enqueues nothing:
        the optimizer's step counters and the weight epoch the host logic advanced are taken back / carried by the
        first replay."""
        steps_before = [float(st['step']) for st in self._states()]
        self.optim.device_hyper = self.hyper
        has_s2s = 's_caps' in geo.inputs
        try:
            with ops.refresh_only(self._handles):
                # ONE graph, two long branches: the seq2seq unroll (forward AND backward) forks off on self.side before
                # the XE unroll starts on self.stream and joins after it
                geo.g_iter = torch.cuda.CUDAGraph()
                with ops.graph_capture(geo.g_iter, stream=self.stream):
                    s2s = None
                    if has_s2s:
                        self.side.wait_stream(self.stream)
                        with torch.cuda.stream(self.side):
                            s2s = self._phase_s2s(geo, ss_prob)
                    vec_xe = self._phase_xe(geo, ss_prob)
                    if has_s2s:
                        self.stream.wait_stream(self.side)
                    geo.vec = self._phase_add(geo, vec_xe, s2s)
                    if not self._dist():
                        xe_update(self.optim, self.grad_clip)
                geo.g_up = None
                if self._dist():
                    geo.g_up = torch.cuda.CUDAGraph()
                    with ops.graph_capture(geo.g_up, stream=self.stream):
                        xe_update(self.optim, self.grad_clip)
                geo.keep = (vec_xe, s2s)
        finally:
            self.optim.device_hyper = None
        for st, n in zip(self._states(), steps_before):
            st['step'].fill_(n)
        geo.layout = ops.h3_weights_scope.cold_begins(self._scope_keys)
        self._valid_key = self.cap._weights_key()
        self.captures += 1

    def _replay(self, geo):
        states = self._states()
        for st in states:
            st['step'] += 1
        self._set_hyper(int(states[0]['step']))
        geo.g_iter.replay()
        vec = geo.vec
        if geo.g_up is not None:
            vec = self._exchange(vec)
            geo.g_up.replay()
        # what the captured host logic did once, per replay: the weights moved behind torch's back ...
        epoch_before = ops.WEIGHT_EPOCH
        ops.WEIGHT_EPOCH += 1
        ops.h3_weights_scope.rekey_epoch(self._scope_keys, epoch_before)      # ... and OUR planes were refreshed
        self._valid_key = self.cap._weights_key()
        self.replays += 1
        return vec

    def _eager(self, geo, ss_prob):
        """The same phases, not captured (a geometry's first iterations; any iteration after something else moved the
        weights): same streams, same order of dependencies."""
        with ops.refresh_only(self._handles):
            has_s2s = 's_caps' in geo.inputs
            if has_s2s:
                self.side.wait_stream(self.stream)
            vec_xe = self._phase_xe(geo, ss_prob)
            s2s = None
            if has_s2s:
                with torch.cuda.stream(self.side):
                    s2s = self._phase_s2s(geo, ss_prob)
                self.stream.wait_stream(self.side)
                for g in s2s[1]:
                    if g is not None:
                        g.record_stream(self.stream)
            vec = self._phase_add(geo, vec_xe, s2s)
            vec = self._exchange(vec)
            xe_update(self.optim, self.grad_clip)
        self._valid_key = self.cap._weights_key()
        self.eager_steps += 1
        return vec

    # ------------------------------------------------------------------ the step
    def step(self, fact_batch, xe_senti_labels, scs_batch=None, ss_prob=0.0):
        """One iteration on the batch tuples of train.xe_train_step; returns its loss dictionary (0-dim device
        tensors, global values under DP), valid on the caller's current stream."""
        _, fc, att, (caps, lengths), cpts = fact_batch[:5]
        lengths = self._as_list(lengths)
        if caps.size(1) - 1 != max(lengths):
            raise ValueError('captions are %d tokens wide, max(lengths)=%d (+1 for <SOS>)' % (caps.size(1), max(lengths)))
        t = dict(fc=fc, att=att, caps=caps, cpts=cpts, labels=xe_senti_labels,
                 len=torch.tensor(lengths, dtype=torch.int32))
        s_lengths = None
        if scs_batch is not None:
            (s_caps, s_lengths), s_cpts, s_sentis, s_labels = scs_batch
            s_lengths = self._as_list(s_lengths)
            if s_caps.size(1) - 1 != max(s_lengths):
                raise ValueError('seq2seq captions are %d tokens wide, max(lengths)=%d' % (s_caps.size(1), max(s_lengths)))
            t.update(s_caps=s_caps, s_cpts=s_cpts, s_sentis=s_sentis, s_labels=s_labels,
                     s_len=torch.tensor(s_lengths, dtype=torch.int32))
        self.cap.cpt_feats = self.cap.fc_feats = None      # (see the module docstring: stale AccumulateGrad nodes)
        sig = self._signature(t, ss_prob)
        geo = self._geoms.get(sig)
        if geo is None:
            geo = self._geoms[sig] = _Geometry()
            while len(self._geoms) > self._max_geoms:
                self._geoms.popitem(last=False)
        else:
            self._geoms.move_to_end(sig)
        caller = torch.cuda.current_stream(self.device)
        self.stream.wait_stream(caller)
        with torch.cuda.stream(self.stream):
            self._stage(geo, t)
            if self._dist():
                self.shares.copy_(dp_shares(lengths, s_lengths, fc.shape[0], self.device, self.group))
            planes_ok = self._valid_key is not None and self._valid_key == self.cap._weights_key()
            if ops.h3_weights_scope.cold_begins(self._scope_keys) != geo.layout:
                geo.g_iter = geo.g_up = None                 # a scope of ours was rebuilt: addresses may differ
            if geo.g_iter is None and planes_ok and geo.eager_runs >= self.warmup:
                self._capture(geo, ss_prob)
            if geo.g_iter is not None and planes_ok:
                vec = self._replay(geo)
            else:
                vec = self._eager(geo, ss_prob)
                geo.eager_runs += 1
            out = loss_dict(vec.clone())
        caller.wait_stream(self.stream)
        for v in out.values():
            v.record_stream(caller)
        return out
