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
    KEEP_GRAPHS = False          # True (tests, tools): graphs keep their hipGraph_t so that ops.graph_node_counts can walk it

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

    def _claim(self):
        """One training-graph object drives a captioner at a time.  The graphs of this class keep the autograd graph of their
        captured forward passes alive between replays, and with it the parameters' AccumulateGrad nodes - which run on the
        stream they were created under (the module docstring's last point).  Another object (an XETrainGraph next to a
        Detector's RLTrainGraph, say) that captured a backward reaching those nodes would pull a stream of the first
        object into its capture: the runtime's end-of-capture then crashes (seen: 600 RL iterations, then the first XE
        capture).  So the object that steps takes the captioner over: the previous owner's captured graphs are dropped
        (it captures again when it steps next - two eager iterations per geometry)."""
        import gc
        ref = self.cap.__dict__.get('_train_graph_owner')
        other = ref() if ref is not None else None
        if other is not None and other is not self and other._geoms:
            torch.cuda.synchronize(self.device)         # its replays have finished before their graphs go
            other._geoms.clear()
            other._valid_key = None
            self.cap.cpt_feats = self.cap.fc_feats = self.cap.s2s_cpt_feats = None
            gc.collect()                                # (autograd graphs sit in reference cycles with the geometry objects)
        if other is not self:
            self.cap.__dict__['_train_graph_owner'] = weakref.ref(self)

    def _evicted(self, geo):
        """A captured geometry fell out of the `max_geometries` this object keeps.  Real batches differ in their longest
        caption; padded to that length each width is a geometry of its own, and a trainer that cycles through more widths
        than are kept pays two eager iterations + a capture every few steps - say so once."""
        if geo.g_iter is None:
            return
        self.evictions = getattr(self, 'evictions', 0) + 1
        if self.evictions == 8:
            import warnings
            warnings.warn('insenticap_model_amd: 8 captured training geometries evicted - the batch geometry keeps changing '
                          '(a geometry = batch size x padded caption length). Pad captions to a fixed width '
                          '(data.create_collate_fn(caption_width="full" or a multiple)) or raise max_geometries.')

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
                bool(self.cap.training), self._dist(), bool(self._pair()))

    def _stage(self, geo, t):
        """The batch -> the geometry's static device buffers.  Device tensors go in ONE multi-copy launch per eight
        (isc_copy_multi); host tensors (the caption lengths, host-resident batches) are packed into ONE pinned buffer of
        the geometry and travel in one non-blocking copy + one multi-copy - an iteration used to start with 16 separate
        copies, each a host round trip in front of the replay (0.45 ms of a 5 ms iteration)."""
        if geo.inputs is None:
            geo.inputs = {k: torch.empty(v.shape, dtype=v.dtype, device=self.device) for k, v in t.items()}
            geo.pin = None
        dev_d, dev_s, host = [], [], []
        for k, v in t.items():
            dst = geo.inputs[k]
            if v.device.type == 'meta':          # a late input (RLTrainGraph._late_inputs): shape and dtype only
                continue
            if v.is_cuda and v.dtype == dst.dtype and v.is_contiguous():
                dev_d.append(dst)
                dev_s.append(v)
            elif v.is_cuda:
                dst.copy_(v, non_blocking=True)
            else:
                host.append((dst, v))
        if host:
            sizes = [(d.numel() * d.element_size() + 15) & ~15 for d, _ in host]
            total = sum(sizes)
            if geo.pin is None or geo.pin[0].numel() != total:
                geo.pin = (torch.empty(total, dtype=torch.uint8).pin_memory(),
                           torch.empty(total, dtype=torch.uint8, device=self.device), torch.cuda.Event())
                geo.pin[2].record(torch.cuda.current_stream(self.device))
            pin, stage, ev = geo.pin
            ev.synchronize()                 # the previous iteration's H2D copy has read the pinned buffer
            off = 0
            for (d, v), n in zip(host, sizes):
                nb = d.numel() * d.element_size()
                pin[off:off + nb].view(d.dtype).view(d.shape).copy_(v.to(d.dtype))
                dev_d.append(d.view(-1).view(torch.uint8))
                dev_s.append(stage[off:off + nb])
                off += n
            stage.copy_(pin, non_blocking=True)
            ev.record(torch.cuda.current_stream(self.device))
        if dev_d:
            ops.copy_multi([d.view(-1).view(torch.uint8) if d.dtype != torch.uint8 else d.view(-1) for d in dev_d],
                           [s.view(-1).view(torch.uint8) if s.dtype != torch.uint8 else s.view(-1) for s in dev_s])

    def _phase_args(self, geo):
        i = geo.inputs
        fact = (i['fc'], i['att'], i['caps'], i['len'], i['cpts']) if 'caps' in i else None     # (RL 'senti' iterations: no captions)
        scs = (i['s_caps'], i['s_len'], i['s_cpts'], i['s_sentis'], i['s_labels']) if 's_caps' in i else None
        return fact, i['labels'], scs

    # ---- the phases of an iteration (run eagerly while a geometry warms up, captured afterwards) ----------------
    def _pair(self):
        """Both unrolls through one step chain on self.stream (autograd_pair: 367 nodes per B = 128 + 80 iteration, 5.3 ms)
        or - the default inside graphs, 4.85 ms - the seq2seq unroll as a branch on self.side with its gradients added
        afterwards (autograd_pair.use_pair; `captioner.pair_unrolls` forces either)."""
        from .autograd_pair import use_pair
        return use_pair(self.cap, True)

    def _phase_xe(self, geo, ss_prob, with_scs=False):
        """XE unroll + domain-align loss (with_scs: and the seq2seq unroll, merged into the same step chain), forward
        and backward, on self.stream; gradients land in p.grad (the arena's views under DP).  Returns the detached
        [xe, da, seq2seq or 0] losses."""
        fact, labels, scs = self._phase_args(geo)
        self.cap.cpt_feats = self.cap.fc_feats = self.cap.s2s_cpt_feats = None
        return xe_forward_backward(self.cap, self.optim, self.xe_crit, self.da_crit, fact, labels,
                                   scs if with_scs else None, ss_prob, self.arena,
                                   self.shares if self._dist() else None, False, None, pair=with_scs)

    def _phase_s2s(self, geo, ss_prob):
        """seq2seq unroll, forward and backward, on self.side - concurrently with _phase_xe, so its gradients go to
        tensors of their own and are added afterwards (_phase_add): the same two-operand sums autograd's accumulation
        forms in the eager step.  Returns (loss, gradient list)."""
        from .train import _xe_loss
        _, _, scs = self._phase_args(geo)
        s_caps, s_len, s_cpts, s_sentis, s_labels = scs
        params = self._params
        self.cap.cpt_feats = self.cap.fc_feats = self.cap.s2s_cpt_feats = None          # the accumulate nodes are re-made under THIS stream
        with self.cap.token_logprobs():
            pred2 = self.cap(s_caps, s_cpts, s_sentis, s_labels, ss_prob, mode='seq2seq')
        loss = _xe_loss(self.xe_crit, pred2, s_caps[:, 1:], s_len)
        if self._dist():
            loss = loss * self.shares[1]
        # (autograd.grad, not backward(): the gradients come back as tensors, no parameter's .grad or accumulation node is
        # touched - those belong to the stream of the main backward, which may not be part of the capture this runs in)
        own = list(torch.autograd.grad(loss, params, allow_unused=True))
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
                geo.g_iter = torch.cuda.CUDAGraph(keep_graph=self.KEEP_GRAPHS)
                with ops.graph_capture(geo.g_iter, stream=self.stream):
                    s2s = None
                    if has_s2s and self._pair():
                        vec_xe = geo.vec = self._phase_xe(geo, ss_prob, True)
                    else:
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
                    geo.g_up = torch.cuda.CUDAGraph(keep_graph=self.KEEP_GRAPHS)
                    with ops.graph_capture(geo.g_up, stream=self.stream, settle=False):
                        xe_update(self.optim, self.grad_clip)
                geo.keep = (vec_xe, s2s)
        except BaseException:
            geo.g_iter = geo.g_up = None        # a half-captured graph must not read as "captured" to the next step
            raise
        finally:
            self.optim.device_hyper = None
            for st, n in zip(self._states(), steps_before):      # (also when the capture fails half-way)
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
            if has_s2s and self._pair():
                vec = self._phase_xe(geo, ss_prob, True)
            else:
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
        # (wider is fine - the criterion masks by row: captions padded to a fixed width keep ONE geometry, i.e. one captured
        # graph, for batches whose longest caption varies: data.create_collate_fn(caption_width=...))
        if caps.size(1) - 1 < max(lengths):
            raise ValueError('captions are %d tokens wide, max(lengths)=%d (+1 for <SOS>)' % (caps.size(1), max(lengths)))
        t = dict(fc=fc, att=att, caps=caps, cpts=cpts, labels=xe_senti_labels,
                 len=torch.tensor(lengths, dtype=torch.int32))
        s_lengths = None
        if scs_batch is not None:
            (s_caps, s_lengths), s_cpts, s_sentis, s_labels = scs_batch
            s_lengths = self._as_list(s_lengths)
            if s_caps.size(1) - 1 < max(s_lengths):
                raise ValueError('seq2seq captions are %d tokens wide, max(lengths)=%d' % (s_caps.size(1), max(s_lengths)))
            t.update(s_caps=s_caps, s_cpts=s_cpts, s_sentis=s_sentis, s_labels=s_labels,
                     s_len=torch.tensor(s_lengths, dtype=torch.int32))
        self._claim()
        self.cap.cpt_feats = self.cap.fc_feats = self.cap.s2s_cpt_feats = None      # (see the module docstring: stale AccumulateGrad nodes)
        sig = self._signature(t, ss_prob)
        geo = self._geoms.get(sig)
        if geo is None:
            geo = self._geoms[sig] = _Geometry()
            while len(self._geoms) > self._max_geoms:
                self._evicted(self._geoms.popitem(last=False)[1])
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


class _RLGeometry(_Geometry):
    def __init__(self):
        super().__init__()
        self.g_roll = self.g_greedy = self.g_fwd = self.g_bwd = None     # (g_up: the update graph under a process group)
        self.pool = None
        self.host = None            # pinned host copies of the two token matrices
        self.reward = None          # static [B, T] reward the REINFORCE loss reads
        self.late = {}              # inputs handed over as callables, still to be produced this iteration
        self.stats = None           # static [7] device statistics of the iteration


class RLTrainGraph(XETrainGraph):
    """The self-critical RL training iteration of `Detector.forward(data, 'fact', True)` (models/decoder.py:65-167) - and
    of `Detector.forward(data, 'senti', True)`, the other half of the reference's RL epochs (train_rl.py:232-235: images
    with sentiment labels, no captions: the same phases without the XE unroll and without the CIDEr-D reward) -
    served from HIP graphs - four of them per input geometry, with the two things that cannot live in a graph between:

        g_roll   sampled roll-out (activations kept for REINFORCE), domain-align loss, the device->host copy of its token
                 matrix and lengths                                                                [decoder.py:85-91]
        g_greedy greedy baseline roll-out, the copy of its token matrix - on a stream of its own, next to g_roll and g_fwd
                 (it draws no random numbers and writes nothing they read)                        [decoder.py:93-98]
        eager    the classifier reward on the side stream, behind g_roll (a frozen helper net: an LSTM of the stock library;
                 lengths stay on the device, so its launches are enqueued at once)                [utils.py:120-151]
        host     scores CIDEr-D of the sampled captions (host library) as soon as their tokens have landed, while the device
                 decodes the greedy ones and runs g_fwd; then theirs
        g_fwd    XE unroll (ss_prob 0.5), forward, on a stream of its own NEXT TO the roll-outs (it reads the batch and the
                 weights only); seq2seq unroll (ss_prob 0.25) forward AND backward as a branch on the side stream, its
                 gradients in tensors of their own                                                [decoder.py:131-158]
        eager    rewards -> the static reward buffer
        g_bwd    RewardCriterion, the sum of the losses, backward (two branches: the roll-out's reverse sweep and the XE
                 unroll's, each on the stream of its forward), the seq2seq gradients added, clamp + Adam + plane
                 refresh (under a process group: the exchange, then the update as a graph of its own) [decoder.py:126-167]

    Streams, weight-plane scopes, device-side Adam scalars, step counters, the run-eagerly-once-after-a-foreign-weight-
    change rule and the re-capture after a scope rebuild are XETrainGraph's.  The phases run eagerly (same streams, same
    order) while a geometry warms up; a replay is bit-identical to that eager form.  Against the plain eager
    Detector.forward the gradient is the same sum with the seq2seq unroll's part added last (fp32 rounding).

        graph = RLTrainGraph(detector)
        stats = graph.step(fact_item, scs_item, senti_labels, xe_senti_labels)    # 7 device scalars of this iteration
    """
    KEYS = ('da_loss', 'fact_reward', 'cls_reward', 'all_rewards', 'cap_loss', 'xe_loss', 'seq2seq_loss')

    def __init__(self, detector, warmup=2, max_geometries=2):
        super().__init__(detector.captioner, detector.cap_optim, detector.cap_xe_crit, detector.cap_da_crit,
                         grad_clip=0.1, arena=detector.dp_arena, group=detector.dp_group, warmup=warmup,
                         max_geometries=max_geometries)
        self.det = detector
        self.share_rl = torch.ones(1, dtype=torch.float32, device=self.device)
        # A third stream: the XE unroll's forward runs there, so autograd replays its reverse sweep there as well - next
        # to the sampled roll-out's on self.stream.  At 512 rows a sweep's launches are 128-192 workgroups on 256 compute
        # units; as two branches of g_bwd the two sweeps fill each other's gaps.
        self.xe_stream = ops.private_stream(self.device)
        # ... and a fourth: the greedy baseline reads what the sampled roll-out reads and draws no random numbers - the two
        # roll-outs run side by side (a graph and a memory pool of its own: graphs that share a pool must not overlap)
        self.greedy_stream = ops.private_stream(self.device)
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self._scope_keys = self._scope_keys + ((idx, self.xe_stream.cuda_stream), (idx, self.greedy_stream.cuda_stream))
        self._handles = self._handles + (self.xe_stream.cuda_stream, self.greedy_stream.cuda_stream)
        self._finalizer.detach()
        self._finalizer = weakref.finalize(self, XETrainGraph._release, self._scope_keys)
        warn_off = getattr(torch.autograd.graph, 'set_warn_on_accumulate_grad_stream_mismatch', None)
        if warn_off is not None:            # gradients of one parameter arrive from two streams by design
            warn_off(False)

    # ---- phases ---------------------------------------------------------------------------------------------------
    def _phase_roll(self, geo):
        """Sampled roll-out (graph kept), domain-align loss; its token matrix on its way to the host (the lengths stay on the
        device: the classifier reward reads them there)."""
        det, cap, i = self.det, self.cap, geo.inputs
        cap.cpt_feats = cap.fc_feats = cap.s2s_cpt_feats = None
        cap.train(True)
        seq, lp, mk = cap(i['fc'], i['att'], i['cpts'], i['sentis'], i['labels'], det.max_seq_len, sample_max=0, mode='rl')
        da = det.cap_da_crit(cap.cpt_feats, cap.fc_feats.detach())
        geo.lens_d = mk.sum(dim=-1).type(torch.int32)
        geo.host[0].copy_(seq, non_blocking=True)
        return seq, lp, mk, da

    def _phase_greedy(self, geo):
        """Greedy baseline, its token matrix on its way to the host.  A phase (and a graph) of its own: the host scores the
        SAMPLED captions while the device decodes these."""
        det, cap, i = self.det, self.cap, geo.inputs
        cap.eval()
        with torch.no_grad():
            gseq, _, gmk = cap(i['fc'], i['att'], i['cpts'], i['sentis'], i['labels'], det.max_seq_len, sample_max=1,
                               mode='rl')
        cap.train(True)
        geo.host[1].copy_(gseq, non_blocking=True)
        return gseq, gmk

    def _phase_fwd(self, geo):
        """Runs on self.xe_stream (the caller sets it), next to the roll-outs: XE unroll forward (its backward then runs there
        too, inside g_bwd); the seq2seq unroll - forward and backward - as a branch on self.side."""
        from .train import _xe_loss
        det, cap, i = self.det, self.cap, geo.inputs
        if 'caps' not in i:       # a 'senti' iteration (decoder.py:52-167 with data_type 'senti'): no ground-truth captions,
            origin = torch.cuda.current_stream(self.device)          # no XE unroll - the seq2seq branch alone
            self.side.wait_stream(origin)
            cap.cpt_feats = cap.fc_feats = cap.s2s_cpt_feats = None
            with torch.cuda.stream(self.side):
                s2s = self._phase_s2s(geo, det.seq2seq_ss_prob)
            origin.wait_stream(self.side)
            return None, s2s
        if self._pair():          # both unrolls through one step chain: forward only here, ONE backward in _phase_bwd
            cap.cpt_feats = cap.fc_feats = cap.s2s_cpt_feats = None
            with cap.token_logprobs():
                pred, pred2 = cap(i['fc'], i['att'], i['cpts'], i['caps'], i['xe_labels'], det.xe_ss_prob,
                                  i['s_caps'], i['s_cpts'], i['s_sentis'], i['s_labels'], det.seq2seq_ss_prob,
                                  mode='xe_seq2seq')
            xe = _xe_loss(det.cap_xe_crit, pred, i['caps'][:, 1:], i['len'])
            s2s_loss = _xe_loss(det.cap_xe_crit, pred2, i['s_caps'][:, 1:], i['s_len'])
            if self._dist():
                s2s_loss = s2s_loss * self.shares[1]
            return xe, (s2s_loss, None)
        origin = torch.cuda.current_stream(self.device)
        self.side.wait_stream(origin)                       # the branches fork here ...
        self.xe_stream.wait_stream(origin)
        cap.cpt_feats = cap.fc_feats = cap.s2s_cpt_feats = None
        # ... the XE unroll is ENQUEUED first (the reference's call order, hence its order of random draws: decoder.py:138,155)
        with torch.cuda.stream(self.xe_stream):
            with cap.token_logprobs():
                pred = cap(i['fc'], i['att'], i['cpts'], i['caps'], i['xe_labels'], ss_prob=det.xe_ss_prob, mode='xe')
            xe = _xe_loss(det.cap_xe_crit, pred, i['caps'][:, 1:], i['len'])
        keep = (cap.cpt_feats, cap.fc_feats)
        with torch.cuda.stream(self.side):
            s2s = self._phase_s2s(geo, det.seq2seq_ss_prob)
        cap.cpt_feats, cap.fc_feats = keep
        origin.wait_stream(self.side)
        origin.wait_stream(self.xe_stream)
        return xe, s2s

    def _phase_bwd(self, geo, roll, fwd):
        """RewardCriterion on the static reward, the reference's sum of losses (decoder.py:160), backward, the seq2seq
        gradients added, the statistics of the iteration; without a process group also clamp + Adam."""
        det = self.det
        seq, lp, mk, da, gseq, gmk = roll
        xe, (s2s_loss, s2s_grads) = fwd
        dist = self._dist()
        cap_loss = det.cap_rl_crit(lp, mk, geo.reward)
        if xe is None:                                       # 'senti' iteration: no XE term
            xe = torch.zeros((), dtype=torch.float32, device=self.device)
        if dist:
            cap_loss, xe, da = cap_loss * self.share_rl[0], xe * self.shares[0], da * self.shares[2]
        s2s_loss = det.seq_flag * s2s_loss
        total = cap_loss + xe + da
        if s2s_grads is None:                                # merged unrolls: the seq2seq loss is part of the one backward
            total = total + s2s_loss
        if self.arena is not None:
            self.arena.zero_()
        else:
            self.optim.zero_grad()
        total.backward()
        dst, src = [], []
        for q, g in zip(self._params, s2s_grads or ()):
            if g is None:
                continue
            if det.seq_flag != 1.0:
                g = g * det.seq_flag
            if q.grad is None:
                q.grad = g
            else:
                dst.append(q.grad)
                src.append(g)
        if dst:
            torch._foreach_add_(dst, src)
        r = geo.reward
        w_rows = self.shares[2] if dist else 1.0
        geo.stats = torch.stack([da.detach(), geo.fact0.mean() * w_rows, geo.cls.mean(-1).mean(-1) * w_rows,
                                 r.mean(-1).mean(-1) * w_rows, cap_loss.detach(), xe.detach(), s2s_loss.detach()])
        if not dist:
            xe_update(self.optim, self.grad_clip)
        return geo.stats

    # ---- between the graphs -----------------------------------------------------------------------------------------
    def _cls_reward(self, geo, roll):
        """The classifier reward of the sampled captions (utils.py:120-151) -> geo.cls, enqueued on the side stream behind the
        sampled roll-out: eager launches of a frozen helper net (an LSTM of the stock library), with the lengths read on the
        device - the host does not wait for anything here, the launches queue up while the roll-out runs."""
        from .rewards import get_cls_reward
        seq, mk = roll[0], roll[2]
        self.side.wait_event(geo.copied_s)
        with torch.cuda.stream(self.side):
            cls = get_cls_reward(seq, mk, None, None, geo.inputs['labels'], self.det.sent_senti_cls,
                                 sample_lens=geo.lens_d, on_device=True)
            geo.cls.copy_(cls)

    def _rewards(self, geo, roll, item):
        """CIDEr-D on the host -> with geo.cls the static reward buffer the captured RewardCriterion reads.  The sampled
        captions are scored as soon as they have landed (`copied_s`; the device is in the greedy roll-out and g_fwd), the
        greedy ones when they land (`copied`)."""
        from .rewards import self_critical_scores
        det = self.det
        fact = 'caps' in geo.inputs                           # ('senti' iterations: classifier reward only, decoder.py:120-124)
        if fact:
            fns, ground_truth = item[0], item[6]
            geo.copied_s.synchronize()
            sampled = self_critical_scores(geo.host[0].numpy(), fns, ground_truth, det.ciderd_scorer)
        geo.copied_s.synchronize()
        geo.copied.synchronize()
        if getattr(self.cap, 'numerics_checks', True) and ops.device_status(reset=True):
            # the roll-outs met non-finite values (features beyond the split-f16 domain): nothing has been updated yet -
            # Detector.forward redoes this iteration eagerly on the exact-fp32 engine
            self.stream.wait_stream(self.side)
            self.stream.wait_event(geo.fwd_done)
            raise ops.OutOfDomain()
        if not fact:
            self.stream.wait_stream(self.side)
            geo.reward.copy_(0 + det.cls_flag * geo.cls)     # (`fact_reward = 0` there: the same expression as the eager path's)
            return
        greedy = self_critical_scores(geo.host[1].numpy(), fns, ground_truth, det.ciderd_scorer)
        fact0 = ops.upload((sampled - greedy).astype('float32'), torch.float32, self.device)     # (utils.py:56-83: one per row)
        geo.fact0.copy_(fact0)
        self.stream.wait_stream(self.side)
        geo.reward.copy_(fact0.unsqueeze(1) + det.cls_flag * geo.cls)

    def _fork_fwd(self, geo, roll, lengths, s_lengths):
        """self.xe_stream may start g_fwd: it reads the staged inputs and the weights only - not the roll-outs - so it runs
        next to them.  Under a process group it reads the loss shares as well, whose one small all-reduce carries the
        roll-out's mask sum: the branch then starts behind that."""
        if self._dist():
            self._shares(geo, roll, lengths, s_lengths)
            geo.staged.record(self.stream)
        self.xe_stream.wait_event(geo.staged)

    def _late_inputs(self, geo):
        """Inputs the roll-outs do not read, handed to step() as callables: produced now, with the sampled roll-out already
        queued - the host work of producing them (a frozen helper net's eager launches) no longer stands in front of the
        iteration's first graph."""
        for k, fn in geo.late.items():
            geo.inputs[k].copy_(fn(), non_blocking=True)
        geo.late = {}

    def _shares(self, geo, roll, lengths, s_lengths):
        """DP: each term's share of its global normaliser (mask sum, XE tokens, seq2seq tokens, rows): one 4-float
        all-reduce per iteration, as Detector.forward does."""
        if not self._dist():
            return
        n_local, n_global = dp.global_counts([roll[2].sum(), float(sum(lengths or ())), float(sum(s_lengths)),
                                              float(geo.inputs['fc'].shape[0])], self.device, self.group)
        w = n_local / n_global.clamp_min(1.0)
        self.share_rl.copy_(w[0:1])
        self.shares.copy_(w[1:4])

    def _alloc(self, geo):
        i = geo.inputs
        B, T = i['fc'].shape[0], self.det.max_seq_len
        geo.host = (torch.empty(B, T, dtype=torch.int64).pin_memory(), torch.empty(B, T, dtype=torch.int64).pin_memory())
        geo.reward = torch.zeros(B, T, dtype=torch.float32, device=self.device)
        geo.fact0 = torch.zeros(B, dtype=torch.float32, device=self.device)
        geo.cls = torch.zeros(B, T, dtype=torch.float32, device=self.device)
        geo.copied, geo.copied_s, geo.staged, geo.fwd_done = [torch.cuda.Event() for _ in range(4)]

    # ---- eager / capture / replay -----------------------------------------------------------------------------------
    def _run_eager(self, geo, item, lengths, s_lengths):
        with ops.refresh_only(self._handles):
            geo.staged.record(self.stream)
            self.greedy_stream.wait_event(geo.staged)
            roll = self._phase_roll(geo)
            geo.copied_s.record(self.stream)
            with torch.cuda.stream(self.greedy_stream):
                roll = roll + self._phase_greedy(geo)
                geo.copied.record(self.greedy_stream)
            self._cls_reward(geo, roll)
            self._fork_fwd(geo, roll, lengths, s_lengths)
            with torch.cuda.stream(self.xe_stream):
                self._late_inputs(geo)
                fwd = self._phase_fwd(geo)
                geo.fwd_done.record(self.xe_stream)
            for g in (fwd[0], fwd[1][0]) + tuple(fwd[1][1] or ()):
                if g is not None:
                    g.record_stream(self.stream)
            self._rewards(geo, roll, item)
            self.stream.wait_event(geo.fwd_done)
            stats = self._phase_bwd(geo, roll, fwd)
            if self._dist():
                if self.arena is not None:
                    self.arena.all_reduce(self.group)       # (the statistics stay this rank's shares: the caller sums them)
                xe_update(self.optim, self.grad_clip)
        self._valid_key = self.cap._weights_key()
        self.eager_steps += 1
        return stats

    def _capture_rl(self, geo):
        steps_before = [float(st['step']) for st in self._states()]
        self.optim.device_hyper = self.hyper
        geo.pool = torch.cuda.graph_pool_handle()
        try:
            with ops.refresh_only(self._handles):
                geo.g_roll = torch.cuda.CUDAGraph(keep_graph=self.KEEP_GRAPHS)
                with ops.graph_capture(geo.g_roll, stream=self.stream, pool=geo.pool):
                    roll = self._phase_roll(geo)
                geo.g_greedy = torch.cuda.CUDAGraph(keep_graph=self.KEEP_GRAPHS)
                with ops.graph_capture(geo.g_greedy, stream=self.greedy_stream, settle=False):
                    roll = roll + self._phase_greedy(geo)
                geo.g_fwd = torch.cuda.CUDAGraph(keep_graph=self.KEEP_GRAPHS)
                # (a pool of its own: it runs next to g_roll.  Captured from self.stream with self.xe_stream and self.side as
                # two branches of it - a branch forked off a branch ended in a segmentation fault inside the runtime's
                # end-of-capture here - and replayed on self.xe_stream)
                with ops.graph_capture(geo.g_fwd, stream=self.stream, settle=False):
                    fwd = self._phase_fwd(geo)
                geo.g_bwd = torch.cuda.CUDAGraph(keep_graph=self.KEEP_GRAPHS)
                with ops.graph_capture(geo.g_bwd, stream=self.stream, pool=geo.pool, settle=False):
                    self._phase_bwd(geo, roll, fwd)
                geo.g_up = None
                if self._dist():
                    geo.g_up = torch.cuda.CUDAGraph(keep_graph=self.KEEP_GRAPHS)
                    with ops.graph_capture(geo.g_up, stream=self.stream, pool=geo.pool, settle=False):
                        xe_update(self.optim, self.grad_clip)
                geo.keep = (roll, fwd)
        finally:
            self.optim.device_hyper = None
            for st, n in zip(self._states(), steps_before):      # (also when the capture fails half-way)
                st['step'].fill_(n)
        geo.g_iter = geo.g_roll                              # (the base class's "is captured" marker)
        geo.layout = ops.h3_weights_scope.cold_begins(self._scope_keys)
        self._valid_key = self.cap._weights_key()
        self.captures += 1

    def _replay_rl(self, geo, item, lengths, s_lengths):
        states = self._states()
        for st in states:
            st['step'] += 1
        self._set_hyper(int(states[0]['step']))
        roll, fwd = geo.keep
        geo.staged.record(self.stream)
        self.greedy_stream.wait_event(geo.staged)            # (the inputs are staged, the last update's plane refresh is done)
        geo.g_roll.replay()
        geo.copied_s.record(self.stream)
        with torch.cuda.stream(self.greedy_stream):
            geo.g_greedy.replay()
            geo.copied.record(self.greedy_stream)
        self._cls_reward(geo, roll)
        self._fork_fwd(geo, roll, lengths, s_lengths)
        with torch.cuda.stream(self.xe_stream):
            self._late_inputs(geo)
            geo.g_fwd.replay()
            geo.fwd_done.record(self.xe_stream)
        try:
            self._rewards(geo, roll, item)
        except ops.OutOfDomain:
            for st in states:               # this iteration will be redone elsewhere: it has not stepped
                st['step'] -= 1
            raise
        self.stream.wait_event(geo.fwd_done)
        geo.g_bwd.replay()
        stats = geo.stats
        if geo.g_up is not None:
            if self.arena is not None:
                self.arena.all_reduce(self.group)
            geo.g_up.replay()
        epoch_before = ops.WEIGHT_EPOCH
        ops.WEIGHT_EPOCH += 1
        ops.h3_weights_scope.rekey_epoch(self._scope_keys, epoch_before)
        self._valid_key = self.cap._weights_key()
        self.replays += 1
        return stats

    def step(self, item, scs_batch, senti_labels, xe_senti_labels=None):
        """One iteration on a fact item of the rl_fact collate (fns, fc, att, (caps, lengths), cpts, sentis,
        ground_truth) - or a senti item of the rl_senti collate (fns, fc, att, cpts, sentis, labels: no captions, hence no
        XE unroll and no CIDEr-D reward; the statistics then lack 'fact_reward' and 'xe_loss') -, a seq2seq batch, the image sentiment labels and the XE labels of the captions (both from the
        frozen helper nets, computed by the caller; `xe_senti_labels` may be a callable returning them - it is then run behind
        the sampled roll-out, which does not read them, so that its host work overlaps the device's).  Returns {key: 0-dim device tensor} over RLTrainGraph.KEYS (under DP:
        this rank's pre-scaled shares - their sum over the ranks is the global value, as in Detector.forward), valid on
        the caller's current stream."""
        (s_caps, s_lengths), s_cpts, s_sentis, s_labels = scs_batch
        s_lengths = self._as_list(s_lengths)
        if len(item) == 6:        # a 'senti' item of the rl_senti collate: (fns, fc, att, cpts, sentis, labels) - no captions
            fns, fc, att, cpts, sentis, _ = item
            lengths = None
            if s_caps.size(1) - 1 < max(s_lengths):
                raise ValueError('caption tensors are narrower than their longest caption (+1 for <SOS>)')
            t = dict(fc=fc, att=att, cpts=cpts, sentis=sentis, labels=senti_labels, s_caps=s_caps, s_cpts=s_cpts,
                     s_sentis=s_sentis, s_labels=s_labels, s_len=torch.tensor(s_lengths, dtype=torch.int32))
        else:
            fns, fc, att, (caps, lengths), cpts, sentis, ground_truth = item
            lengths = self._as_list(lengths)
            if caps.size(1) - 1 < max(lengths) or s_caps.size(1) - 1 < max(s_lengths):
                raise ValueError('caption tensors are narrower than their longest caption (+1 for <SOS>)')
            t = dict(fc=fc, att=att, caps=caps, cpts=cpts, sentis=sentis, labels=senti_labels, xe_labels=xe_senti_labels,
                     len=torch.tensor(lengths, dtype=torch.int32), s_caps=s_caps, s_cpts=s_cpts, s_sentis=s_sentis,
                     s_labels=s_labels, s_len=torch.tensor(s_lengths, dtype=torch.int32))
        late = {}
        if lengths is not None and callable(xe_senti_labels):       # [B] int64, read by g_fwd only: produced behind the first roll-out (_late_inputs)
            late['xe_labels'] = xe_senti_labels
            t['xe_labels'] = torch.empty(fc.shape[0], dtype=torch.int64, device='meta')
        self._claim()
        self.cap.cpt_feats = self.cap.fc_feats = self.cap.s2s_cpt_feats = None
        sig = (self._signature(t, 0.0), self.det.max_seq_len, self.det.xe_ss_prob, self.det.seq2seq_ss_prob,
               self.det.cls_flag, self.det.seq_flag)
        geo = self._geoms.get(sig)
        if geo is None:
            geo = self._geoms[sig] = _RLGeometry()
            while len(self._geoms) > self._max_geoms:
                self._evicted(self._geoms.popitem(last=False)[1])
        else:
            self._geoms.move_to_end(sig)
        caller = torch.cuda.current_stream(self.device)
        self.stream.wait_stream(caller)
        try:
            with torch.cuda.stream(self.stream):
                self._stage(geo, t)
                geo.late = late
                if geo.host is None:
                    self._alloc(geo)
                planes_ok = self._valid_key is not None and self._valid_key == self.cap._weights_key()
                if ops.h3_weights_scope.cold_begins(self._scope_keys) != geo.layout:
                    geo.g_iter = geo.g_roll = geo.g_greedy = geo.g_fwd = geo.g_bwd = geo.g_up = None
                if geo.g_iter is None and planes_ok and geo.eager_runs >= self.warmup:
                    self._capture_rl(geo)
                if geo.g_iter is not None and planes_ok:
                    stats = self._replay_rl(geo, item, lengths, s_lengths)
                else:
                    stats = self._run_eager(geo, item, lengths, s_lengths)
                    geo.eager_runs += 1
                out = dict(zip(self.KEYS, stats.clone().unbind(0)))
                if lengths is None:                  # ('senti': no fact_reward, no xe_loss - decoder.py:120-158)
                    out = {k: v for k, v in out.items() if k not in ('fact_reward', 'xe_loss')}
        except ops.OutOfDomain:
            caller.wait_stream(self.stream)          # the caller redoes the iteration on ITS stream: behind what was queued
            self.cap.cpt_feats = self.cap.fc_feats = self.cap.s2s_cpt_feats = None
            raise
        caller.wait_stream(self.stream)
        for v in out.values():
            v.record_stream(caller)
        return out
