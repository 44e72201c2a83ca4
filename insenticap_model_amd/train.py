"""The inner XE training step of the reference (train_xe.py:149-192) on the HIP path, optionally
data-parallel.  Batch tuple layouts are the reference collate functions' (dataloader.py:11-58):
  fact batch: (fns, fc_feats [B,F], att_feats [B,...,F], (caps [B,L], lengths), cpts [B,C])
  scs  batch: ((caps [80,L], lengths), cpts [80,C], sentis [80,10], senti_labels [80])
The sentence-sentiment classifier that labels the factual captions (train_xe.py:155-158) is a
frozen helper outside this path: the caller passes its arg-max as `xe_senti_labels`.
"""
import torch

from . import dp, ops
from .optim import FusedClampAdam, clip_gradient


_SIDE_STREAMS = {}


def _side_stream(device):
    key = (device.type, device.index)
    if key not in _SIDE_STREAMS:
        from . import ops
        _SIDE_STREAMS[key] = ops.private_stream(device)
        # gradients of one parameter arrive from two streams by design
        warn_off = getattr(torch.autograd.graph, 'set_warn_on_accumulate_grad_stream_mismatch', None)
        if warn_off is not None:
            warn_off(False)
    return _SIDE_STREAMS[key]


def run_on_side_stream(device, fn, side=None):
    """fn() with its launches on the device's side stream, ordered after everything queued so far on the current
    stream; the current stream then waits for it.  Returns fn()'s tensor result (recorded on the current stream)."""
    main = torch.cuda.current_stream(device)
    side = side if side is not None else _side_stream(device)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        out = fn()
    main.wait_stream(side)
    out.record_stream(main)
    return out


def _xe_loss(xe_crit, pred, target, lengths):
    """xe_crit(pred, target, lengths) (train_xe.py:164); `lengths` already on the device (a captured iteration's static
    input, validated by the caller): straight to the criterion's launch, no host-side max()."""
    if isinstance(lengths, torch.Tensor) and lengths.is_cuda:
        from .autograd import xe_criterion_with_grad
        return xe_criterion_with_grad(pred, target, lengths)
    return xe_crit(pred, target, lengths)


def xe_forward_backward(captioner, optim, xe_crit, da_crit, fact, xe_senti_labels, scs=None, ss_prob=0.0, arena=None,
                        weights3=None, overlap_unrolls=True, side_stream=None, pair=None, sink=None):
    """train_xe.py:160-190 on device tensors: both unrolls, the three losses, backward.  `fact` = (fc, att, caps,
    lengths, cpts), `scs` = (caps, lengths, cpts, sentis, labels) or None; `weights3` = this rank's shares of the three
    global normalisers (XE tokens, seq2seq tokens, rows) as a device tensor (or a callable returning it, resolved after
    the unrolls' forward: dp_shares_async), or None (single process: graph untouched).  Returns the detached [xe, da, seq2seq] losses as one device tensor.  No collective, no host read:
    this is the part of an iteration that train_graph.XETrainGraph captures into a HIP graph."""
    fc_feats, att_feats, caps_tensor, lengths, cpts_tensor = fact
    device = fc_feats.device
    share = (lambda x, w: x * w) if weights3 is not None else (lambda x, w: x)
    if pair is None:
        from .autograd_pair import use_pair
        pair = use_pair(captioner, False)
        if (pair and sink is None and getattr(captioner, 'pair_unrolls', None) is None
                and not (captioner.training and ss_prob > 0.0)        # (scheduled sampling keeps the full unroll)
                and captioner.ragged_applies(lengths)):
            pair = False               # (captioner.ragged_unroll: one chain per unroll, each on the rows inside their captions)
    pair = pair and scs is not None and device.type == 'cuda'
    # (token_logprobs: the unrolls hand the criterion log p(target) [B,T] - the [B,T,V] log-probs are never formed)
    if pair:
        # both unrolls through ONE step chain (Captioner.forward_xe_seq2seq / autograd_pair): the LSTM cells, the
        # classifier and every backward contraction run once over the 128 + 80 rows instead of once per unroll
        s_caps, s_lengths, s_cpts, s_sentis, s_labels = scs
        with captioner.token_logprobs():
            pred, pred2 = captioner(fc_feats, att_feats, cpts_tensor, caps_tensor, xe_senti_labels, ss_prob,
                                    s_caps, s_cpts, s_sentis, s_labels, ss_prob, mode='xe_seq2seq')
    else:
        with captioner.token_logprobs(), captioner.row_counts(lengths):
            pred = captioner(fc_feats, att_feats, cpts_tensor, caps_tensor, xe_senti_labels, ss_prob, mode='xe')
    if callable(weights3):            # (dp_shares_async: the counts' all-reduce ran next to the unroll; wait for it here)
        weights3 = weights3()
    w_xe, w_s2s, w_rows = weights3.unbind(0) if weights3 is not None else (None, None, None)
    xe_bwd = share(_xe_loss(xe_crit, pred, caps_tensor[:, 1:], lengths), w_xe)
    da_bwd = share(da_crit(captioner.cpt_feats, captioner.fc_feats.detach()), w_rows)
    total = xe_bwd + da_bwd
    s2s_d = torch.zeros((), device=device)
    if pair:
        s2s = share(_xe_loss(xe_crit, pred2, s_caps[:, 1:], s_lengths), w_s2s)
        total = total + s2s
        s2s_d = s2s.detach()
    elif scs is not None:
        s_caps, s_lengths, s_cpts, s_sentis, s_labels = scs

        def seq2seq_unroll():
            with captioner.token_logprobs(), captioner.row_counts(s_lengths):
                pred2 = captioner(s_caps, s_cpts, s_sentis, s_labels, ss_prob, mode='seq2seq')
            return share(_xe_loss(xe_crit, pred2, s_caps[:, 1:], s_lengths), w_s2s)
        if overlap_unrolls and device.type == 'cuda':
            s2s = run_on_side_stream(device, seq2seq_unroll, side_stream)
        else:
            s2s = seq2seq_unroll()
        total = total + s2s
        s2s_d = s2s.detach()
    if arena is not None:
        arena.zero_()
    else:
        optim.zero_grad()
    if sink is not None and pair:
        # data-parallel, merged unrolls: the backward writes every gradient into its arena view and starts each bucket's
        # all-reduce as soon as the bucket is complete (dp.GradSink; autograd_pair._pair_backward)
        sink.begin()
        captioner._grad_sink = sink
        try:
            total.backward()
        finally:
            captioner._grad_sink = None
    else:
        total.backward()
    return torch.stack([xe_bwd.detach(), da_bwd.detach(), s2s_d])


def xe_update(optim, grad_clip):
    """train_xe.py:191-192: clamp (fused into the Adam launch) + step."""
    clip_gradient(optim, grad_clip)
    optim.step()


def loss_dict(vec3):
    out = dict(zip(('xe_loss', 'da_loss', 'seq2seq_loss'), vec3.unbind(0)))
    out['cap_loss'] = out['xe_loss'] + out['da_loss']
    out['all_loss'] = out['cap_loss'] + out['seq2seq_loss']
    return out


def dp_shares(lengths, s_lengths, rows, device, group):
    """Each loss term's share of ITS global normaliser (XE tokens, seq2seq tokens, rows): ONE 3-float all-reduce, built
    on the rank's GPU (RCCL moves device tensors only), before any compute depends on it."""
    local, glob = dp.global_counts([float(sum(lengths)), float(sum(s_lengths)) if s_lengths is not None else 0.0,
                                    float(rows)], device, group)
    return local / glob.clamp_min(1.0)


def dp_shares_async(lengths, s_lengths, rows, device, group):
    """dp_shares whose all-reduce does not hold the compute stream: returns a callable that waits for the reduction (on
    the then-current stream) and returns the shares - the eager step calls it after the unrolls' forward passes, where
    the first loss is scaled, so the collective's latency is behind ~2 ms of launches instead of in front of them."""
    local, glob, work = dp.global_counts([float(sum(lengths)), float(sum(s_lengths)) if s_lengths is not None else 0.0,
                                          float(rows)], device, group, asynchronous=True)

    def shares():
        if work is not None:
            work.wait()
        return local / glob.clamp_min(1.0)
    return shares


def xe_train_step(captioner, optim, xe_crit, da_crit, fact_batch, xe_senti_labels, scs_batch=None,
                  ss_prob=0.0, grad_clip=0.1, arena=None, group=None, device=None, overlap_unrolls=True, bucketed=True):
    """One iteration. Returns dict(xe_loss, da_loss, cap_loss, seq2seq_loss, all_loss) of 0-dim
    tensors (global values under DP).  `arena`: dp.GradArena when gradients are all-reduced.
    `overlap_unrolls`: the seq2seq unroll (80 text-only rows) runs on a side HIP stream.  It shares nothing
    with the XE unroll but the weights, and at these batch sizes both are chains of small launches that leave
    most of the chip idle; autograd replays each unroll's backward on the stream its forward ran on, so the
    two backward sweeps overlap as well.  Same numbers either way (two-operand gradient sums commute).
    (train_graph.XETrainGraph runs the same three phases from HIP graphs.)
    `bucketed` (with an arena and the merged unrolls): the gradient exchange goes out in four buckets from inside the
    backward pass, overlapped with its dW contractions, and clamp + Adam run per bucket behind each reduction
    (dp.GradSink); False = one flat all-reduce after the backward.  Same parameters after the step, bit for bit."""
    device = torch.device(device) if device is not None else next(captioner.parameters()).device
    _, fc_feats, att_feats, (caps_tensor, lengths), cpts_tensor = fact_batch[:5]
    fact = (ops.to_device(fc_feats, device), ops.to_device(att_feats, device), ops.to_device(caps_tensor, device), lengths,
            ops.to_device(cpts_tensor, device))
    xe_senti_labels = ops.to_device(xe_senti_labels, device)
    scs = None
    if scs_batch is not None:
        (s_caps, s_lengths), s_cpts, s_sentis, s_labels = scs_batch
        scs = (ops.to_device(s_caps, device), s_lengths, ops.to_device(s_cpts, device), ops.to_device(s_sentis, device),
               ops.to_device(s_labels, device))
    # data-parallel: taken whenever a process group exists (also a one-rank one: shares are then exactly 1.0), so the
    # single-GPU RCCL test crosses every branch an 8-rank run does
    dist_on = dp.distributed(group)
    weights3 = dp_shares_async(lengths, scs[1] if scs is not None else None, fact[0].shape[0], device, group) \
        if dist_on else None
    from .autograd_pair import use_pair
    sink = None
    if (arena is not None and bucketed and scs is not None and device.type == 'cuda' and use_pair(captioner, False)
            and isinstance(optim, FusedClampAdam)):
        sink = captioner.__dict__.get('_dp_sink')
        if sink is None or sink.arena is not arena or sink.group is not group:
            sink = captioner.__dict__['_dp_sink'] = dp.GradSink(captioner, arena, group)
    vec = xe_forward_backward(captioner, optim, xe_crit, da_crit, fact, xe_senti_labels, scs, ss_prob, arena, weights3,
                              overlap_unrolls, sink=sink)
    if sink is not None and sink.order:
        # the loss statistics: queued behind the buckets on the backend's stream, NOT waited for here - the compute stream
        # would otherwise stand behind every bucket's reduction before the first bucket's update
        vec, work = dp.all_reduce_async_(vec, group) if dist_on else (vec, None)
        sink.finish(optim, grad_clip)    # per bucket: wait for its reduction, clamp + Adam
        if work is not None:
            work.wait()
        return loss_dict(vec)
    if arena is not None:
        arena.all_reduce(group)          # one 88 MB sum over xGMI; clamp must see reduced grads
    if dist_on:                          # report global losses (sum of the pre-scaled locals): one 3-float all-reduce
        vec = dp.all_reduce_(vec, group)
    xe_update(optim, grad_clip)
    return loss_dict(vec)
