"""`Detector`: the RL-stage trainer wrapper of the reference (/root/reference/models/decoder.py:21-192)
on the MI355X path - same constructor, setters, `forward(data, data_type, training)` loss dictionary
and `sample(...)`.

One RL iteration = sampled roll-out (with REINFORCE gradients) + greedy roll-out (baseline) +
CIDEr-D and classifier rewards + XE pass (ss_prob 0.5) + seq2seq pass (ss_prob 0.25) + backward +
clamp(0.1) + Adam (decoder.py:65-167).  The decoder work runs on the HIP kernels through
`Captioner`; CIDEr-D runs in the native host library; the two frozen helper nets are stock
PyTorch-ROCm modules (helper_nets.py).
"""
from collections import defaultdict

import torch
import torch.nn as nn

from . import dp, ops
from .captioner import Captioner
from .helper_nets import SentenceSentimentClassifier, SentimentDetector
from .optim import clip_gradient
from .train import run_on_side_stream
from .ops import OutOfDomain
from .rewards import RewardCriterion, get_ciderd_scorer, get_cls_reward, get_self_critical_reward


class Detector(nn.Module):
    MAX_BATCHES_PER_CALL = 500      # decoder.py:65

    def __init__(self, idx2word, max_seq_len, sentiment_categories, lrs, settings):
        super().__init__()
        self.idx2word = idx2word
        self.pad_id = idx2word.index('<PAD>')
        self.max_seq_len = max_seq_len
        self.captioner = Captioner(idx2word, sentiment_categories, settings)
        self.senti_detector = SentimentDetector(sentiment_categories, settings)
        self.sent_senti_cls = SentenceSentimentClassifier(idx2word, sentiment_categories, settings)
        self.senti_detector.eval()
        self.sent_senti_cls.eval()
        self.cap_optim, self.cap_xe_crit, self.cap_da_crit = self.captioner.get_optim_criterion(lrs['cap_lr'])
        self.cap_rl_crit = RewardCriterion()
        self.cls_flag = 0.4
        self.seq_flag = 1.0
        self.senti_threshold = 0.7
        self.xe_ss_prob, self.seq2seq_ss_prob = 0.5, 0.25     # decoder.py:139,155 (hard-coded there)
        self.dp_arena, self.dp_group = None, None             # data-parallel training: enable_data_parallel()
        # The image sentiment detector is frozen (decoder.py:31,34: only captioner.parameters() are
        # optimised), so its label is a pure function of the image: cache it per file name instead of
        # re-running 1.7-9.2 GFLOP of convolutions per image in every RL iteration.
        self.cache_image_sentiments = True
        self._senti_cache, self._senti_cache_key = {}, None
        # 'fact' training iterations are served from HIP graphs (train_graph.RLTrainGraph: roll-outs / unrolls /
        # backward + update captured per batch geometry after two eager iterations; CIDEr-D and the classifier reward
        # between them).  False: every iteration runs the eager sequence below.
        self.train_graphs = True
        self._rl_graph = None

    def enable_data_parallel(self, group=None, broadcast=True):
        """Data-parallel RL training over the ranks of `group` (default process group; backend nccl = RCCL over xGMI,
        gloo in tests).  The reference is single-device; under DP every rank runs `forward` on ITS shard of the fact /
        seq2seq batches (the caller shards the loaders), and the step becomes:
          * roll-outs, helper nets, CIDEr-D reward (document frequencies replicated) and classifier reward rank-local;
          * every loss term pre-scaled by local_count / global_count of ITS normaliser - RewardCriterion by the mask
            sum (utils.py:175), the two XECriterion terms by their token counts (captioner.py:438), the domain-align
            MSE by the row count - one 4-float all-reduce per iteration;
          * ONE sum-all-reduce of the flat 88 MB gradient arena between backward and the elementwise clamp
            (decoder.py:165-166: the clamp is nonlinear, it must see the reduced gradient), then identical
            clamp+Adam on every rank;
          * the returned dictionary holds global values (one all-reduce of the statistics per call)."""
        self.dp_group = group
        self.dp_arena = dp.GradArena(self.captioner.parameters())
        if broadcast:
            dp.broadcast_parameters(self.captioner, group=group)
        return self

    def set_ciderd_scorer(self, captions):
        self.ciderd_scorer = get_ciderd_scorer(captions, self.captioner.sos_id, self.captioner.eos_id)

    def set_sentiment_words(self, sentiment_words):
        self.sentiment_words = sentiment_words

    def set_lms(self, lms):
        self.lms = lms

    def forward(self, data, data_type, training):
        """data = (caption loader[, seq2seq loader]); data_type 'fact' | 'senti'. Returns the dict of
        loss sums divided by len(data) - the tuple length, as the reference does (decoder.py:178-179)."""
        if data_type not in ('fact', 'senti'):
            raise Exception('data_type(%s) is wrong!' % data_type)
        cap = self.captioner
        cap.train(training)
        # this call looks at the numerics flags at each iteration's own synchronisation point (below) instead of reducing
        # the features in front of every roll-out (Captioner._features_in_domain: a host read per new tensor)
        own_check, cap._domain_check_off = cap.__dict__.get('_domain_check_off'), True
        try:
            return self._forward(data, data_type, training)
        finally:
            cap._domain_check_off = own_check

    def _forward(self, data, data_type, training):
        cap = self.captioner
        # statistics stay on the device until the end of the call: every float(tensor) would stall the host
        # behind the whole queue (the reference reads them one by one, decoder.py:90-163)
        sums = defaultdict(float)

        def add(key, value):
            sums[key] = sums[key] + (value.detach() if torch.is_tensor(value) else value)

        device = next(self.parameters()).device
        # data-parallel branches: taken whenever DP is enabled and a process group exists - also a one-rank group
        # (shares are then exactly 1.0), so a one-GPU RCCL run crosses the code an 8-rank run does
        dist_on = self.dp_arena is not None and dp.distributed(self.dp_group)
        seq2seq_iter = iter(data[1]) if training else None
        caption_iter = iter(data[0])
        n_iter = min(self.MAX_BATCHES_PER_CALL, len(data[0]))
        if dist_on:
            # every iteration issues collectives: ranks whose loaders differ in length (or in data_type) would hang
            # inside RCCL - check once per call instead (shard with dp.shard(..., drop_last=True))
            dp.assert_same_across_ranks(n_iter * 4 + (2 if training else 0) + (data_type == 'fact'), device,
                                        self.dp_group, 'Detector.forward: iterations / data_type / training')
        for _ in range(n_iter):
            item = next(caption_iter)
            s2s_batch = None
            if training:                                  # fetched here (same loader order) so that its token
                try:                                      # count can ride in the iteration's one count all-reduce
                    s2s_batch = next(seq2seq_iter)
                except StopIteration:
                    seq2seq_iter = iter(data[1])
                    s2s_batch = next(seq2seq_iter)
            if data_type == 'fact':
                fns, fc_feats, att_feats, (caps_tensor, lengths), cpts_tensor, sentis_tensor, ground_truth = item
                caps_tensor = ops.to_device(caps_tensor, device)
            else:
                fns, fc_feats, att_feats, cpts_tensor, sentis_tensor, senti_labels = item
                senti_labels = ops.to_device(senti_labels, device)
            fc_feats, att_feats = ops.to_device(fc_feats, device), ops.to_device(att_feats, device)
            # (ops.to_device: through pinned memory, non-blocking - `x.to(device)` from pageable memory holds the host until the
            # stream gets there, i.e. until the previous iteration has finished)
            cpts_tensor, sentis_tensor = ops.to_device(cpts_tensor, device), ops.to_device(sentis_tensor, device)
            del item

            if data_type == 'fact' or not training:      # labels from the image sentiment detector
                senti_labels = self._image_sentiments(fns, att_feats)

            def run_iteration(exact, put):
                """One iteration (models/decoder.py:69-167).  `exact`: on the exact-fp32 GEMM engine, eagerly - the retry of
                an iteration whose roll-outs left the split-f16 operand domain (OutOfDomain is raised at the iteration's
                own synchronisation point, before anything is updated)."""
                if (training and self.train_graphs and device.type == 'cuda'
                        and ops.TIMER.arm_step is None and ops.graphs_allowed_here() and not exact):
                    # the same iteration from HIP graphs (train_graph.RLTrainGraph): same calls in the same order
                    from .train_graph import RLTrainGraph
                    g = self._rl_graph
                    if g is None or g.arena is not self.dp_arena or g.group is not self.dp_group or \
                            g.optim is not self.cap_optim or g.xe_crit is not self.cap_xe_crit or \
                            g.da_crit is not self.cap_da_crit:
                        # (the graph object holds its own references to the optimizer, the criteria, the arena and the
                        # group: a replaced optimizer - a new learning-rate schedule object, say - must not leave replays
                        # updating through the old one)
                        self._rl_graph = RLTrainGraph(self)
                    (s_caps, s_lengths), s_cpts, s_sentis, s_labels = s2s_batch
                    scs_dev = ((ops.to_device(s_caps, device), s_lengths), ops.to_device(s_cpts, device),
                               ops.to_device(s_sentis, device), ops.to_device(s_labels, device))
                    if data_type == 'fact':
                        def xe_senti_labels():
                            with torch.no_grad():
                                logits = self.sent_senti_cls(caps_tensor[:, 1:], lengths)[0]
                                return logits.softmax(dim=-1).argmax(dim=-1).detach()
                        if self.sent_senti_cls.training:     # (dropout draws: keep the reference's order of random numbers)
                            xe_senti_labels = xe_senti_labels()
                        # else: a callable - the graph object runs it behind its first roll-out, whose input it is not
                        stats = self._rl_graph.step(
                            (fns, fc_feats, att_feats, (caps_tensor, lengths), cpts_tensor, sentis_tensor, ground_truth),
                            scs_dev, senti_labels, xe_senti_labels)
                    else:                                    # 'senti': labelled images, no captions (decoder.py:75-78)
                        stats = self._rl_graph.step((fns, fc_feats, att_feats, cpts_tensor, sentis_tensor, senti_labels),
                                                    scs_dev, senti_labels)
                    for k, v in stats.items():
                        put(k, v)
                    return

                # sampled roll-out (graph kept in train mode) and the domain-alignment loss on its prologue
                sample_captions, sample_logprobs, seq_masks = cap(
                    fc_feats, att_feats, cpts_tensor, sentis_tensor, senti_labels, self.max_seq_len,
                    sample_max=0, mode='rl')
                da_loss = self.cap_da_crit(cap.cpt_feats, cap.fc_feats.detach())
                # DP: each term's share of the global normaliser (mask sum, XE tokens, seq2seq tokens, rows)
                w_rl = w_xe = w_s2s = w_rows = None
                share = (lambda x, w: x * w) if dist_on else (lambda x, w: x)      # single process: graph untouched
                if dist_on:
                    n_local, n_global = dp.global_counts(
                        [seq_masks.sum(), float(sum(lengths)) if data_type == 'fact' else 0.0,
                         float(sum(s2s_batch[0][1])) if s2s_batch is not None else 0.0,
                         float(fc_feats.shape[0])], device, self.dp_group)
                    w_rl, w_xe, w_s2s, w_rows = (n_local / n_global.clamp_min(1.0)).unbind(0)
                da_loss = share(da_loss, w_rows)
                put('da_loss', da_loss)

                cap.eval()                                   # greedy baseline
                with torch.no_grad():
                    greedy_captions, _, greedy_masks = cap(
                        fc_feats, att_feats, cpts_tensor, sentis_tensor, senti_labels, self.max_seq_len,
                        sample_max=1, mode='rl')
                cap.train(training)

                # The rewards need the token matrices on the host (CIDEr-D is host code).  Start their copies
                # now, enqueue the XE / seq2seq forward passes, and only then wait: the host scores the captions
                # while the device works through the two unrolls.  The order of the captioner calls - hence of
                # every random draw - is the reference's; only host-side waiting moved.
                host_sample = torch.empty(sample_captions.shape, dtype=sample_captions.dtype).pin_memory()
                host_greedy = torch.empty(greedy_captions.shape, dtype=greedy_captions.dtype).pin_memory()
                host_lens = torch.empty(seq_masks.shape[0], dtype=torch.int32).pin_memory()
                host_sample.copy_(sample_captions, non_blocking=True)
                host_greedy.copy_(greedy_captions, non_blocking=True)
                host_lens.copy_(seq_masks.sum(dim=-1).type(torch.int32), non_blocking=True)
                copied = torch.cuda.Event()
                copied.record()

                xe_loss = 0.0
                seq2seq_loss = 0.0
                from .autograd_pair import use_pair
                merged = data_type == 'fact' and training and device.type == 'cuda' and use_pair(cap, False)
                if merged:                                   # XE + seq2seq unrolls through one step chain (autograd_pair)
                    with torch.no_grad():
                        xe_senti_labels = self.sent_senti_cls(caps_tensor[:, 1:], lengths)[0]
                        xe_senti_labels = xe_senti_labels.softmax(dim=-1).argmax(dim=-1).detach()
                    (s_caps, s_lengths), s_cpts, s_sentis, s_labels = s2s_batch
                    s_caps, s_cpts = ops.to_device(s_caps, device), ops.to_device(s_cpts, device)
                    s_sentis, s_labels = ops.to_device(s_sentis, device), ops.to_device(s_labels, device)
                    with cap.token_logprobs():       # (log p(target) [B,T]: the [B,T,V] log-probs are never formed)
                        pred, pred2 = cap(fc_feats, att_feats, cpts_tensor, caps_tensor, xe_senti_labels, self.xe_ss_prob,
                                          s_caps, s_cpts, s_sentis, s_labels, self.seq2seq_ss_prob, mode='xe_seq2seq')
                    xe_loss = share(self.cap_xe_crit(pred, caps_tensor[:, 1:], lengths), w_xe)
                    put('xe_loss', xe_loss)
                    seq2seq_loss = share(self.seq_flag * self.cap_xe_crit(pred2, s_caps[:, 1:], s_lengths), w_s2s)
                    put('seq2seq_loss', seq2seq_loss)
                elif data_type == 'fact':                    # XE on the ground truth, labelled by the classifier
                    with torch.no_grad():
                        xe_senti_labels = self.sent_senti_cls(caps_tensor[:, 1:], lengths)[0]
                        xe_senti_labels = xe_senti_labels.softmax(dim=-1).argmax(dim=-1).detach()
                    with cap.token_logprobs():
                        pred = cap(fc_feats, att_feats, cpts_tensor, caps_tensor, xe_senti_labels,
                                   ss_prob=self.xe_ss_prob, mode='xe')
                    xe_loss = share(self.cap_xe_crit(pred, caps_tensor[:, 1:], lengths), w_xe)
                    put('xe_loss', xe_loss)

                if training and not merged:
                    (s_caps, s_lengths), s_cpts, s_sentis, s_labels = s2s_batch
                    s_caps, s_cpts = ops.to_device(s_caps, device), ops.to_device(s_cpts, device)
                    s_sentis, s_labels = ops.to_device(s_sentis, device), ops.to_device(s_labels, device)
                    def seq2seq_unroll():                     # 80 text-only rows: a chain of small launches that
                        with cap.token_logprobs():
                            pred = cap(s_caps, s_cpts, s_sentis, s_labels, ss_prob=self.seq2seq_ss_prob, mode='seq2seq')
                        return share(self.seq_flag * self.cap_xe_crit(pred, s_caps[:, 1:], s_lengths), w_s2s)
                    # ... overlaps with the XE unroll queued above when it runs on the side stream (forward and,
                    # through autograd, backward); same numbers either way
                    seq2seq_loss = run_on_side_stream(device, seq2seq_unroll) if device.type == 'cuda' \
                        else seq2seq_unroll()
                    put('seq2seq_loss', seq2seq_loss)

                copied.synchronize()
                if not exact and getattr(cap, 'numerics_checks', True) and ops.device_status(reset=True):
                    raise OutOfDomain()      # the roll-outs met non-finite values: nothing has been updated yet
                cls_reward = get_cls_reward(sample_captions, seq_masks, greedy_captions, greedy_masks, senti_labels,
                                            self.sent_senti_cls, sample_lens=host_lens.tolist(), on_device=True)
                if data_type == 'fact':
                    fact_reward = get_self_critical_reward(
                        host_sample.numpy(), host_greedy.numpy(), fns, ground_truth, cap.sos_id, cap.eos_id,
                        self.ciderd_scorer)
                    fact_reward = ops.upload(fact_reward.astype('float32'), torch.float32, device)
                    put('fact_reward', share(fact_reward[:, 0].mean(), w_rows))
                else:
                    fact_reward = 0
                put('cls_reward', share(cls_reward.mean(-1).mean(-1), w_rows))

                rewards = fact_reward + self.cls_flag * cls_reward
                put('all_rewards', share(rewards.mean(-1).mean(-1), w_rows))
                cap_loss = share(self.cap_rl_crit(sample_logprobs, seq_masks, rewards), w_rl)
                put('cap_loss', cap_loss)

                total = cap_loss + xe_loss + da_loss + seq2seq_loss
                if training:
                    if self.dp_arena is not None:
                        self.dp_arena.zero_()
                    else:
                        self.cap_optim.zero_grad()
                    total.backward()
                    if self.dp_arena is not None:
                        self.dp_arena.all_reduce(self.dp_group)      # one 88 MB sum over xGMI, before the clamp
                    clip_gradient(self.cap_optim)            # 0.1, fused into the Adam launch
                    self.cap_optim.step()

            stats_it = []
            try:
                run_iteration(False, lambda k, v: stats_it.append((k, v)))
            except OutOfDomain:
                # features beyond |x| < 65504: the reference trains / evaluates on whatever its encoder produced
                # (models/decoder.py:86-98) - this iteration is redone on the exact-fp32 engine
                cap._warn_out_of_domain('non-finite values in a roll-out')
                del stats_it[:]
                with ops.exact_fp32_engine():
                    run_iteration(True, lambda k, v: stats_it.append((k, v)))
            for k, v in stats_it:
                add(k, v)

        if dist_on and sums:                             # global statistics: one small all-reduce per call
            keys = sorted(sums)                          # (same keys on every rank: data_type / training were checked)
            vec = torch.stack([torch.as_tensor(sums[k], dtype=torch.float32, device=device).reshape(()) for k in keys])
            dp.all_reduce_(vec, self.dp_group)
            sums = dict(zip(keys, vec.tolist()))
        if getattr(self.captioner, 'numerics_checks', True):
            ops.check_numerics('Detector.forward')       # (the statistics below are read by the host anyway)
        # len(data) is the length of the (loader, loader) TUPLE, as in the reference (decoder.py:178-179) - the same
        # on every rank, so globally summed statistics divide by the same number everywhere
        return {k: float(v) / len(data) for k, v in sums.items()}

    def invalidate_image_sentiments(self):
        """Forget the cached image-sentiment labels.  The cache is keyed by file name (+ the feature geometry, the
        threshold and the detector's weights): call this when the FEATURES behind a file name change - a second dataset
        that reuses names, features edited in place - or construct with cache_image_sentiments = False."""
        self._senti_cache, self._senti_cache_key = {}, None

    def _image_sentiments(self, fns, att_feats):
        if not self.cache_image_sentiments:
            return self.senti_detector.sample(att_feats, self.senti_threshold)[0].detach()
        key = (self.senti_threshold, tuple(att_feats.shape[1:]), att_feats.dtype) + \
            tuple((q.data_ptr(), q._version) for q in self.senti_detector.parameters())
        if key != self._senti_cache_key:                 # weights reloaded / threshold changed
            self._senti_cache, self._senti_cache_key = {}, key
        cache = self._senti_cache
        if all(fn in cache for fn in fns):
            return ops.upload([cache[fn] for fn in fns], torch.int64, att_feats.device)
        labels = self.senti_detector.sample(att_feats, self.senti_threshold)[0].detach()
        for fn, lab in zip(fns, labels.tolist()):
            cache[fn] = lab
        return labels

    def sample(self, fc_feats, att_feats, sentis_tensor, beam_size=3, decoding_constraint=1):
        """One image: detect its sentiment, then beam search (decoder.py:182-192)."""
        self.eval()
        att_feats = att_feats.unsqueeze(0)
        # the label stays on the device until the search is queued: the category name (a host read) is looked up behind the
        # search's own read-back instead of draining the device between the two (tools/eval_loop_probe.py)
        senti_label, _, _ = self.senti_detector.sample_device(att_feats, self.senti_threshold)
        captions, _ = self.captioner.sample(fc_feats, att_feats, sentis_tensor, senti_label, beam_size,
                                            decoding_constraint, self.max_seq_len)
        return captions, self.senti_detector.names(senti_label)
