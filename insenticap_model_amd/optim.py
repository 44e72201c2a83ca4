"""clip_gradient + Adam of the reference trainers (train_xe.py:19-23,191-192; decoder.py:14-18,
166-167; captioner.py:422-423) as ONE multi-tensor HIP launch instead of ~40 x (clamp_ + Adam's
per-tensor op chain).

`FusedClampAdam` subclasses torch.optim.Adam only for its bookkeeping (param_groups, state_dict
layout: 'step', 'exp_avg', 'exp_avg_sq'), so optimizer checkpoints written by the reference load
here and vice versa (train_xe.py:245). `step()` never calls torch's update.
"""
import torch

from . import ops


class FusedClampAdam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        self._pending_clip = 0.0
        self.refresh_weight_planes = True
        self.device_hyper = None        # train_graph.py: {lr, bc1, bc2_sqrt} on the device while a step is captured

    def set_clip(self, grad_clip):
        """Elementwise clamp to +-grad_clip fused into the next step() (then cleared)."""
        self._pending_clip = float(grad_clip)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        clip, self._pending_clip = self._pending_clip, 0.0
        epoch_before = ops.WEIGHT_EPOCH
        for group in self.param_groups:
            ps, gs, ms, vs = [], [], [], []
            step_no = None
            for q in group['params']:
                if q.grad is None:
                    continue
                ops.require_device(q)
                st = self.state[q]
                if len(st) == 0:
                    st['step'] = torch.tensor(0.0, dtype=torch.float32)
                    st['exp_avg'] = torch.zeros_like(q, memory_format=torch.preserve_format)
                    st['exp_avg_sq'] = torch.zeros_like(q, memory_format=torch.preserve_format)
                st['step'] += 1
                n = int(st['step'])
                if step_no is None:
                    step_no = n
                elif n != step_no:      # tensors that joined later: give them their own launch
                    self._launch([q], [q.grad], [st['exp_avg']], [st['exp_avg_sq']], group, clip, n)
                    continue
                if not q.grad.is_contiguous():
                    q.grad = q.grad.contiguous()
                ps.append(q)
                gs.append(q.grad)
                ms.append(st['exp_avg'])
                vs.append(st['exp_avg_sq'])
            if ps:
                self._launch(ps, gs, ms, vs, group, clip, step_no)
        if ops.WEIGHT_EPOCH != epoch_before and self.refresh_weight_planes:
            # the f16 weight planes the suspended weights scopes hold: re-split them now, in few batched launches,
            # instead of one launch per weight operand inside the next iteration's sweeps
            ops.refresh_weight_planes(epoch_before)
        return loss

    @torch.no_grad()
    def step_params(self, params, clip, first=True, last=True):
        """clamp + Adam on a SUBSET of the parameters (one launch): the bucketed data-parallel step (dp.GradSink.finish)
        updates each bucket as its all-reduce lands.  `first` / `last` bracket one optimizer step: the weight planes of
        the suspended scopes are refreshed once, after the last subset."""
        if first:
            self._epoch_before = ops.WEIGHT_EPOCH
        group = self.param_groups[0]
        ps, gs, ms, vs, step_no = [], [], [], [], None
        for q in params:
            if q.grad is None:
                continue
            ops.require_device(q)
            st = self.state[q]
            if len(st) == 0:
                st['step'] = torch.tensor(0.0, dtype=torch.float32)
                st['exp_avg'] = torch.zeros_like(q, memory_format=torch.preserve_format)
                st['exp_avg_sq'] = torch.zeros_like(q, memory_format=torch.preserve_format)
            st['step'] += 1
            n = int(st['step'])
            if step_no is None:
                step_no = n
            elif n != step_no:
                self._launch([q], [q.grad], [st['exp_avg']], [st['exp_avg_sq']], group, clip, n)
                continue
            ps.append(q)
            gs.append(q.grad)
            ms.append(st['exp_avg'])
            vs.append(st['exp_avg_sq'])
        if ps:
            self._launch(ps, gs, ms, vs, group, clip, step_no)
        if last and ops.WEIGHT_EPOCH != self._epoch_before and self.refresh_weight_planes:
            ops.refresh_weight_planes(self._epoch_before)

    def _launch(self, ps, gs, ms, vs, group, clip, step_no):
        b1, b2 = group['betas']
        ops.clamp_adam(ps, gs, ms, vs, group['lr'], b1, b2, group['eps'], group['weight_decay'], clip, step_no,
                       hyper=self.device_hyper)


def clip_gradient(optimizer, grad_clip=0.1):
    """Drop-in for train_xe.py:19-23 / decoder.py:14-18. With a FusedClampAdam the clamp is
    deferred into the optimizer's next step() (same result, one launch); other optimizers get an
    immediate in-place clamp through the same kernel family (clamp-only Adam call is not needed:
    isc_clamp_adam clamps in place, so a plain optimizer is handled by torch here only as plumbing)."""
    if isinstance(optimizer, FusedClampAdam):
        optimizer.set_clip(grad_clip)
        return
    for group in optimizer.param_groups:
        for prm in group['params']:
            if prm.grad is not None:
                prm.grad.data.clamp_(-grad_clip, grad_clip)
