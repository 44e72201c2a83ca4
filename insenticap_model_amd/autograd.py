"""Backward pass of the decode path (BPTT through the HIP kernels). Filled in by the
training milestone; until then every entry point fails loudly - there is no silent
fallback to stock torch autograd."""


def _nyi(what):
    raise NotImplementedError(
        '%s with gradients is not implemented yet in insenticap_model_amd; run under torch.no_grad() '
        'for inference' % what)


def xe_with_grad(*a, **k):
    _nyi('forward_xe / forward_seq2seq')


def rollout_with_grad(*a, **k):
    _nyi('sampled forward_rl')


def xe_criterion_with_grad(*a, **k):
    _nyi('XECriterion')
