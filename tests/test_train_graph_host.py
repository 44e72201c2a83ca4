"""Host-side contract of train_graph.XETrainGraph that needs no GPU: it is the product path (no CPU fallback), takes
the fused optimizer only, and the three scalars it hands the captured optimizer launch are the eager launch's."""
import math

import pytest
import torch

from insenticap_model_amd import Captioner, ops, synth
from insenticap_model_amd._lib import HipLibraryError
from insenticap_model_amd.train_graph import XETrainGraph


def _tiny():
    return Captioner(synth.make_idx2word(64), synth.SENTIMENT_CATEGORIES, synth.TINY_SETTINGS)


def test_refuses_a_foreign_optimizer_and_cpu_parameters():
    cap = _tiny()
    optim, xc, dc = cap.get_optim_criterion(4e-4)
    with pytest.raises(TypeError):
        XETrainGraph(cap, torch.optim.Adam(cap.parameters(), lr=4e-4), xc, dc)
    with pytest.raises(HipLibraryError):          # parameters on the CPU: there is no CPU path
        XETrainGraph(cap, optim, xc, dc)


def test_adam_hyper_is_what_the_eager_launch_derives():
    """isc_clamp_adam computes 1 - beta1^t and sqrt(1 - beta2^t) in double and rounds to float (backward.hip:
    clamp_adam_launch); ops.adam_hyper must give the same three floats for the device-resident form."""
    for step in (1, 2, 10, 1000, 123456):
        lr, b1, b2 = 4e-4, 0.9, 0.999
        got = torch.tensor(ops.adam_hyper(lr, b1, b2, step), dtype=torch.float32)
        want = torch.tensor([lr, 1.0 - math.pow(b1, float(step)), math.sqrt(1.0 - math.pow(b2, float(step)))],
                            dtype=torch.float32)
        assert torch.equal(got, want), step
