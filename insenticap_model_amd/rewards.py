"""RL criterion of the reference (self_critical/utils.py:169-177)."""
import torch.nn as nn


class RewardCriterion(nn.Module):
    """Masked REINFORCE loss: -sum(logp * mask * reward) / sum(mask) over [B,T] tensors.
    `seq_logprobs` comes from Captioner.forward_rl (differentiable when sampling in train mode);
    the three [B,T] operands are tiny, so this stays elementwise tensor math on the device."""

    def forward(self, seq_logprobs, seq_masks, reward):
        output = -seq_logprobs * seq_masks * reward
        return output.sum() / seq_masks.sum()
