"""Action-distribution heads mirroring reference a2c_ppo_acktr/distributions.py:59-95.

The modules own the parameters under the reference's names (`dist.linear.*`, `dist.fc_mean.*`,
`dist.logstd._bias`) and are initialised by the same torch calls in the same order, so a given
seed yields the same initial weights as the reference.  Log-prob / entropy arithmetic runs in
csrc/loss.hip; the small `Fixed*` objects below only carry device tensors for `Policy.act`.
"""
import torch
import torch.nn as nn

from .utils import AddBias, init


class FixedCategorical(object):
    """Result of Categorical head for rollout-time use: sample / mode / log_probs / entropy on GPU logits."""

    def __init__(self, logits):
        self.logits = logits - logits.logsumexp(dim=-1, keepdim=True)
        self.probs = torch.softmax(self.logits, dim=-1)

    def sample(self):
        return torch.multinomial(self.probs, 1)

    def mode(self):
        return self.probs.argmax(dim=-1, keepdim=True)

    def log_probs(self, actions):
        return self.logits.gather(-1, actions.long().view(-1, 1))

    def entropy(self):
        return -(self.logits.clamp(min=torch.finfo(self.logits.dtype).min) * self.probs).sum(-1)


class FixedNormal(object):
    def __init__(self, mean, std):
        self.mean, self.stddev = mean, std

    def sample(self):
        return torch.normal(self.mean, self.stddev)

    def mode(self):
        return self.mean

    def log_probs(self, actions):
        import math
        var = self.stddev ** 2
        lp = -((actions - self.mean) ** 2) / (2 * var) - self.stddev.log() - math.log(math.sqrt(2 * math.pi))
        return lp.sum(-1, keepdim=True)

    def entropy(self):
        import math
        return (0.5 + 0.5 * math.log(2 * math.pi) + torch.log(self.stddev)).sum(-1)


class Categorical(nn.Module):
    kind = 0   # dist id of the C-ABI

    def __init__(self, num_inputs, num_outputs):
        super(Categorical, self).__init__()
        init_ = lambda m: init(m, nn.init.orthogonal_, lambda x: nn.init.constant_(x, 0), gain=0.01)
        self.linear = init_(nn.Linear(num_inputs, num_outputs))
        self.num_outputs = num_outputs


class DiagGaussian(nn.Module):
    kind = 1

    def __init__(self, num_inputs, num_outputs):
        super(DiagGaussian, self).__init__()
        init_ = lambda m: init(m, nn.init.orthogonal_, lambda x: nn.init.constant_(x, 0))
        self.fc_mean = init_(nn.Linear(num_inputs, num_outputs))
        self.logstd = AddBias(torch.zeros(num_outputs))
        self.num_outputs = num_outputs
