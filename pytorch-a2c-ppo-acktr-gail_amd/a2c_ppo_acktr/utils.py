"""Host-side helpers mirroring reference a2c_ppo_acktr/utils.py:31-56 (AddBias, init, lr schedule).

The env-side helpers of the reference file (get_render_func, get_vec_normalize, cleanup_log_dir)
belong to the training driver and are out of this build's scope.
"""
import torch
import torch.nn as nn


class AddBias(nn.Module):
    """Bias as a module of its own (reference utils.py:32-43), parameter `_bias` of shape [C, 1]."""

    def __init__(self, bias):
        super(AddBias, self).__init__()
        self._bias = nn.Parameter(bias.unsqueeze(1))

    def forward(self, x):
        raise RuntimeError('AddBias is a parameter container here; the arithmetic runs in the fused HIP layers')


def update_linear_schedule(optimizer, epoch, total_num_epochs, initial_lr):
    """Decreases the learning rate linearly (reference utils.py:46-50)."""
    lr = initial_lr - (initial_lr * (epoch / float(total_num_epochs)))
    for param_group in optimizer.param_groups:
        param_group['lr'] = lr


def init(module, weight_init, bias_init, gain=1):
    """reference utils.py:53-56 -- same torch initialisers, same RNG consumption."""
    weight_init(module.weight.data, gain=gain)
    bias_init(module.bias.data)
    return module
