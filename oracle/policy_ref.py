"""Oracle: policy networks + action distributions (plain torch CPU ops).

TEST INFRASTRUCTURE -- see oracle/__init__.py.

Parameters are an ordered dict ``name -> tensor`` using the reference's
``state_dict`` names so a reference ``Policy.state_dict()`` drops straight in:

NatureCNN + Categorical (reference model.py:169-195, distributions.py:59-73)
    base.main.{0,2,4}.{weight,bias}  conv 8x8s4 / 4x4s2 / 3x3s1, 32-64-32 ch
    base.main.7.{weight,bias}        Linear(32*7*7 -> hidden)
    base.critic_linear.{weight,bias} Linear(hidden -> 1)
    dist.linear.{weight,bias}        Linear(hidden -> A)
MLPBase + DiagGaussian (model.py:198-229, distributions.py:76-95)
    base.actor.{0,2}.*, base.critic.{0,2}.*, base.critic_linear.*,
    dist.fc_mean.*, dist.logstd._bias [A,1]
Recurrent bases (model.py:89-95,111-166) add, FIRST in parameter order,
    base.gru.{weight_ih_l0 [3H,I], weight_hh_l0 [3H,H], bias_ih_l0 [3H], bias_hh_l0 [3H]}  (gate order r, z, n)
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F


def gru_param_shapes(input_size, hidden):
    """nn.GRU(input_size, hidden) as registered by NNBase.__init__ (model.py:90)."""
    return [('base.gru.weight_ih_l0', (3 * hidden, input_size)), ('base.gru.weight_hh_l0', (3 * hidden, hidden)),
            ('base.gru.bias_ih_l0', (3 * hidden,)), ('base.gru.bias_hh_l0', (3 * hidden,))]


def cnn_param_shapes(num_inputs=4, hidden=512, num_actions=6, recurrent=False):
    """Order == reference ``Policy.parameters()`` order (model.py:90,176-185)."""
    return OrderedDict((gru_param_shapes(hidden, hidden) if recurrent else []) + [
        ('base.main.0.weight', (32, num_inputs, 8, 8)), ('base.main.0.bias', (32,)),
        ('base.main.2.weight', (64, 32, 4, 4)), ('base.main.2.bias', (64,)),
        ('base.main.4.weight', (32, 64, 3, 3)), ('base.main.4.bias', (32,)),
        ('base.main.7.weight', (hidden, 32 * 7 * 7)), ('base.main.7.bias', (hidden,)),
        ('base.critic_linear.weight', (1, hidden)), ('base.critic_linear.bias', (1,)),
        ('dist.linear.weight', (num_actions, hidden)), ('dist.linear.bias', (num_actions,)),
    ])


def mlp_param_shapes(num_inputs, hidden=64, num_actions=3, recurrent=False):
    """Order == reference ``Policy.parameters()`` order (model.py:90,202-216)."""
    tower_in = hidden if recurrent else num_inputs
    return OrderedDict((gru_param_shapes(num_inputs, hidden) if recurrent else []) + [
        ('base.actor.0.weight', (hidden, tower_in)), ('base.actor.0.bias', (hidden,)),
        ('base.actor.2.weight', (hidden, hidden)), ('base.actor.2.bias', (hidden,)),
        ('base.critic.0.weight', (hidden, tower_in)), ('base.critic.0.bias', (hidden,)),
        ('base.critic.2.weight', (hidden, hidden)), ('base.critic.2.bias', (hidden,)),
        ('base.critic_linear.weight', (1, hidden)), ('base.critic_linear.bias', (1,)),
        ('dist.fc_mean.weight', (num_actions, hidden)), ('dist.fc_mean.bias', (num_actions,)),
        ('dist.logstd._bias', (num_actions, 1)),
    ])


def is_cnn(P):
    return 'base.main.0.weight' in P


def is_categorical(P):
    return 'dist.linear.weight' in P


def head_param_shapes(hidden, num_actions, categorical):
    """Parameters of the action head in the reference's order (distributions.py:59-95)."""
    if categorical:
        return [('dist.linear.weight', (num_actions, hidden)), ('dist.linear.bias', (num_actions,))]
    return [('dist.fc_mean.weight', (num_actions, hidden)), ('dist.fc_mean.bias', (num_actions,)),
            ('dist.logstd._bias', (num_actions, 1))]


def with_head(shapes, hidden, num_actions, categorical):
    """A base's parameter table (cnn_param_shapes / mlp_param_shapes) with the OTHER head: the reference pairs either
    base with either action space."""
    out = OrderedDict((k, v) for k, v in shapes.items() if not k.startswith('dist.'))
    out.update(head_param_shapes(hidden, num_actions, categorical))
    return out


def is_recurrent(P):
    return 'base.gru.weight_ih_l0' in P


def gru_cell(P, x, h):
    """One nn.GRU step (torch/nn/modules/rnn.py GRU equations, gate order r,z,n; ATen's cell forms the new state as
    n + z*(h - n)):  r = s(W_ir x + b_ir + W_hr h + b_hr), z likewise, n = tanh(W_in x + b_in + r*(W_hn h + b_hn))."""
    gi = F.linear(x, P['base.gru.weight_ih_l0'], P['base.gru.bias_ih_l0'])
    gh = F.linear(h, P['base.gru.weight_hh_l0'], P['base.gru.bias_hh_l0'])
    i_r, i_z, i_n = gi.chunk(3, 1)
    h_r, h_z, h_n = gh.chunk(3, 1)
    r = torch.sigmoid(i_r + h_r)
    z = torch.sigmoid(i_z + h_z)
    n = torch.tanh(i_n + r * h_n)
    return n + z * (h - n)


def forward_gru(P, x, hxs, masks):
    """model.py:111-166.  x [T*N, I] (row t*N+n), hxs [N,H], masks [T*N,1] -> (outputs [T*N,H], final hxs [N,H]).
    The reference multiplies the carried state by masks[t] at t = 0 and at every step where some env has a zero
    mask, and runs the steps in between as one nn.GRU call; on the other steps masks[t] is all ones, so multiplying
    at EVERY step (done here) is the same arithmetic (x*1.0 is exact).  T = 1 is model.py:112-115."""
    N = hxs.size(0)
    T = x.size(0) // N
    x = x.view(T, N, x.size(1))
    masks = masks.view(T, N, 1)
    h = hxs
    outs = []
    for t in range(T):
        h = gru_cell(P, x[t], h * masks[t])
        outs.append(h)
    return torch.cat(outs, 0), h


def cnn_base(P, obs, rnn_hxs=None, masks=None):
    """model.py:189-195: returns (value [B,1], features [B,hidden], rnn_hxs)."""
    x = obs / 255.0
    x = F.relu(F.conv2d(x, P['base.main.0.weight'], P['base.main.0.bias'], stride=4))
    x = F.relu(F.conv2d(x, P['base.main.2.weight'], P['base.main.2.bias'], stride=2))
    x = F.relu(F.conv2d(x, P['base.main.4.weight'], P['base.main.4.bias'], stride=1))
    x = x.reshape(x.size(0), -1)
    x = F.relu(F.linear(x, P['base.main.7.weight'], P['base.main.7.bias']))
    if is_recurrent(P):
        x, rnn_hxs = forward_gru(P, x, rnn_hxs, masks)
    value = F.linear(x, P['base.critic_linear.weight'], P['base.critic_linear.bias'])
    return value, x, rnn_hxs


def mlp_base(P, obs, rnn_hxs=None, masks=None):
    """model.py:220-229: separate tanh towers (on the GRU output when recurrent)."""
    x = obs
    if is_recurrent(P):
        x, rnn_hxs = forward_gru(P, x, rnn_hxs, masks)
    hc = torch.tanh(F.linear(x, P['base.critic.0.weight'], P['base.critic.0.bias']))
    hc = torch.tanh(F.linear(hc, P['base.critic.2.weight'], P['base.critic.2.bias']))
    ha = torch.tanh(F.linear(x, P['base.actor.0.weight'], P['base.actor.0.bias']))
    ha = torch.tanh(F.linear(ha, P['base.actor.2.weight'], P['base.actor.2.bias']))
    value = F.linear(hc, P['base.critic_linear.weight'], P['base.critic_linear.bias'])
    return value, ha, rnn_hxs


def categorical_logp_entropy(logits_raw, actions):
    """distributions.py:18-29 over torch/distributions/categorical.py:78,150-162.

    logits -= logsumexp; log_prob = gather; entropy = -sum(clamp(logits)*softmax).
    Returns (log_probs [B,1], entropy [B]).
    """
    logits = logits_raw - logits_raw.logsumexp(dim=-1, keepdim=True)
    probs = F.softmax(logits, dim=-1)
    logp = logits.gather(-1, actions.long().view(-1, 1))
    min_real = torch.finfo(logits.dtype).min
    ent = -(torch.clamp(logits, min=min_real) * probs).sum(-1)
    return logp, ent


def normal_logp_entropy(mean, logstd_bias, actions):
    """distributions.py:36-41,86-95 over torch/distributions/normal.py:88-103,115-116.

    logstd = zeros + bias.t(); std = exp(logstd);
    log_prob = -(a-mu)^2/(2 var) - log(std) - log(sqrt(2 pi)), summed over A (keepdim);
    entropy = sum_A (0.5 + 0.5 log(2 pi) + log(std)).
    """
    logstd = torch.zeros_like(mean) + logstd_bias.t().view(1, -1)
    std = logstd.exp()
    var = std ** 2
    log_scale = std.log()
    lp = -((actions - mean) ** 2) / (2 * var) - log_scale - math.log(math.sqrt(2 * math.pi))
    ent = 0.5 + 0.5 * math.log(2 * math.pi) + torch.log(std)
    return lp.sum(-1, keepdim=True), ent.sum(-1)


def evaluate_actions(P, obs, actions, rnn_hxs=None, masks=None, return_hxs=False):
    """model.py:72-79: (value [B,1], action_log_probs [B,1], entropy.mean() scalar[, rnn_hxs])."""
    value, feat, rnn_hxs = (cnn_base if is_cnn(P) else mlp_base)(P, obs, rnn_hxs, masks)
    if is_categorical(P):            # the head follows the action space, not the base (model.py:30-38)
        logits = F.linear(feat, P['dist.linear.weight'], P['dist.linear.bias'])
        logp, ent = categorical_logp_entropy(logits, actions)
    else:
        mean = F.linear(feat, P['dist.fc_mean.weight'], P['dist.fc_mean.bias'])
        logp, ent = normal_logp_entropy(mean, P['dist.logstd._bias'], actions)
    if return_hxs:
        return value, logp, ent.mean(), rnn_hxs
    return value, logp, ent.mean()


def act_deterministic(P, obs, rnn_hxs=None, masks=None):
    """model.py:54-66 with deterministic=True: (value, mode action, its log-prob, rnn_hxs)."""
    value, feat, rnn_hxs = (cnn_base if is_cnn(P) else mlp_base)(P, obs, rnn_hxs, masks)
    if is_categorical(P):
        logits = F.linear(feat, P['dist.linear.weight'], P['dist.linear.bias'])
        action = logits.argmax(dim=-1, keepdim=True)
        logp, _ = categorical_logp_entropy(logits, action)
    else:
        action = F.linear(feat, P['dist.fc_mean.weight'], P['dist.fc_mean.bias'])
        logp, _ = normal_logp_entropy(action, P['dist.logstd._bias'], action)
    return value, action, logp, rnn_hxs
