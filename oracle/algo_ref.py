"""Oracle: PPO / A2C update, minibatch sampler, clip-norm, Adam / RMSprop (torch CPU).

TEST INFRASTRUCTURE -- see oracle/__init__.py.

Gradients come from torch autograd over oracle/policy_ref.py (what the
reference does); the optimiser arithmetic is restated explicitly from the
installed torch formulas the reference reaches:
  torch/optim/adam.py:457,476,529-547      (single-tensor Adam)
  torch/optim/rmsprop.py:316-339           (single-tensor RMSprop, not centred)
  torch/nn/utils/clip_grad.py:102-108,165-169 (norm of per-tensor norms, coef clamp)
"""
import math
from collections import OrderedDict

import torch

from . import policy_ref


# --------------------------------------------------------------------------
# sampler: storage.py:107-143
# --------------------------------------------------------------------------
def minibatch_indices(batch_size, num_mini_batch=None, mini_batch_size=None):
    """BatchSampler(SubsetRandomSampler(range(n)), mbs, drop_last=True).

    One ``torch.randperm(n)`` on the CPU default generator per call
    (torch/utils/data/sampler.py:205-207); trailing partial batch dropped.
    """
    if mini_batch_size is None:
        assert batch_size >= num_mini_batch
        mini_batch_size = batch_size // num_mini_batch
    perm = torch.randperm(batch_size).tolist()
    nb = batch_size // mini_batch_size
    return [perm[i * mini_batch_size:(i + 1) * mini_batch_size] for i in range(nb)]


def gather_minibatch(ro, indices, advantages):
    """storage.py:127-140 (recurrent_hidden_states / masks are ignored downstream)."""
    idx = torch.as_tensor(indices, dtype=torch.long)
    flat = lambda t: t.reshape(-1, *t.shape[2:])
    obs = flat(ro['obs'][:-1])[idx]
    actions = flat(ro['actions'])[idx]
    vp = flat(ro['value_preds'][:-1])[idx]
    ret = flat(ro['returns'][:-1])[idx]
    old_lp = flat(ro['action_log_probs'])[idx]
    adv = None if advantages is None else advantages.reshape(-1, 1)[idx]
    return obs, actions, vp, ret, old_lp, adv


def recurrent_env_batches(num_processes, num_mini_batch):
    """storage.py:146-153: one randperm over ENVS per call; consecutive groups of N // num_mini_batch envs."""
    assert num_processes >= num_mini_batch
    per = num_processes // num_mini_batch
    perm = torch.randperm(num_processes)
    return [perm[s:s + per] for s in range(0, num_processes, per)]


def gather_recurrent(ro, envs, advantages):
    """storage.py:154-202: whole trajectories of the chosen envs, flattened to row t*Nb + j."""
    envs = torch.as_tensor(envs, dtype=torch.long)
    flat = lambda t: t.reshape(-1, *t.shape[2:])
    obs = flat(ro['obs'][:-1][:, envs])
    hxs = ro['recurrent_hidden_states'][0, envs]
    actions = flat(ro['actions'][:, envs])
    vp = flat(ro['value_preds'][:-1][:, envs])
    ret = flat(ro['returns'][:-1][:, envs])
    masks = flat(ro['masks'][:-1][:, envs])
    old_lp = flat(ro['action_log_probs'][:, envs])
    adv = flat(advantages[:, envs])
    return obs, hxs, actions, vp, ret, masks, old_lp, adv


# --------------------------------------------------------------------------
# optimiser pieces
# --------------------------------------------------------------------------
def clip_grad_norm_(grads, max_norm):
    """clip_grad.py: total = ||(||g_i||)||_2 ; coef = clamp(max/(total+1e-6), max=1)."""
    norms = [torch.linalg.vector_norm(g, 2.0) for g in grads]
    total = torch.linalg.vector_norm(torch.stack(norms), 2.0)
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in grads:
        g.mul_(coef)
    return total


def adam_init(P):
    return {'step': 0,
            'exp_avg': OrderedDict((k, torch.zeros_like(v)) for k, v in P.items()),
            'exp_avg_sq': OrderedDict((k, torch.zeros_like(v)) for k, v in P.items())}


def adam_step(P, G, st, lr, eps, beta1=0.9, beta2=0.999):
    st['step'] += 1
    step = st['step']
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    step_size = lr / bc1
    bc2_sqrt = bc2 ** 0.5
    for k in P:
        m, v, g = st['exp_avg'][k], st['exp_avg_sq'][k], G[k]
        m.lerp_(g, 1 - beta1)
        v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
        denom = (v.sqrt() / bc2_sqrt).add_(eps)
        P[k].data.addcdiv_(m, denom, value=-step_size)


def rmsprop_init(P):
    return {'square_avg': OrderedDict((k, torch.zeros_like(v)) for k, v in P.items())}


def rmsprop_step(P, G, st, lr, eps, alpha):
    for k in P:
        sq, g = st['square_avg'][k], G[k]
        sq.mul_(alpha).addcmul_(g, g, value=1 - alpha)
        avg = sq.sqrt().add_(eps)
        P[k].data.addcdiv_(g, avg, value=-lr)


# --------------------------------------------------------------------------
# losses
# --------------------------------------------------------------------------
def ppo_losses(values, logp, entropy, old_values, returns, old_logp, adv,
               clip_param, use_clipped_value_loss=True):
    """ppo.py:61-77."""
    ratio = torch.exp(logp - old_logp)
    surr1 = ratio * adv
    surr2 = torch.clamp(ratio, 1.0 - clip_param, 1.0 + clip_param) * adv
    action_loss = -torch.min(surr1, surr2).mean()
    if use_clipped_value_loss:
        vpc = old_values + (values - old_values).clamp(-clip_param, clip_param)
        vl = (values - returns).pow(2)
        vlc = (vpc - returns).pow(2)
        value_loss = 0.5 * torch.max(vl, vlc).mean()
    else:
        value_loss = 0.5 * (returns - values).pow(2).mean()
    return value_loss, action_loss, entropy


def normalize_advantages(ro):
    """ppo.py:35-37 (fp32 torch ops, unbiased std)."""
    adv = ro['returns'][:-1] - ro['value_preds'][:-1]
    return (adv - adv.mean()) / (adv.std() + 1e-5)


def _grads(P, loss):
    names = list(P.keys())
    gs = torch.autograd.grad(loss, [P[k] for k in names])
    return OrderedDict((k, g.clone()) for k, g in zip(names, gs))


def ppo_minibatch_grads(P, obs, actions, old_values, returns, old_logp, adv,
                        clip_param, value_loss_coef, entropy_coef,
                        use_clipped_value_loss=True, rnn_hxs=None, masks=None):
    """One forward+backward of ppo.py:57-81; returns (losses triple, grads dict)."""
    for v in P.values():
        v.requires_grad_(True)
    values, logp, ent = policy_ref.evaluate_actions(P, obs, actions, rnn_hxs, masks)
    vl, al, en = ppo_losses(values, logp, ent, old_values, returns, old_logp, adv,
                            clip_param, use_clipped_value_loss)
    G = _grads(P, vl * value_loss_coef + al - en * entropy_coef)
    return (vl.item(), al.item(), en.item()), G


def ppo_update(P, opt, ro, clip_param, ppo_epoch, num_mini_batch, value_loss_coef,
               entropy_coef, lr, eps, max_grad_norm, use_clipped_value_loss=True):
    """ppo.py:34-96.  ``P`` updated in place; ``opt`` from adam_init()."""
    adv = normalize_advantages(ro)
    T, N = ro['rewards'].shape[:2]
    sums = [0.0, 0.0, 0.0]
    for _ in range(ppo_epoch):
        if policy_ref.is_recurrent(P):                                  # ppo.py:44-46
            batches = [gather_recurrent(ro, envs, adv) for envs in recurrent_env_batches(N, num_mini_batch)]
        else:
            batches = [gather_minibatch(ro, idx, adv) for idx in minibatch_indices(T * N, num_mini_batch)]
        for b in batches:
            if policy_ref.is_recurrent(P):
                obs, hxs, act, vp, ret, masks, olp, a = b
            else:
                (obs, act, vp, ret, olp, a), hxs, masks = b, None, None
            losses, G = ppo_minibatch_grads(P, obs, act, vp, ret, olp, a, clip_param,
                                            value_loss_coef, entropy_coef,
                                            use_clipped_value_loss, hxs, masks)
            clip_grad_norm_(list(G.values()), max_grad_norm)
            adam_step(P, G, opt, lr, eps)
            for i in range(3):
                sums[i] += losses[i]
    n = ppo_epoch * num_mini_batch
    return tuple(s / n for s in sums)


def a2c_grads(P, ro, value_loss_coef, entropy_coef):
    """a2c_acktr.py:33-51,70-72 (A2C branch): full-batch forward, loss, grads."""
    for v in P.values():
        v.requires_grad_(True)
    T, N = ro['rewards'].shape[:2]
    obs = ro['obs'][:-1].reshape(T * N, *ro['obs'].shape[2:])
    act = ro['actions'].reshape(T * N, -1)
    hxs = masks = None
    if policy_ref.is_recurrent(P):                                      # a2c_acktr.py:38-43
        hxs = ro['recurrent_hidden_states'][0]
        masks = ro['masks'][:-1].reshape(T * N, 1)
    values, logp, ent = policy_ref.evaluate_actions(P, obs, act, hxs, masks)
    values = values.view(T, N, 1)
    logp = logp.view(T, N, 1)
    adv = ro['returns'][:-1] - values
    value_loss = adv.pow(2).mean()
    action_loss = -(adv.detach() * logp).mean()
    G = _grads(P, value_loss * value_loss_coef + action_loss - ent * entropy_coef)
    return (value_loss.item(), action_loss.item(), ent.item()), G


def a2c_update(P, opt, ro, value_loss_coef, entropy_coef, lr, eps, alpha, max_grad_norm):
    """a2c_acktr.py:33-80 with acktr=False; ``opt`` from rmsprop_init()."""
    losses, G = a2c_grads(P, ro, value_loss_coef, entropy_coef)
    clip_grad_norm_(list(G.values()), max_grad_norm)
    rmsprop_step(P, G, opt, lr, eps, alpha)
    return losses
