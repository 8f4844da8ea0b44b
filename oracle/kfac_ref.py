"""Oracle: ACKTR update = A2C losses + sampled-Fisher statistics + KFAC step (torch CPU).

TEST INFRASTRUCTURE -- see oracle/__init__.py.

Restates reference a2c_ppo_acktr/algo/a2c_acktr.py:33-80 (acktr=True) and
a2c_ppo_acktr/algo/kfac.py:16-71,144-242 functionally: the reference's forward-pre / backward
hooks become explicit captures of every layer's input and of the gradient w.r.t. every layer's
output.  `torch.symeig` (removed from torch) is `torch.linalg.eigh(UPLO='U')`, ascending.

"Modules" in reference order (kfac.py:179-188 after split_bias): for each layer the bias-less
Linear/Conv2d followed by its AddBias.  State per module: m_aa, m_gg (running factors), d_a, Q_a,
d_g, Q_g (eigendecompositions, refreshed every Tf steps).
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

from . import policy_ref

CNN_LAYERS = [('base.main.0', 'conv', 4), ('base.main.2', 'conv', 2), ('base.main.4', 'conv', 1),
              ('base.main.7', 'linear', None), ('base.critic_linear', 'linear', None), ('dist.linear', 'linear', None)]
MLP_LAYERS = [('base.actor.0', 'linear', None), ('base.actor.2', 'linear', None), ('base.critic.0', 'linear', None),
              ('base.critic.2', 'linear', None), ('base.critic_linear', 'linear', None),
              ('dist.fc_mean', 'linear', None)]


def extract_patches(x, k, s):
    """kfac.py:16-26 (no padding): [B,C,H,W] -> [B,oh,ow,C*kh*kw], patch index = (c, ky, kx)."""
    x = x.unfold(2, k[0], s).unfold(3, k[1], s)           # [B,C,oh,ow,kh,kw]
    x = x.permute(0, 2, 3, 1, 4, 5).contiguous()
    return x.view(x.size(0), x.size(1), x.size(2), -1)


def cov_a(a, kind, ksize=None, stride=None, fast_cnn=False):
    """kfac.py:29-46."""
    B = a.size(0)
    if kind == 'conv':
        p = extract_patches(a, ksize, stride)
        if fast_cnn:                                       # kfac.py:33-36: one mean patch per image
            a = p.view(p.size(0), -1, p.size(-1)).mean(1)
        else:
            a = p.view(-1, p.size(-1)).div(p.size(1)).div(p.size(2))
    elif kind == 'bias':
        a = torch.ones(B, 1, dtype=a.dtype)
    return a.t() @ (a / B)


def cov_g(g, kind, fast_cnn=False):
    """kfac.py:49-64."""
    B = g.size(0)
    if kind == 'conv' and fast_cnn:                        # kfac.py:53-55: summed over space
        g = g.view(g.size(0), g.size(1), -1).sum(-1)
    elif kind == 'conv':
        g = g.permute(0, 2, 3, 1).contiguous()
        g = g.view(-1, g.size(-1)).mul(g.size(1)).mul(g.size(2))
    elif kind == 'bias':
        g = g.view(g.size(0), g.size(1), -1).sum(-1)
    g_ = g * B
    return g_.t() @ (g_ / g.size(0))


def running_stat(x, m, decay):
    """kfac.py:67-71: three separate in-place ops."""
    m *= decay / (1 - decay)
    m += x
    m *= (1 - decay)


def new_state(lr=0.25, momentum=0.9, stat_decay=0.99, kl_clip=0.001, damping=1e-2, weight_decay=0, Ts=1, Tf=10,
              fast_cnn=False):
    return dict(fast_cnn=fast_cnn, steps=0, lr=lr, momentum=momentum, stat_decay=stat_decay, kl_clip=kl_clip, damping=damping,
                weight_decay=weight_decay, Ts=Ts, Tf=Tf, m_aa=OrderedDict(), m_gg=OrderedDict(), Q_a={}, Q_g={},
                d_a={}, d_g={}, mom={})


def forward_capture(P, obs, actions):
    """Forward pass keeping, per layer, its input and its (pre-activation, post-bias) output."""
    cnn = policy_ref.is_cnn(P)
    caps = OrderedDict()

    def lin(name, x):
        z = F.linear(x, P[name + '.weight'], P[name + '.bias'])
        caps[name] = (x, z)
        return z

    if cnn:
        x = obs / 255.0
        for name, _, s in CNN_LAYERS[:3]:
            z = F.conv2d(x, P[name + '.weight'], P[name + '.bias'], stride=s)
            caps[name] = (x, z)
            x = F.relu(z)
        x = x.reshape(x.size(0), -1)
        feat = F.relu(lin('base.main.7', x))
        value = lin('base.critic_linear', feat)
    else:
        feat = torch.tanh(lin('base.actor.0', obs))
        feat = torch.tanh(lin('base.actor.2', feat))
        hc = torch.tanh(lin('base.critic.0', obs))
        hc = torch.tanh(lin('base.critic.2', hc))
        value = lin('base.critic_linear', hc)
    if policy_ref.is_categorical(P):            # the head follows the action space, not the base (model.py:30-38)
        logits = lin('dist.linear', feat)
        logp, ent = policy_ref.categorical_logp_entropy(logits, actions)
    else:
        mean = lin('dist.fc_mean', feat)
        zeros = torch.zeros_like(mean)
        logstd = zeros + P['dist.logstd._bias'].t().view(1, -1)
        caps['dist.logstd'] = (zeros, logstd)
        std = logstd.exp()
        lp = -((actions - mean) ** 2) / (2 * std ** 2) - std.log() - math.log(math.sqrt(2 * math.pi))
        logp = lp.sum(-1, keepdim=True)
        ent = (0.5 + 0.5 * math.log(2 * math.pi) + torch.log(std)).sum(-1)
    return value, logp, ent.mean(), caps


def module_list(P):
    """[(module key, kind, param name, layer name)] in the reference's module order."""
    out = []
    for name, kind, _ in layer_list(P):
        out.append((name + '#w', kind, name + '.weight', name))
        out.append((name + '#b', 'bias', name + '.bias', name))
    if not policy_ref.is_categorical(P):
        out.append(('dist.logstd#b', 'bias', 'dist.logstd._bias', 'dist.logstd'))
    return out


def layer_list(P):
    """Base layers + the head's linear (dist.linear / dist.fc_mean by action space)."""
    layers = list(CNN_LAYERS if policy_ref.is_cnn(P) else MLP_LAYERS)
    head = 'dist.linear' if policy_ref.is_categorical(P) else 'dist.fc_mean'
    return layers[:-1] + [(head, 'linear', None)]


def acktr_update(P, st, ro, value_loss_coef, entropy_coef, noise=None):
    """One A2C_ACKTR.update with acktr=True.  P updated in place.  `noise` defaults to torch.randn on the
    CPU default generator, like the reference (a2c_acktr.py:58).  Returns (losses, info)."""
    for v in P.values():
        v.requires_grad_(True)
    T, N = ro['rewards'].shape[:2]
    obs = ro['obs'][:-1].reshape(T * N, *ro['obs'].shape[2:])
    act = ro['actions'].reshape(T * N, -1)
    values, logp, ent, caps = forward_capture(P, obs, act)
    mods = module_list(P)
    layers = layer_list(P)
    geom = {n: (k, s) for n, k, s in layers}
    do_stats = st['steps'] % st['Ts'] == 0
    if do_stats:                                        # _save_input (kfac.py:144-159)
        for key, kind, pname, lname in mods:
            a_in = caps[lname][0].detach()
            if kind == 'conv':
                w = P[pname]
                aa = cov_a(a_in, 'conv', (w.size(2), w.size(3)), geom[lname][1], st.get('fast_cnn', False))
            else:
                aa = cov_a(a_in, kind)
            if st['steps'] == 0:
                st['m_aa'][key] = aa.clone()
            running_stat(aa, st['m_aa'][key], st['stat_decay'])
    values3 = values.view(T, N, 1)
    logp3 = logp.view(T, N, 1)
    adv = ro['returns'][:-1] - values3
    value_loss = adv.pow(2).mean()
    action_loss = -(adv.detach() * logp3).mean()
    if do_stats:                                        # a2c_acktr.py:53-68 + _save_grad_output (kfac.py:161-177)
        pg_fisher = -logp3.mean()
        if noise is None:
            noise = torch.randn(values3.size())
        sample_values = values3 + noise
        vf_fisher = -(values3 - sample_values.detach()).pow(2).mean()
        outs = [caps[l][1] for l in caps]
        gouts = torch.autograd.grad(pg_fisher + vf_fisher, outs, retain_graph=True)
        gmap = dict(zip(caps.keys(), gouts))
        for key, kind, pname, lname in mods:
            gg = cov_g(gmap[lname].detach(), kind, st.get('fast_cnn', False))
            if st['steps'] == 0:
                st['m_gg'][key] = gg.clone()
            running_stat(gg, st['m_gg'][key], st['stat_decay'])
    names = list(P.keys())
    grads = torch.autograd.grad(value_loss * value_loss_coef + action_loss - ent * entropy_coef, [P[k] for k in names])
    G = OrderedDict((k, g.clone()) for k, g in zip(names, grads))
    info = kfac_step(P, G, st, mods)
    return (value_loss.item(), action_loss.item(), ent.item()), info


def kfac_step(P, G, st, mods):
    """kfac.py:190-242."""
    if st['weight_decay'] > 0:
        for k in P:
            G[k].add_(P[k].data, alpha=st['weight_decay'])
    la = st['damping'] + st['weight_decay']
    updates = {}
    for key, kind, pname, lname in mods:
        if st['steps'] % st['Tf'] == 0:
            st['d_a'][key], st['Q_a'][key] = torch.linalg.eigh(st['m_aa'][key], UPLO='U')
            st['d_g'][key], st['Q_g'][key] = torch.linalg.eigh(st['m_gg'][key], UPLO='U')
            st['d_a'][key].mul_((st['d_a'][key] > 1e-6).float())
            st['d_g'][key].mul_((st['d_g'][key] > 1e-6).float())
        g = G[pname]
        gm = g.view(g.size(0), -1)
        v1 = st['Q_g'][key].t() @ gm @ st['Q_a'][key]
        v2 = v1 / (st['d_g'][key].unsqueeze(1) * st['d_a'][key].unsqueeze(0) + la)
        v = st['Q_g'][key] @ v2 @ st['Q_a'][key].t()
        updates[pname] = v.view(g.size())
    vg_sum = 0
    for k in P:
        vg_sum += (updates[k] * G[k] * st['lr'] * st['lr']).sum()
    nu = min(1, math.sqrt(st['kl_clip'] / vg_sum))
    lr_sgd = st['lr'] * (1 - st['momentum'])
    final = OrderedDict()
    for k in P:
        gk = updates[k] * nu
        final[k] = gk.clone()
        if k not in st['mom']:                          # torch/optim/sgd.py:358-380
            st['mom'][k] = gk.clone()
        else:
            st['mom'][k].mul_(st['momentum']).add_(gk)
        P[k].data.add_(st['mom'][k], alpha=-lr_sgd)
    st['steps'] += 1
    return dict(nu=float(nu), vg_sum=float(vg_sum), grads=final)
