"""Checkpoint round trip the way the reference saves and loads policies: torch.save([actor_critic, obs_rms]) pickles the
whole nn.Module (reference main.py:165-176), enjoy.py:52-54 loads it back.  The flat parameter buffer and the offset
table of a2c_ppo_acktr.model.Policy are derived state and must be rebuilt on load (they are keyed by id())."""
import io
from collections import OrderedDict

import numpy as np
import pytest
import torch


class Discrete(object):
    def __init__(self, n):
        self.n = n


class Box(object):
    def __init__(self, d):
        self.shape = (d,)


def _roundtrip(obj):
    buf = io.BytesIO()
    torch.save(obj, buf)
    buf.seek(0)
    return torch.load(buf, weights_only=False)          # our own file, written two lines above


def _check_flat(pol):
    assert set(pol._offsets.keys()) == set(id(p) for p in pol.parameters())
    base = pol.flat_params.data_ptr()
    for p in pol.parameters():
        o, n = pol._offsets[id(p)]
        assert p.data.data_ptr() == base + 4 * o and p.numel() == n
    before = [p.detach().clone() for p in pol.parameters()]
    pol.flat_params.mul_(2.0)                           # parameters are views of the flat buffer (x2, x0.5: exact)
    for p, b in zip(pol.parameters(), before):
        assert torch.equal(p.detach(), b * 2.0)
    pol.flat_params.mul_(0.5)


@pytest.mark.parametrize('kind', ['cnn', 'mlp', 'mlp_rec', 'cnn_kfac'])
def test_policy_pickle_roundtrip_cpu(kind):
    from a2c_ppo_acktr.model import Policy
    torch.manual_seed(3)
    if kind.startswith('cnn'):
        pol = Policy((4, 84, 84), Discrete(6))
    else:
        pol = Policy((11,), Box(3), base_kwargs={'recurrent': kind == 'mlp_rec'})
    if kind == 'cnn_kfac':                              # after KFAC's split_bias surgery (parameters renamed)
        from a2c_ppo_acktr.algo.kfac import KFACOptimizer
        KFACOptimizer(pol)
    pol2, rms = _roundtrip([pol, None])
    assert rms is None and type(pol2) is type(pol)
    sd, sd2 = pol.state_dict(), pol2.state_dict()
    assert list(sd.keys()) == list(sd2.keys())
    for k in sd:
        assert torch.equal(sd[k], sd2[k]), k
    _check_flat(pol2)
    _check_flat(pol)


@pytest.mark.gpu
@pytest.mark.parametrize('kind', ['cnn', 'mlp'])
def test_policy_pickle_roundtrip_gpu(kind):
    """Saved from the GPU after an update, loaded back: same actions, values and log-probs, and it keeps training."""
    from a2c_ppo_acktr import algo
    from a2c_ppo_acktr.model import Policy
    from a2c_ppo_acktr.storage import RolloutStorage
    torch.manual_seed(4)
    cnn = kind == 'cnn'
    obs_shape, space = ((4, 84, 84), Discrete(6)) if cnn else ((11,), Box(3))
    pol = Policy(obs_shape, space)
    pol.to('cuda')
    T, N = 4, 4
    rs = RolloutStorage(T, N, obs_shape, space, 1)
    g = torch.Generator().manual_seed(1)
    rs.obs.copy_(torch.randint(0, 256, rs.obs.shape, generator=g).float() if cnn else torch.randn(rs.obs.shape, generator=g))
    rs.actions.copy_(torch.randint(0, 6, rs.actions.shape, generator=g) if cnn else torch.randn(rs.actions.shape, generator=g))
    rs.rewards.copy_(torch.randn(rs.rewards.shape, generator=g))
    rs.action_log_probs.fill_(-1.5)
    rs.to('cuda')
    rs.compute_returns(torch.zeros(N, 1, device='cuda'), True, 0.99, 0.95, False)
    agent = algo.PPO(pol, 0.1, 1, 2, 0.5, 0.01, lr=1e-3, eps=1e-5, max_grad_norm=0.5)
    torch.manual_seed(5)
    agent.update(rs)
    pol2, _ = _roundtrip([pol, None])
    assert next(pol2.parameters()).is_cuda
    _check_flat(pol2)
    x = rs.obs[0]
    r1, r1b, r2 = (p_.act(x, None, None, deterministic=True) for p_ in (pol, pol, pol2))
    assert torch.equal(pol.flat_params, pol2.flat_params)
    for name, a, a2, b in zip(('value', 'action', 'logp'), r1, r1b, r2):
        print(name, 'repeat max diff', float((a.double() - a2.double()).abs().max()), 'reloaded max diff',
              float((a.double() - b.double()).abs().max()))
    for a, b in zip(r1, r2):
        if torch.is_tensor(a):
            assert torch.equal(a, b)
    agent2 = algo.PPO(pol2, 0.1, 1, 2, 0.5, 0.01, lr=1e-3, eps=1e-5, max_grad_norm=0.5)
    torch.manual_seed(6)
    out = agent2.update(rs)
    assert all(np.isfinite(v) for v in out)
