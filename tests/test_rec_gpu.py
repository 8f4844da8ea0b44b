"""Recurrent (GRU) path on the MI355X: HIP kernels through the C-ABI vs the goldens generated
from the reference (tests/golden/make_golden_rec.py) and vs the oracle on the same inputs.
Bars: losses / gradients 1e-4 relative (north_star); the generator's gathers are exact."""
from collections import OrderedDict

import numpy as np
import pytest
import torch

import synth
from helpers import (A2C_HP, REC_A2C_CASES, REC_EVAL_CASES, REC_PPO_CASES, check_cnn_grads, check_summary, golden,
                     rec_eval_inputs, rec_rollouts, rec_shapes_for, t)

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


class Discrete(object):
    def __init__(self, n):
        self.n = n


class Box(object):
    def __init__(self, dim):
        self.shape = (dim,)


def make_policy(kind, od, A, salt):
    from a2c_ppo_acktr.model import Policy
    P = synth.make_params(rec_shapes_for(kind, od, A), salt)
    if kind == 'cnn':
        pol = Policy((4, 84, 84), Discrete(A), base_kwargs={'recurrent': True})
    else:
        pol = Policy((od,), Box(A), base_kwargs={'recurrent': True})
    assert [n for n, _ in pol.named_parameters()] == list(P.keys())
    pol.load_state_dict(OrderedDict((k, t(v)) for k, v in P.items()), strict=True)
    pol.to(DEV)
    return pol


def make_storage(ro, kind, od, A):
    from a2c_ppo_acktr.storage import RolloutStorage
    T, N = ro['rewards'].shape[:2]
    obs_shape = (4, 84, 84) if kind == 'cnn' else (od,)
    rs = RolloutStorage(T, N, obs_shape, Discrete(A) if kind == 'cnn' else Box(A), 512 if kind == 'cnn' else 64)
    for k in ('obs', 'rewards', 'value_preds', 'returns', 'action_log_probs', 'actions', 'masks', 'bad_masks',
              'recurrent_hidden_states'):
        getattr(rs, k).copy_(t(ro[k]))
    rs.to(DEV)
    return rs


@pytest.mark.parametrize('ci', range(len(REC_EVAL_CASES)))
def test_rec_evaluate_actions_and_grads(ci):
    tag, kind, od, A, T, N = REC_EVAL_CASES[ci]
    G = golden('rec')
    k = 'eval_' + tag
    pol = make_policy(kind, od, A, 140 + ci)
    x, ro = rec_eval_inputs(ci)
    d = lambda a: t(a).to(DEV)
    clip, c_e = (0.1, 0.01) if kind == 'cnn' else (0.2, 0.0)
    values, logp, ent, hout = pol.evaluate_actions(d(x['obs']), d(x['hxs']), d(x['masks']), d(x['act']))
    np.testing.assert_allclose(values.detach().cpu().numpy(), G[k + '/values'], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(logp.detach().cpu().numpy(), G[k + '/logp'], rtol=1e-4, atol=1e-5)
    assert abs(ent.item() - float(G[k + '/entropy'])) < 1e-4 * abs(float(G[k + '/entropy']))
    check_summary(G, k + '/hxs_out', hout.detach().cpu().numpy(), 1e-4, tag)
    olp, adv, vp, ret = d(G[k + '/old_logp']), d(x['adv']), d(x['vp']), d(x['ret'])
    ratio = torch.exp(logp - olp)
    action_loss = -torch.min(ratio * adv, torch.clamp(ratio, 1.0 - clip, 1.0 + clip) * adv).mean()
    vpc = vp + (values - vp).clamp(-clip, clip)
    value_loss = 0.5 * torch.max((values - ret).pow(2), (vpc - ret).pow(2)).mean()
    assert abs(value_loss.item() - float(G[k + '/value_loss'])) < 1e-4 * abs(float(G[k + '/value_loss']))
    assert abs(action_loss.item() - float(G[k + '/action_loss'])) < 1e-4 * abs(float(G[k + '/action_loss'])) + 1e-6
    pol.zero_grad()
    (value_loss * 0.5 + action_loss - ent * c_e).backward()
    grads = OrderedDict((n, p.grad.detach().cpu().numpy()) for n, p in pol.named_parameters())
    if kind == 'cnn':       # golden at 1e-4, else the gate-aware fp64 check (undetermined ReLU gates, see helpers)
        from a2c_ppo_acktr import engine
        with torch.no_grad():
            acts = engine.forward(pol, d(x['obs']), rnn_hxs=d(x['hxs']), masks=d(x['masks']))
        gates = [(v > 0).cpu().numpy() for v in (acts.y1, acts.y2, acts.y3, acts.feat)]
        P = synth.make_params(rec_shapes_for(kind, od, A), 140 + ci)
        how = check_cnn_grads(G, k, P, gates, x, G[k + '/old_logp'], clip, c_e, grads, 1e-4)
        print(k, 'gradient parity via', how)
    else:
        for n, g_ in grads.items():
            check_summary(G, k + '/grad/' + n, g_, 1e-4, tag)
    # rollout side: single-step form
    v1, a1, lp1, h1 = pol.act(d(ro['obs'][1]), d(ro['recurrent_hidden_states'][1]), d(ro['masks'][1]), deterministic=True)
    np.testing.assert_allclose(v1.cpu().numpy(), G[k + '/act/value'], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(a1.cpu().numpy(), G[k + '/act/action'], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(lp1.cpu().numpy(), G[k + '/act/logp'], rtol=1e-4, atol=1e-5)
    check_summary(G, k + '/act/hxs', h1.cpu().numpy(), 1e-4, tag)
    v2 = pol.get_value(d(ro['obs'][2]), d(ro['recurrent_hidden_states'][2]), d(ro['masks'][2]))
    np.testing.assert_allclose(v2.cpu().numpy(), G[k + '/get_value'], rtol=1e-4, atol=1e-5)


def test_rec_generator_matches_reference():
    G = golden('rec')
    T, N = 6, 6
    ro = rec_rollouts('mlp', 11, 3, T, N, 2100)
    rs = make_storage(ro, 'mlp', 11, 3)
    rs.compute_returns(t(ro['next_value']).to(DEV), True, 0.99, 0.95, True)
    adv = rs.normalized_advantages()
    torch.manual_seed(2024)
    for e in range(2):
        nb = 0
        for b, s in enumerate(rs.recurrent_generator(adv, 3)):
            nb += 1
            for name, v in zip(('obs', 'hxs', 'act', 'vp', 'ret', 'masks', 'olp', 'adv'), s):
                want = G['gen/e%d_b%d/%s' % (e, b, name)]
                if name in ('adv', 'ret'):
                    np.testing.assert_allclose(v.cpu().numpy(), want, rtol=2e-6, atol=1e-6)
                else:
                    assert np.array_equal(v.cpu().numpy(), want), name
        assert nb == 3
    with pytest.raises(AssertionError):
        next(rs.recurrent_generator(adv, 7))


def _ppo_storage(tag, kind, od, A, T, N, ci, G):
    ro = rec_rollouts(kind, od, A, T, N, 2200 + ci)
    last = ro['value_preds'][-1].copy()
    ro['action_log_probs'] = G['ppo_' + tag + '/action_log_probs'].copy()
    ro['value_preds'] = G['ppo_' + tag + '/value_preds'].copy()
    ro['value_preds'][-1] = last
    rs = make_storage(ro, kind, od, A)
    rs.compute_returns(t(ro['next_value']).to(DEV), True, 0.99, 0.95, True)
    return rs


@pytest.mark.parametrize('ci', range(len(REC_PPO_CASES)))
def test_rec_ppo_update(ci):
    from a2c_ppo_acktr import algo
    tag, kind, od, A, T, N, hp = REC_PPO_CASES[ci]
    G = golden('rec')
    pol = make_policy(kind, od, A, 160 + ci)
    rs = _ppo_storage(tag, kind, od, A, T, N, ci, G)
    agent = algo.PPO(pol, **hp)
    for it in range(2):
        torch.manual_seed(5321 + it)
        res = agent.update(rs)
        np.testing.assert_allclose(np.array(res), G['ppo_%s/update%d/losses' % (tag, it)], rtol=1e-4, atol=1e-6)
        for n, p in pol.named_parameters():
            check_summary(G, 'ppo_%s/update%d/param/%s' % (tag, it, n), p.detach().cpu().numpy(), 1e-4, tag)


@pytest.mark.parametrize('ci', range(len(REC_A2C_CASES)))
def test_rec_a2c_update(ci):
    from a2c_ppo_acktr import algo
    tag, kind, od, A, T, N = REC_A2C_CASES[ci]
    G = golden('rec')
    pol = make_policy(kind, od, A, 180 + ci)
    ro = rec_rollouts(kind, od, A, T, N, 2300 + ci)
    rs = make_storage(ro, kind, od, A)
    rs.compute_returns(t(ro['next_value']).to(DEV), False, 0.99, 0.95, False)
    agent = algo.A2C_ACKTR(pol, acktr=False, **A2C_HP)
    for it in range(2):
        res = agent.update(rs)
        np.testing.assert_allclose(np.array(res), G['a2c_%s/update%d/losses' % (tag, it)], rtol=1e-4, atol=1e-6)
        for n, p in pol.named_parameters():
            check_summary(G, 'a2c_%s/update%d/param/%s' % (tag, it, n), p.detach().cpu().numpy(), 1e-4, tag)


def test_rec_acktr_is_refused():
    from a2c_ppo_acktr import algo
    pol = make_policy('mlp', 11, 3, 5)
    with pytest.raises(NotImplementedError):
        algo.A2C_ACKTR(pol, 0.5, 0.01, acktr=True)


def test_rec_long_sequence_matches_oracle():
    """T = 128 steps x 4 envs on the 512-wide GRU (the Atari PPO shape per minibatch) against the oracle: the error
    must not grow along the sequence."""
    from oracle import policy_ref
    from helpers import params_torch
    kind, od, A, T, N = 'mlp', 24, 4, 128, 4
    P = synth.make_params(rec_shapes_for(kind, od, A), 77)
    from a2c_ppo_acktr.model import Policy
    pol = Policy((od,), Box(A), base_kwargs={'recurrent': True})
    pol.load_state_dict(OrderedDict((k, t(v)) for k, v in P.items()), strict=True)
    pol.to(DEV)
    obs = synth.normalish((T * N, od), 78)
    act = synth.normalish((T * N, A), 79)
    hxs = synth.uniform((N, 64), 80, -0.9, 0.9)
    masks = synth.bernoulli((T * N, 1), 81, 0.9)
    Pt = params_torch(P)
    for v in Pt.values():
        v.requires_grad_(True)
    v0, lp0, e0, h0 = policy_ref.evaluate_actions(Pt, t(obs), t(act), t(hxs), t(masks), return_hxs=True)
    (v0.pow(2).mean() + lp0.mean() + e0).backward()
    d = lambda a: t(a).to(DEV)
    v1, lp1, e1, h1 = pol.evaluate_actions(d(obs), d(hxs), d(masks), d(act))
    pol.zero_grad()
    (v1.pow(2).mean() + lp1.mean() + e1).backward()
    np.testing.assert_allclose(v1.detach().cpu().numpy(), v0.detach().numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(h1.detach().cpu().numpy(), h0.detach().numpy(), rtol=1e-4, atol=1e-5)
    for n, p in pol.named_parameters():
        a, b = p.grad.detach().cpu().numpy().reshape(-1), Pt[n].grad.numpy().reshape(-1)
        err = np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)
        assert err < 1e-4, '%s grad rel err %g' % (n, err)
