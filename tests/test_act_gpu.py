"""GPU: rollout-side API (Policy.act / get_value, reference model.py:54-70) on the HIP path and the two base x head
pairings the Atari / MuJoCo configurations do not use (CNNBase + Box, MLPBase + Discrete, model.py:20-38), against
goldens from the reference (tests/golden/make_golden_act.py -> act.npz) and the CPU oracle.
Tolerance: north_star, 1e-4 relative on values, log-probs, losses and gradients; mode actions exact."""
from collections import OrderedDict

import numpy as np
import pytest
import torch

import synth
from helpers import ACT_CASES, act_case_inputs, check_summary, golden, params_torch, rel_err, t

pytestmark = pytest.mark.gpu
TOL = 1e-4


class Discrete(object):
    def __init__(self, n):
        self.n = n


class Box(object):
    def __init__(self, d):
        self.shape = (d,)


def make_policy(ci):
    from a2c_ppo_acktr.model import Policy
    tag, kind, od, cat, A, B = ACT_CASES[ci]
    P, obs, ro = act_case_inputs(ci)
    pol = Policy((4, 84, 84) if kind == 'cnn' else (od,), Discrete(A) if cat else Box(A))
    assert [n for n, _ in pol.named_parameters()] == list(P.keys()), 'parameter names / order differ from the reference'
    pol.load_state_dict(OrderedDict((k, t(v)) for k, v in P.items()), strict=True)
    pol.to('cuda')
    return pol, P, obs, ro


@pytest.mark.parametrize('ci', range(len(ACT_CASES)))
def test_act_deterministic_and_get_value_vs_reference(ci):
    tag, kind, od, cat, A, B = ACT_CASES[ci]
    G = golden('act')
    pol, P, obs, ro = make_policy(ci)
    x = t(obs).cuda()
    value, action, logp, hxs = pol.act(x, None, None, deterministic=True)
    assert value.shape == (B, 1) and logp.shape == (B, 1)
    np.testing.assert_allclose(value.cpu().numpy(), G[tag + '/act/value'], rtol=TOL, atol=2e-5)
    np.testing.assert_allclose(pol.get_value(x, None, None).cpu().numpy(), G[tag + '/get_value'], rtol=TOL, atol=2e-5)
    if cat:
        assert action.dtype == torch.int64 and action.shape == (B, 1)
        assert np.array_equal(action.cpu().numpy(), G[tag + '/act/action'])
    else:
        assert action.shape == (B, A)
        np.testing.assert_allclose(action.cpu().numpy(), G[tag + '/act/action'], rtol=TOL, atol=2e-5)
    np.testing.assert_allclose(logp.cpu().numpy(), G[tag + '/act/logp'], rtol=TOL, atol=2e-5)


@pytest.mark.parametrize('ci', range(len(ACT_CASES)))
def test_act_sampling(ci):
    """Sampled actions: the returned log-prob is evaluate_actions' log-prob of that action; a Gaussian sample is
    mean + std * (the device generator's next normal draw); categorical samples follow softmax(logits)."""
    tag, kind, od, cat, A, B = ACT_CASES[ci]
    pol, P, obs, ro = make_policy(ci)
    pol.sample_stream = 'device'                 # (the reference-stream mode has its own test below)
    x = t(obs).cuda()
    torch.manual_seed(11)
    value, action, logp, _ = pol.act(x, None, None)
    v2, lp2, ent, _ = pol.evaluate_actions(x, None, None, action)
    np.testing.assert_allclose(logp.cpu().numpy(), lp2.detach().cpu().numpy(), rtol=TOL, atol=2e-5)
    np.testing.assert_allclose(value.cpu().numpy(), v2.detach().cpu().numpy(), rtol=1e-6, atol=1e-6)
    _, mode, _, _ = pol.act(x, None, None, deterministic=True)
    if not cat:
        torch.manual_seed(11)
        eps = torch.randn(B, A, device='cuda')
        std = pol.dist.logstd._bias.detach().view(1, A).exp()
        np.testing.assert_allclose(action.cpu().numpy(), (mode + std * eps).cpu().numpy(), rtol=1e-6, atol=1e-6)
        return
    assert int(action.min()) >= 0 and int(action.max()) < A
    # frequencies over many draws of row 0 against softmax of its logits (4 sigma of a binomial proportion)
    from a2c_ppo_acktr import engine
    with torch.no_grad():
        logits = engine.forward(pol, x[:1].repeat(4096, *([1] * (x.dim() - 1)))).dist_in()
    probs = torch.softmax(logits[0].double(), -1).cpu().numpy()
    torch.manual_seed(12)
    counts = np.zeros(A)
    n = 0
    for _ in range(4):
        _, a, _, _ = pol.act(x[:1].repeat(4096, *([1] * (x.dim() - 1))), None, None)
        counts += np.bincount(a.view(-1).cpu().numpy(), minlength=A)
        n += 4096
    freq = counts / n
    assert np.all(np.abs(freq - probs) <= 4.0 * np.sqrt(probs * (1 - probs) / n) + 1e-3), (freq, probs)


@pytest.mark.parametrize('ci', range(len(ACT_CASES)))
def test_act_sampling_reference_stream(ci):
    """Policy.act(deterministic=False) in its default mode consumes the CPU default generator like the reference's CPU
    path (Categorical: torch.multinomial's one-sample exponential race; DiagGaussian: torch.normal), so the same
    torch.manual_seed gives the REFERENCE'S actions: two consecutive calls and a 256-row batch against goldens written
    by the reference's own Policy.act (tests/golden/make_golden_act.py).  Categorical actions exact, Gaussian actions
    and every log-prob to 1e-4."""
    tag, kind, od, cat, A, B = ACT_CASES[ci]
    G = golden('act')
    pol, P, obs, ro = make_policy(ci)
    assert getattr(pol, 'sample_stream', 'reference') == 'reference'
    x = t(obs).cuda()
    calls = [(x, 'action0', 'logp0'), (x, 'action1', 'logp1')]
    if kind == 'mlp':
        wide = synth.make_rollouts(1, 256, (od,), 'discrete' if cat else 'box', A, salt=1480 + ci)['obs'][0]
        calls.append((t(wide).cuda(), 'wide_action', 'wide_logp'))
    torch.manual_seed(77 + ci)
    for xin, ka, kl in calls:
        _, action, logp, _ = pol.act(xin, None, None)
        ga, gl = G[tag + '/sample/' + ka], G[tag + '/sample/' + kl]
        if cat:
            assert action.dtype == torch.int64 and np.array_equal(action.cpu().numpy(), ga), (ka, action.view(-1), ga.ravel())
        else:
            np.testing.assert_allclose(action.cpu().numpy(), ga, rtol=TOL, atol=2e-5)
        np.testing.assert_allclose(logp.cpu().numpy(), gl, rtol=TOL, atol=2e-5)


@pytest.mark.parametrize('ci', [2, 3])
def test_pairing_evaluate_actions_and_fused_loss_vs_reference(ci):
    """CNNBase + Box and MLPBase + Discrete: evaluate_actions through autograd and the fused loss path PPO.update
    runs, against the reference's values / log-probs / entropy / losses / gradients."""
    from a2c_ppo_acktr import engine
    tag, kind, od, cat, A, B = ACT_CASES[ci]
    G = golden('act')
    pol, P, obs, ro = make_policy(ci)
    x = t(obs).cuda()
    act = t(ro['actions'][0]).cuda()
    vp, ret, adv = (t(a).cuda() for a in (ro['value_preds'][0], synth.normalish((B, 1), 1450 + ci),
                                          synth.normalish((B, 1), 1460 + ci)))
    olp = t(G[tag + '/old_logp']).cuda()
    values, logp, ent, _ = pol.evaluate_actions(x, None, None, act)
    np.testing.assert_allclose(values.detach().cpu().numpy(), G[tag + '/values'], rtol=TOL, atol=2e-5)
    np.testing.assert_allclose(logp.detach().cpu().numpy(), G[tag + '/logp'], rtol=TOL, atol=2e-5)
    assert abs(ent.item() - float(G[tag + '/entropy'])) <= TOL * abs(float(G[tag + '/entropy']))
    ratio = torch.exp(logp - olp)
    action_loss = -torch.min(ratio * adv, torch.clamp(ratio, 0.8, 1.2) * adv).mean()
    vpc = vp + (values - vp).clamp(-0.2, 0.2)
    value_loss = 0.5 * torch.max((values - ret).pow(2), (vpc - ret).pow(2)).mean()
    (value_loss * 0.5 + action_loss - ent * 0.01).backward()
    for n, p in pol.named_parameters():
        check_summary(G, tag + '/grad/' + n, p.grad.cpu().numpy(), TOL, tag + ' autograd')
    # fused loss path
    loss_acc = torch.zeros(3, device='cuda')
    gflat = torch.full_like(pol.flat_params, 5.0)
    with torch.no_grad():
        acts = engine.forward(pol, x)
        engine.policy_loss(pol, acts, engine.ALGO_PPO, act.view(-1).contiguous() if cat else act.contiguous(), loss_acc,
                           1.0 / B, old_logp=olp.view(-1), old_values=vp.view(-1), returns=ret.view(-1), adv=adv.view(-1),
                           clip=0.2, value_coef=0.5, entropy_coef=0.01, gflat=gflat)
        engine.backward(pol, acts, gflat)
    vl, al, en = loss_acc.tolist()
    # (the action loss is a mean of +-O(0.3) summands that cancel to ~1e-3 here: 1e-4 relative to the summand scale)
    for a_, b_ in ((vl, G[tag + '/value_loss']), (al, G[tag + '/action_loss']), (en, G[tag + '/entropy'])):
        assert abs(a_ - float(b_)) <= TOL * max(abs(float(b_)), 1e-2)
    for n, p in pol.named_parameters():
        check_summary(G, tag + '/grad/' + n, pol.grad_view(p, gflat).cpu().numpy(), TOL, tag + ' fused')


@pytest.mark.parametrize('ci', [2, 3])
@pytest.mark.parametrize('which', ['ppo', 'a2c', 'acktr'])
def test_pairing_updates_vs_oracle(ci, which):
    """PPO.update, A2C update and ACKTR update for the two extra pairings against the CPU oracle (oracle/algo_ref.py,
    oracle/kfac_ref.py -- pinned for these heads by test_oracle_golden.py) on the same rollouts, seeds and noise."""
    from a2c_ppo_acktr import algo
    from a2c_ppo_acktr.storage import RolloutStorage
    from oracle import algo_ref, kfac_ref, returns_ref
    tag, kind, od, cat, A, B = ACT_CASES[ci]
    T, N = (4, 4) if kind == 'cnn' else (8, 8)
    pol, Pn, _, _ = make_policy(ci)
    P = params_torch(Pn)
    obs_shape = (4, 84, 84) if kind == 'cnn' else (od,)
    ro = synth.make_rollouts(T, N, obs_shape, 'discrete' if cat else 'box', A, salt=1500 + ci)
    rs = RolloutStorage(T, N, obs_shape, Discrete(A) if cat else Box(A), 1)
    for k in ('obs', 'rewards', 'value_preds', 'returns', 'action_log_probs', 'actions', 'masks', 'bad_masks'):
        getattr(rs, k).copy_(t(ro[k]))
    rs.to('cuda')
    gae = which == 'ppo'
    rs.compute_returns(t(ro['next_value']).cuda(), gae, 0.99, 0.95, gae)
    returns_ref.compute_returns(ro['rewards'], ro['value_preds'], ro['returns'], ro['masks'], ro['bad_masks'],
                                ro['next_value'], gae, 0.99, 0.95, gae)
    rot = {k: t(v) for k, v in ro.items()}
    if which == 'ppo':
        hp = dict(clip_param=0.2, ppo_epoch=2, num_mini_batch=2, value_loss_coef=0.5, entropy_coef=0.01, lr=3e-4, eps=1e-5,
                  max_grad_norm=0.5)
        agent, opt = algo.PPO(pol, **hp), algo_ref.adam_init(P)
        step = lambda: algo_ref.ppo_update(P, opt, rot, **hp)
    elif which == 'a2c':
        hp = dict(value_loss_coef=0.5, entropy_coef=0.01, lr=7e-4, eps=1e-5, alpha=0.99, max_grad_norm=0.5)
        agent, opt = algo.A2C_ACKTR(pol, acktr=False, **hp), algo_ref.rmsprop_init(P)
        step = lambda: algo_ref.a2c_update(P, opt, rot, **hp)
    else:
        agent, st = algo.A2C_ACKTR(pol, value_loss_coef=0.5, entropy_coef=0.01, acktr=True), kfac_ref.new_state()
        step = lambda: kfac_ref.acktr_update(P, st, rot, 0.5, 0.01)[0]
    names = {n: n.replace('.module.weight', '.weight').replace('.add_bias._bias', '.bias') for n, _ in pol.named_parameters()}
    for it in range(2):
        torch.manual_seed(321 + it)
        got = agent.update(rs)
        torch.manual_seed(321 + it)
        want = step()
        # (ACKTR's second update sees parameters moved by a KFAC step whose damped inverse amplifies fp32 noise, DESIGN.md
        # section 2: its losses are held to 5e-3, everything else to the north_star 1e-4)
        ltol = 5e-3 if (which == 'acktr' and it > 0) else TOL
        np.testing.assert_allclose(np.array(got), np.array(want), rtol=ltol, atol=1e-6)
    # parameters after two updates.  RMSprop right after initialisation and KFAC's damped inverse amplify fp32
    # summation-order noise, so those two get the 1e-3 bar of the KFAC update
    ptol = TOL if which == 'ppo' else 1e-3
    num = den = 0.0
    for n, p in pol.named_parameters():
        a_, b_ = p.detach().cpu().double().reshape(-1), P[names[n]].detach().double().reshape(-1)
        num += float(((a_ - b_) ** 2).sum())
        den += float((b_ ** 2).sum())
    assert (num / den) ** 0.5 <= ptol, (which, tag, (num / den) ** 0.5)
