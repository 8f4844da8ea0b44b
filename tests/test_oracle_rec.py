"""Pin the oracle's recurrent (GRU) restatement against goldens generated from the reference itself
(tests/golden/make_golden_rec.py -> rec.npz).  CPU only."""
import numpy as np
import pytest
import torch

import synth
from helpers import (A2C_HP, REC_A2C_CASES, REC_EVAL_CASES, REC_PPO_CASES, check_summary, golden, params_torch,
                     rec_eval_inputs, rec_rollouts, rec_shapes_for, t)
from oracle import algo_ref, policy_ref, returns_ref

torch.set_num_threads(4)


@pytest.mark.parametrize('ci', range(len(REC_EVAL_CASES)))
def test_rec_eval_act_and_grads(ci):
    tag, kind, od, A, T, N = REC_EVAL_CASES[ci]
    G = golden('rec')
    k = 'eval_' + tag
    P = params_torch(synth.make_params(rec_shapes_for(kind, od, A), 140 + ci))
    x, ro = rec_eval_inputs(ci)
    clip, c_e = (0.1, 0.01) if kind == 'cnn' else (0.2, 0.0)
    with torch.no_grad():
        values, logp, ent, hout = policy_ref.evaluate_actions(P, t(x['obs']), t(x['act']), t(x['hxs']), t(x['masks']),
                                                              return_hxs=True)
    np.testing.assert_allclose(values.numpy(), G[k + '/values'], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(logp.numpy(), G[k + '/logp'], rtol=2e-5, atol=2e-6)
    assert abs(ent.item() - float(G[k + '/entropy'])) < 1e-5 * abs(float(G[k + '/entropy']))
    check_summary(G, k + '/hxs_out', hout.numpy(), 1e-5, tag)
    losses, grads = algo_ref.ppo_minibatch_grads(P, t(x['obs']), t(x['act']), t(x['vp']), t(x['ret']),
                                                 t(G[k + '/old_logp']), t(x['adv']), clip, 0.5, c_e,
                                                 rnn_hxs=t(x['hxs']), masks=t(x['masks']))
    assert abs(losses[0] - float(G[k + '/value_loss'])) < 1e-5 * abs(float(G[k + '/value_loss']))
    assert abs(losses[1] - float(G[k + '/action_loss'])) < 1e-5 * abs(float(G[k + '/action_loss'])) + 1e-7
    for n, g in grads.items():
        check_summary(G, k + '/grad/' + n, g.numpy(), 2e-5, tag)
    with torch.no_grad():
        v1, a1, lp1, h1 = policy_ref.act_deterministic(P, t(ro['obs'][1]), t(ro['recurrent_hidden_states'][1]),
                                                       t(ro['masks'][1]))
        v2 = policy_ref.evaluate_actions(P, t(ro['obs'][2]), a1, t(ro['recurrent_hidden_states'][2]), t(ro['masks'][2]))[0]
    np.testing.assert_allclose(v1.numpy(), G[k + '/act/value'], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(a1.numpy(), G[k + '/act/action'], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(lp1.numpy(), G[k + '/act/logp'], rtol=2e-5, atol=2e-6)
    check_summary(G, k + '/act/hxs', h1.numpy(), 1e-5, tag)
    np.testing.assert_allclose(v2.numpy(), G[k + '/get_value'], rtol=2e-5, atol=2e-6)


def test_rec_generator():
    G = golden('rec')
    T, N = 6, 6
    ro = rec_rollouts('mlp', 11, 3, T, N, 2100)
    returns_ref.compute_returns(ro['rewards'], ro['value_preds'], ro['returns'], ro['masks'], ro['bad_masks'],
                                ro['next_value'], True, 0.99, 0.95, True)
    ro = {k: t(v) for k, v in ro.items()}
    adv = algo_ref.normalize_advantages(ro)
    torch.manual_seed(2024)
    for e in range(2):
        batches = algo_ref.recurrent_env_batches(N, 3)
        assert len(batches) == 3
        for b, envs in enumerate(batches):
            got = algo_ref.gather_recurrent(ro, envs, adv)
            for name, v in zip(('obs', 'hxs', 'act', 'vp', 'ret', 'masks', 'olp', 'adv'), got):
                want = G['gen/e%d_b%d/%s' % (e, b, name)]
                if name == 'adv':
                    np.testing.assert_allclose(v.numpy(), want, rtol=1e-6, atol=1e-7)
                else:
                    assert np.array_equal(v.numpy(), want), name


def _ppo_rollout(tag, kind, od, A, T, N, ci, G):
    ro = rec_rollouts(kind, od, A, T, N, 2200 + ci)
    last = ro['value_preds'][-1].copy()
    ro['action_log_probs'] = G['ppo_' + tag + '/action_log_probs'].copy()
    ro['value_preds'] = G['ppo_' + tag + '/value_preds'].copy()
    ro['value_preds'][-1] = last
    returns_ref.compute_returns(ro['rewards'], ro['value_preds'], ro['returns'], ro['masks'], ro['bad_masks'],
                                ro['next_value'], True, 0.99, 0.95, True)
    return {k: t(v) for k, v in ro.items()}


@pytest.mark.parametrize('ci', range(len(REC_PPO_CASES)))
def test_rec_ppo_update(ci):
    tag, kind, od, A, T, N, hp = REC_PPO_CASES[ci]
    G = golden('rec')
    P = params_torch(synth.make_params(rec_shapes_for(kind, od, A), 160 + ci))
    ro = _ppo_rollout(tag, kind, od, A, T, N, ci, G)
    opt = algo_ref.adam_init(P)
    for it in range(2):
        torch.manual_seed(5321 + it)
        res = algo_ref.ppo_update(P, opt, ro, **hp)
        np.testing.assert_allclose(np.array(res), G['ppo_%s/update%d/losses' % (tag, it)], rtol=2e-5, atol=1e-7)
        for n, p in P.items():
            check_summary(G, 'ppo_%s/update%d/param/%s' % (tag, it, n), p.detach().numpy(), 1e-5, tag)


@pytest.mark.parametrize('ci', range(len(REC_A2C_CASES)))
def test_rec_a2c_update(ci):
    tag, kind, od, A, T, N = REC_A2C_CASES[ci]
    G = golden('rec')
    P = params_torch(synth.make_params(rec_shapes_for(kind, od, A), 180 + ci))
    ro = rec_rollouts(kind, od, A, T, N, 2300 + ci)
    returns_ref.compute_returns(ro['rewards'], ro['value_preds'], ro['returns'], ro['masks'], ro['bad_masks'],
                                ro['next_value'], False, 0.99, 0.95, False)
    ro = {k: t(v) for k, v in ro.items()}
    opt = algo_ref.rmsprop_init(P)
    for it in range(2):
        res = algo_ref.a2c_update(P, opt, ro, **A2C_HP)
        np.testing.assert_allclose(np.array(res), G['a2c_%s/update%d/losses' % (tag, it)], rtol=2e-5, atol=1e-7)
        for n, p in P.items():
            check_summary(G, 'a2c_%s/update%d/param/%s' % (tag, it, n), p.detach().numpy(), 1e-5, tag)
