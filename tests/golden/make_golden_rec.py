"""Golden vectors for the recurrent (GRU) path, generated from the REFERENCE (build container only).

    python tests/golden/make_golden_rec.py

Covers reference model.py:54-79,111-166 (act / evaluate_actions through `_forward_gru`), storage.py:145-202
(`recurrent_generator`), ppo.py:44-46 and a2c_acktr.py:38-43 with recurrent policies.  Inputs come from synth.py and
are never stored.
"""
import os
import sys
from collections import OrderedDict

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import synth           # noqa: E402
import ref_harness     # noqa: E402

ref_harness.install()
from a2c_ppo_acktr.storage import RolloutStorage            # noqa: E402  (reference)
from a2c_ppo_acktr.model import Policy                      # noqa: E402  (reference)
from a2c_ppo_acktr import algo                              # noqa: E402  (reference)
from oracle import policy_ref                               # noqa: E402  (shape tables only)
from make_golden import t, summarize                        # noqa: E402

torch.set_num_threads(1)

# (tag, kind, obs_dim, A, T, N): masks drawn with p_one = .8 so most trajectories contain resets
EVAL_CASES = [('mlp', 'mlp', 11, 3, 6, 4), ('cnn', 'cnn', None, 6, 5, 3)]
PPO_CASES = [('mlp', 'mlp', 11, 3, 8, 4, dict(clip_param=0.2, ppo_epoch=2, num_mini_batch=2, value_loss_coef=0.5,
                                               entropy_coef=0.0, lr=3e-4, eps=1e-5, max_grad_norm=0.5)),
             ('cnn', 'cnn', None, 6, 6, 4, dict(clip_param=0.1, ppo_epoch=2, num_mini_batch=2, value_loss_coef=0.5,
                                                entropy_coef=0.01, lr=2.5e-4, eps=1e-5, max_grad_norm=0.5))]
A2C_CASES = [('cnn', 'cnn', None, 6, 5, 4), ('mlp', 'mlp', 17, 6, 5, 8)]
A2C_HP = dict(value_loss_coef=0.5, entropy_coef=0.01, lr=7e-4, eps=1e-5, alpha=0.99, max_grad_norm=0.5)


def hidden_of(kind):
    return 512 if kind == 'cnn' else 64


def make_policy(kind, salt, obs_dim, A):
    if kind == 'cnn':
        shapes = policy_ref.cnn_param_shapes(4, 512, A, recurrent=True)
        pol = Policy((4, 84, 84), ref_harness.Discrete(A), base_kwargs={'recurrent': True})
    else:
        shapes = policy_ref.mlp_param_shapes(obs_dim, 64, A, recurrent=True)
        pol = Policy((obs_dim,), ref_harness.Box(A), base_kwargs={'recurrent': True})
    P = synth.make_params(shapes, salt)
    pol.load_state_dict(OrderedDict((k, t(v)) for k, v in P.items()), strict=True)
    assert [n for n, _ in pol.named_parameters()] == list(shapes.keys()), 'parameter order differs'
    return pol, P


def rollouts(kind, od, A, T, N, salt):
    obs_shape = (4, 84, 84) if kind == 'cnn' else (od,)
    ro = synth.make_rollouts(T, N, obs_shape, 'discrete' if kind == 'cnn' else 'box', A, salt=salt, p_mask=0.8,
                             hidden=hidden_of(kind))
    space = ref_harness.Discrete(A) if kind == 'cnn' else ref_harness.Box(A)
    rs = RolloutStorage(T, N, obs_shape, space, hidden_of(kind))
    for k in ('obs', 'rewards', 'value_preds', 'returns', 'action_log_probs', 'actions', 'masks', 'bad_masks',
              'recurrent_hidden_states'):
        getattr(rs, k).copy_(t(ro[k]))
    return ro, rs, obs_shape


def main():
    out = {}
    # ---- evaluate_actions (sequence form) + PPO-loss gradients, act (single-step form)
    for ci, (tag, kind, od, A, T, N) in enumerate(EVAL_CASES):
        pol, P = make_policy(kind, 140 + ci, od, A)
        ro, rs, obs_shape = rollouts(kind, od, A, T, N, 1900 + ci)
        B = T * N
        obs = rs.obs[:-1].view(B, *obs_shape)
        act = rs.actions.view(B, -1)
        hxs = rs.recurrent_hidden_states[0]
        masks = rs.masks[:-1].view(B, 1)
        vp = rs.value_preds[:-1].view(B, 1)
        ret = t(synth.normalish((B, 1), 1950 + ci))
        adv = t(synth.normalish((B, 1), 1960 + ci))
        with torch.no_grad():
            _, lp0, _, _ = pol.evaluate_actions(obs, hxs, masks, act)
        olp = lp0 + 0.3 * t(synth.normalish((B, 1), 1970 + ci))
        clip, c_e = (0.1, 0.01) if kind == 'cnn' else (0.2, 0.0)
        values, logp, ent, hout = pol.evaluate_actions(obs, hxs, masks, act)
        ratio = torch.exp(logp - olp)
        action_loss = -torch.min(ratio * adv, torch.clamp(ratio, 1.0 - clip, 1.0 + clip) * adv).mean()
        vpc = vp + (values - vp).clamp(-clip, clip)
        value_loss = 0.5 * torch.max((values - ret).pow(2), (vpc - ret).pow(2)).mean()
        pol.zero_grad()
        (value_loss * 0.5 + action_loss - ent * c_e).backward()
        k = 'eval_' + tag
        out[k + '/old_logp'] = olp.numpy().copy()
        out[k + '/values'] = values.detach().numpy().copy()
        out[k + '/logp'] = logp.detach().numpy().copy()
        out[k + '/entropy'] = np.float64(ent.item())
        out[k + '/value_loss'] = np.float64(value_loss.item())
        out[k + '/action_loss'] = np.float64(action_loss.item())
        summarize(k + '/hxs_out', hout.detach().numpy(), out)
        for n, p in pol.named_parameters():
            summarize(k + '/grad/' + n, p.grad.numpy(), out)
        with torch.no_grad():           # rollout-side single step (x.size(0) == hxs.size(0), model.py:112-115)
            v1, a1, lp1, h1 = pol.act(rs.obs[1], rs.recurrent_hidden_states[1], rs.masks[1], deterministic=True)
            v2 = pol.get_value(rs.obs[2], rs.recurrent_hidden_states[2], rs.masks[2])
        out[k + '/act/value'] = v1.numpy().copy()
        out[k + '/act/action'] = a1.numpy().copy()
        out[k + '/act/logp'] = lp1.numpy().copy()
        summarize(k + '/act/hxs', h1.numpy(), out)
        out[k + '/get_value'] = v2.numpy().copy()
    # ---- recurrent_generator
    T, N = 6, 6
    ro, rs, obs_shape = rollouts('mlp', 11, 3, T, N, 2100)
    rs.compute_returns(t(ro['next_value']), True, 0.99, 0.95, True)
    adv = rs.returns[:-1] - rs.value_preds[:-1]
    adv = (adv - adv.mean()) / (adv.std() + 1e-5)
    torch.manual_seed(2024)
    for e in range(2):
        for b, s in enumerate(rs.recurrent_generator(adv, 3)):
            for name, v in zip(('obs', 'hxs', 'act', 'vp', 'ret', 'masks', 'olp', 'adv'), s):
                out['gen/e%d_b%d/%s' % (e, b, name)] = v.numpy().copy()
    # ---- PPO.update with a recurrent policy
    for ci, (tag, kind, od, A, T, N, hp) in enumerate(PPO_CASES):
        pol, P = make_policy(kind, 160 + ci, od, A)
        ro, rs, obs_shape = rollouts(kind, od, A, T, N, 2200 + ci)
        with torch.no_grad():
            v, lp, _, _ = pol.evaluate_actions(rs.obs[:-1].view(T * N, *obs_shape), rs.recurrent_hidden_states[0],
                                               rs.masks[:-1].view(T * N, 1), rs.actions.view(T * N, -1))
            rs.action_log_probs.copy_(lp.view(T, N, 1) + 0.05 * t(synth.normalish((T, N, 1), 2277 + ci)))
            rs.value_preds[:-1].copy_(v.view(T, N, 1) + 0.1 * t(synth.normalish((T, N, 1), 2278 + ci)))
        k = 'ppo_' + tag
        out[k + '/action_log_probs'] = rs.action_log_probs.numpy().copy()
        out[k + '/value_preds'] = rs.value_preds.numpy().copy()
        rs.compute_returns(t(ro['next_value']), True, 0.99, 0.95, True)
        agent = algo.PPO(pol, **hp)
        for it in range(2):
            torch.manual_seed(5321 + it)
            res = agent.update(rs)
            out[k + '/update%d/losses' % it] = np.array(res, dtype=np.float64)
            for n, p in pol.named_parameters():
                summarize(k + '/update%d/param/%s' % (it, n), p.detach().numpy(), out)
    # ---- A2C update with a recurrent policy
    for ci, (tag, kind, od, A, T, N) in enumerate(A2C_CASES):
        pol, P = make_policy(kind, 180 + ci, od, A)
        ro, rs, obs_shape = rollouts(kind, od, A, T, N, 2300 + ci)
        rs.compute_returns(t(ro['next_value']), False, 0.99, 0.95, False)
        agent = algo.A2C_ACKTR(pol, acktr=False, **A2C_HP)
        k = 'a2c_' + tag
        for it in range(2):
            res = agent.update(rs)
            out[k + '/update%d/losses' % it] = np.array(res, dtype=np.float64)
            for n, p in pol.named_parameters():
                summarize(k + '/update%d/param/%s' % (it, n), p.detach().numpy(), out)
    np.savez_compressed(os.path.join(HERE, 'rec.npz'), **out)
    print('wrote rec.npz with %d arrays' % len(out))


if __name__ == '__main__':
    main()
