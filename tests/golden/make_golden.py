"""Generate golden vectors from the REFERENCE itself (build container only).

    python tests/golden/make_golden.py [group ...]

Imports /root/reference behind the stubs in ref_harness.py, feeds it the
closed-form inputs of synth.py and stores the reference's outputs as small
.npz fixtures next to this file.  Large tensors (NatureCNN weights/grads) are
stored as per-tensor L2 norms + strided samples; inputs are never stored --
tests regenerate them from synth.py.

Groups: returns, sampler, eval, ppo, a2c, kfac
"""
import os
import sys
from collections import OrderedDict

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))   # repo root, for oracle.policy_ref shapes
import synth           # noqa: E402
import ref_harness     # noqa: E402

ref_harness.install()
from a2c_ppo_acktr.storage import RolloutStorage            # noqa: E402  (reference)
from a2c_ppo_acktr.model import Policy                      # noqa: E402  (reference)
from a2c_ppo_acktr import algo                              # noqa: E402  (reference)
from oracle import policy_ref                               # noqa: E402  (shape tables only)

torch.set_num_threads(1)        # main.py:38
SAMPLE_STRIDE = 101


def t(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def summarize(name, arr, out):
    """Full tensor if small, else norm + strided sample."""
    a = np.asarray(arr, dtype=np.float32).reshape(-1)
    out[name + '#norm'] = np.float64(np.sqrt((a.astype(np.float64) ** 2).sum()))
    if a.size <= 4096:
        out[name + '#full'] = a.copy()
    else:
        out[name + '#sample'] = a[::SAMPLE_STRIDE].copy()


def fill_storage(ro, obs_shape, action_space):
    T, N = ro['rewards'].shape[:2]
    rs = RolloutStorage(T, N, obs_shape, action_space, 1)
    for k in ('obs', 'rewards', 'value_preds', 'returns', 'action_log_probs', 'actions', 'masks', 'bad_masks'):
        getattr(rs, k).copy_(t(ro[k]))
    return rs


def make_policy(kind, salt, obs_dim=None, A=None):
    if kind == 'cnn':
        shapes = policy_ref.cnn_param_shapes(4, 512, A)
        pol = Policy((4, 84, 84), ref_harness.Discrete(A))
    else:
        shapes = policy_ref.mlp_param_shapes(obs_dim, 64, A)
        pol = Policy((obs_dim,), ref_harness.Box(A))
    P = synth.make_params(shapes, salt)
    missing = pol.load_state_dict(OrderedDict((k, t(v)) for k, v in P.items()), strict=True)
    assert [n for n, _ in pol.named_parameters()] == list(shapes.keys()), 'parameter order differs'
    return pol, P


# ---------------------------------------------------------------- returns
def g_returns():
    out = {}
    for (T, N) in [(5, 16), (128, 8), (2048, 1), (257, 33)]:
        ro = synth.make_rollouts(T, N, (3,), 'box', 2, salt=7 * T + N, p_mask=0.95, p_bad=0.96)
        for use_gae in (0, 1):
            for proper in (0, 1):
                rs = fill_storage(ro, (3,), ref_harness.Box(2))
                rs.compute_returns(t(ro['next_value']), bool(use_gae), 0.99, 0.95, bool(proper))
                key = 'T%d_N%d_gae%d_proper%d' % (T, N, use_gae, proper)
                out[key + '/returns'] = rs.returns.numpy().copy()
                out[key + '/value_preds_last'] = rs.value_preds[-1].numpy().copy()
    np.savez_compressed(os.path.join(HERE, 'returns.npz'), **out)


# ---------------------------------------------------------------- sampler + adv norm
def g_sampler():
    out = {}
    T, N = 128, 8
    ro = synth.make_rollouts(T, N, (11,), 'box', 3, salt=501)
    rs = fill_storage(ro, (11,), ref_harness.Box(3))
    rs.compute_returns(t(ro['next_value']), True, 0.99, 0.95, True)
    adv = rs.returns[:-1] - rs.value_preds[:-1]
    adv = (adv - adv.mean()) / (adv.std() + 1e-5)          # ppo.py:35-37
    out['adv'] = adv.numpy().copy()
    torch.manual_seed(1234)
    for e in range(2):
        for b, s in enumerate(rs.feed_forward_generator(adv, 4)):
            obs_b, rhs_b, act_b, vp_b, ret_b, m_b, olp_b, adv_b = s
            out['e%d_b%d/adv' % (e, b)] = adv_b.numpy().copy()
            out['e%d_b%d/obs_sum' % (e, b)] = obs_b.double().sum(1).numpy().copy()
            out['e%d_b%d/ret' % (e, b)] = ret_b.numpy().copy()
            out['e%d_b%d/act' % (e, b)] = act_b.numpy().copy()
    # GAIL-style call: advantages None, explicit mini_batch_size (gail.py:60-61)
    torch.manual_seed(99)
    for b, s in enumerate(rs.feed_forward_generator(None, mini_batch_size=100)):
        assert s[7] is None
        out['gail_b%d/olp' % b] = s[6].numpy().copy()
    out['gail_nb'] = np.int64(b + 1)
    np.savez_compressed(os.path.join(HERE, 'sampler.npz'), **out)


# ---------------------------------------------------------------- evaluate_actions + PPO minibatch grads
def ppo_loss_ref(pol, obs, act, vp, ret, olp, adv, clip, c_v, c_e):
    """The loss block of reference ppo.py:57-81 is inline in PPO.update; drive the same ops through the
    reference Policy so the gradient golden comes from the reference's own modules."""
    values, logp, ent, _ = pol.evaluate_actions(obs, None, None, act)
    ratio = torch.exp(logp - olp)
    surr1 = ratio * adv
    surr2 = torch.clamp(ratio, 1.0 - clip, 1.0 + clip) * adv
    action_loss = -torch.min(surr1, surr2).mean()
    vpc = vp + (values - vp).clamp(-clip, clip)
    value_loss = 0.5 * torch.max((values - ret).pow(2), (vpc - ret).pow(2)).mean()
    pol.zero_grad()
    (value_loss * c_v + action_loss - ent * c_e).backward()
    return values, logp, ent, value_loss, action_loss


def g_eval():
    out = {}
    cases = [('mlp11', 'mlp', 11, 3, 64, 0.2, 0.0), ('mlp17', 'mlp', 17, 6, 64, 0.2, 0.01),
             ('cnn16', 'cnn', None, 6, 16, 0.1, 0.01), ('cnn80', 'cnn', None, 6, 80, 0.1, 0.01)]
    for ci, (tag, kind, od, A, B, clip, c_e) in enumerate(cases):
        pol, P = make_policy(kind, 40 + ci, od, A)
        obs_shape = (4, 84, 84) if kind == 'cnn' else (od,)
        ro = synth.make_rollouts(1, B, obs_shape, 'discrete' if kind == 'cnn' else 'box', A, salt=900 + ci)
        obs = t(ro['obs'][0]); act = t(ro['actions'][0])
        vp = t(ro['value_preds'][0]); ret = t(synth.normalish((B, 1), 950 + ci))
        olp = t(ro['action_log_probs'][0]); adv = t(synth.normalish((B, 1), 960 + ci))
        if kind == 'mlp':  # keep ratios near 1: old log-probs close to the current ones
            with torch.no_grad():
                _, lp0, _, _ = pol.evaluate_actions(obs, None, None, act)
            olp = lp0 + 0.3 * t(synth.normalish((B, 1), 970 + ci))
        else:
            olp = t((np.log(1.0 / A) + 0.3 * synth.normalish((B, 1), 970 + ci)).astype(np.float32))
        values, logp, ent, vl, al = ppo_loss_ref(pol, obs, act, vp, ret, olp, adv, clip, 0.5, c_e)
        out[tag + '/old_logp'] = olp.numpy().copy()
        out[tag + '/values'] = values.detach().numpy().copy()
        out[tag + '/logp'] = logp.detach().numpy().copy()
        out[tag + '/entropy'] = np.float64(ent.item())
        out[tag + '/value_loss'] = np.float64(vl.item())
        out[tag + '/action_loss'] = np.float64(al.item())
        for n, p in pol.named_parameters():
            summarize(tag + '/grad/' + n, p.grad.numpy(), out)
    np.savez_compressed(os.path.join(HERE, 'eval.npz'), **out)


# ---------------------------------------------------------------- full PPO.update
def g_ppo():
    out = {}
    cases = [('mlp', 'mlp', 11, 3, 32, 4, dict(clip_param=0.2, ppo_epoch=3, num_mini_batch=4, value_loss_coef=0.5,
                                                entropy_coef=0.0, lr=3e-4, eps=1e-5, max_grad_norm=0.5)),
             ('cnn', 'cnn', None, 6, 8, 4, dict(clip_param=0.1, ppo_epoch=2, num_mini_batch=2, value_loss_coef=0.5,
                                                entropy_coef=0.01, lr=2.5e-4, eps=1e-5, max_grad_norm=0.5))]
    for ci, (tag, kind, od, A, T, N, hp) in enumerate(cases):
        pol, P = make_policy(kind, 60 + ci, od, A)
        obs_shape = (4, 84, 84) if kind == 'cnn' else (od,)
        space = ref_harness.Discrete(A) if kind == 'cnn' else ref_harness.Box(A)
        ro = synth.make_rollouts(T, N, obs_shape, 'discrete' if kind == 'cnn' else 'box', A, salt=700 + ci)
        rs = fill_storage(ro, obs_shape, space)
        with torch.no_grad():   # make stored log-probs / values consistent with the policy, as a real rollout would
            v, lp, _, _ = pol.evaluate_actions(rs.obs[:-1].view(T * N, *obs_shape), None, None,
                                               rs.actions.view(T * N, -1))
            rs.action_log_probs.copy_(lp.view(T, N, 1) + 0.05 * t(synth.normalish((T, N, 1), 777 + ci)))
            rs.value_preds[:-1].copy_(v.view(T, N, 1) + 0.1 * t(synth.normalish((T, N, 1), 778 + ci)))
        out[tag + '/action_log_probs'] = rs.action_log_probs.numpy().copy()
        out[tag + '/value_preds'] = rs.value_preds.numpy().copy()
        rs.compute_returns(t(ro['next_value']), True, 0.99, 0.95, True)
        agent = algo.PPO(pol, **hp)
        for it in range(2):
            torch.manual_seed(4321 + it)
            res = agent.update(rs)
            out[tag + '/update%d/losses' % it] = np.array(res, dtype=np.float64)
            for n, p in pol.named_parameters():
                summarize(tag + '/update%d/param/%s' % (it, n), p.detach().numpy(), out)
        for n, p in pol.named_parameters():
            st = agent.optimizer.state[p]
            summarize(tag + '/exp_avg/' + n, st['exp_avg'].numpy(), out)
            summarize(tag + '/exp_avg_sq/' + n, st['exp_avg_sq'].numpy(), out)
    np.savez_compressed(os.path.join(HERE, 'ppo.npz'), **out)


# ---------------------------------------------------------------- A2C update (RMSprop)
def g_a2c():
    out = {}
    hp = dict(value_loss_coef=0.5, entropy_coef=0.01, lr=7e-4, eps=1e-5, alpha=0.99, max_grad_norm=0.5)
    for ci, (tag, kind, od, A, T, N) in enumerate([('cnn', 'cnn', None, 6, 5, 16), ('mlp', 'mlp', 17, 6, 5, 16)]):
        pol, P = make_policy(kind, 80 + ci, od, A)
        obs_shape = (4, 84, 84) if kind == 'cnn' else (od,)
        space = ref_harness.Discrete(A) if kind == 'cnn' else ref_harness.Box(A)
        ro = synth.make_rollouts(T, N, obs_shape, 'discrete' if kind == 'cnn' else 'box', A, salt=800 + ci)
        rs = fill_storage(ro, obs_shape, space)
        rs.compute_returns(t(ro['next_value']), False, 0.99, 0.95, False)
        agent = algo.A2C_ACKTR(pol, acktr=False, **hp)
        for it in range(3):
            res = agent.update(rs)
            out[tag + '/update%d/losses' % it] = np.array(res, dtype=np.float64)
            for n, p in pol.named_parameters():
                summarize(tag + '/update%d/param/%s' % (it, n), p.detach().numpy(), out)
    np.savez_compressed(os.path.join(HERE, 'a2c.npz'), **out)


GROUPS = OrderedDict([('returns', g_returns), ('sampler', g_sampler), ('eval', g_eval), ('ppo', g_ppo),
                      ('a2c', g_a2c)])

if __name__ == '__main__':
    todo = sys.argv[1:] or list(GROUPS.keys())
    for g in todo:
        if g == 'kfac':
            import make_golden_kfac
            make_golden_kfac.main()
            continue
        print('golden group', g, flush=True)
        GROUPS[g]()
    print('done')
