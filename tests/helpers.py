"""Shared helpers for parity tests (golden loading, summaries, tolerances)."""
import os
from collections import OrderedDict

import numpy as np
import torch

import synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
SAMPLE_STRIDE = 101


def golden(name):
    return np.load(os.path.join(GOLD, name + '.npz'), allow_pickle=False)


def t(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def params_torch(P):
    return OrderedDict((k, t(v).clone()) for k, v in P.items())


def rel_err(a, b):
    a = np.asarray(a, np.float64).reshape(-1)
    b = np.asarray(b, np.float64).reshape(-1)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def check_summary(G, prefix, arr, tol, what=''):
    """Compare a tensor with the golden's (norm, full|sample) summary; relative L2 metric."""
    a = np.asarray(arr, np.float32).reshape(-1)
    gn = float(G[prefix + '#norm'])
    n = float(np.sqrt((a.astype(np.float64) ** 2).sum()))
    assert abs(n - gn) <= tol * max(gn, 1e-12) + 1e-12, '%s %s norm %g vs %g' % (what, prefix, n, gn)
    if prefix + '#full' in G.files:
        ref = G[prefix + '#full']
        got = a
    else:
        ref = G[prefix + '#sample']
        got = a[::SAMPLE_STRIDE]
    scale = max(gn / np.sqrt(a.size), 1e-30) * np.sqrt(ref.size)   # expected norm of the compared subset
    err = float(np.linalg.norm(got.astype(np.float64) - ref.astype(np.float64)))
    assert err <= tol * max(scale, float(np.linalg.norm(ref))) + 1e-12, \
        '%s %s rel err %g' % (what, prefix, err / max(scale, 1e-30))


def cnn_gate_aware_grads(P, gates, x, old_logp, clip, c_e):
    """fp64 PPO-minibatch gradients of the NatureCNN policy with the ReLU gates IMPOSED (`gates`: four boolean
    arrays, the HIP path's own y1>0, y2>0, y3>0, feat>0), plus the fp64 pre-activations at the positions where an
    imposed gate disagrees with the fp64 sign.

    Why: a pre-activation whose fp64 magnitude lies below the fp32 accumulation noise (|pre| ~ 1e-7) has no
    determined sign in fp32 -- the reference's own conv and ours may round it to different sides, and one
    flipped gate moves the conv1 weight gradient by ~5e-4 relative.  Parity of the arithmetic is therefore
    checked with the gates held fixed, and the disagreeing gates are required to be such undetermined ones."""
    import torch.nn.functional as F
    dd = lambda v: torch.tensor(np.asarray(v), dtype=torch.float64)
    W = OrderedDict((k, dd(v).requires_grad_(True)) for k, v in P.items())
    g = [dd(np.asarray(m, np.float64)) for m in gates]
    a = dd(x['obs']) / 255.0
    flips = []
    for li, (n, s) in enumerate((('base.main.0', 4), ('base.main.2', 2), ('base.main.4', 1))):
        pre = F.conv2d(a, W[n + '.weight'], W[n + '.bias'], stride=s)
        flips.append(pre.detach()[(pre.detach() > 0) != (g[li] > 0)].abs())
        a = pre * g[li]
    pre = a.reshape(a.shape[0], -1) @ W['base.main.7.weight'].t() + W['base.main.7.bias']
    flips.append(pre.detach()[(pre.detach() > 0) != (g[3] > 0)].abs())
    feat = pre * g[3]
    if 'base.gru.weight_ih_l0' in W:                 # recurrent base: the GRU sits between the features and the heads
        from oracle import policy_ref
        feat, _ = policy_ref.forward_gru(W, feat, dd(x['hxs']), dd(x['masks']))
    values = feat @ W['base.critic_linear.weight'].t() + W['base.critic_linear.bias']
    logits = feat @ W['dist.linear.weight'].t() + W['dist.linear.bias']
    logsm = torch.log_softmax(logits, dim=-1)
    act = torch.tensor(np.asarray(x['act']), dtype=torch.int64).view(-1, 1)
    logp = logsm.gather(1, act)
    ent = -(logsm.exp() * logsm).sum(-1).mean()
    adv, vp, ret = dd(x['adv']), dd(x['vp']), dd(x['ret'])
    ratio = torch.exp(logp - dd(old_logp).view(-1, 1))
    action_loss = -torch.min(ratio * adv, torch.clamp(ratio, 1.0 - clip, 1.0 + clip) * adv).mean()
    vpc = vp + (values - vp).clamp(-clip, clip)
    value_loss = 0.5 * torch.max((values - ret).pow(2), (vpc - ret).pow(2)).mean()
    (value_loss * 0.5 + action_loss - ent * c_e).backward()
    return OrderedDict((k, v.grad.numpy()) for k, v in W.items()), torch.cat(flips).numpy()


def check_cnn_grads(G, tag, P, gates, x, old_logp, clip, c_e, grads, tol):
    """Gradient parity for a CNN case: the reference golden at `tol`; if that trips, the gate-aware fp64 check
    (see cnn_gate_aware_grads) -- every disagreeing gate must be an undetermined one (|pre| < 1e-5) and there
    may be at most a handful of them.  With no disagreeing gate at all the fallback is a plain fp64 anchor: the HIP
    result must sit within `tol` of the fp64 gradient (seen on the recurrent NatureCNN case at 15 samples, where
    the reference's own single-thread fp32 conv1 gradient is ~9e-4 from fp64 and ours ~1e-5)."""
    try:
        for n, g_ in grads.items():
            check_summary(G, tag + '/grad/' + n, g_, tol, tag)
        return 'golden'
    except AssertionError:
        pass
    ref, flips = cnn_gate_aware_grads(P, gates, x, old_logp, clip, c_e)
    assert flips.size <= 8 and (flips.size == 0 or float(flips.max()) < 1e-5), \
        '%s: %d gates disagree with fp64, largest |pre| %g' % (tag, flips.size, float(flips.max()) if flips.size else 0)
    for n, g_ in grads.items():
        e = rel_err(g_, ref[n])
        assert e <= tol, '%s %s gate-aware rel err %g' % (tag, n, e)
    return 'gate-aware (%d undetermined gates)' % flips.size


def eval_case_inputs(tag, kind, od, A, B, ci):
    """Inputs of golden group 'eval' (mirrors tests/golden/make_golden.py:g_eval)."""
    obs_shape = (4, 84, 84) if kind == 'cnn' else (od,)
    ro = synth.make_rollouts(1, B, obs_shape, 'discrete' if kind == 'cnn' else 'box', A, salt=900 + ci)
    return dict(obs=ro['obs'][0], act=ro['actions'][0], vp=ro['value_preds'][0],
                ret=synth.normalish((B, 1), 950 + ci), adv=synth.normalish((B, 1), 960 + ci))


EVAL_CASES = [('mlp11', 'mlp', 11, 3, 64, 0.2, 0.0), ('mlp17', 'mlp', 17, 6, 64, 0.2, 0.01),
              ('cnn16', 'cnn', None, 6, 16, 0.1, 0.01), ('cnn80', 'cnn', None, 6, 80, 0.1, 0.01)]

PPO_CASES = [('mlp', 'mlp', 11, 3, 32, 4, dict(clip_param=0.2, ppo_epoch=3, num_mini_batch=4, value_loss_coef=0.5,
                                               entropy_coef=0.0, lr=3e-4, eps=1e-5, max_grad_norm=0.5)),
             ('cnn', 'cnn', None, 6, 8, 4, dict(clip_param=0.1, ppo_epoch=2, num_mini_batch=2, value_loss_coef=0.5,
                                                entropy_coef=0.01, lr=2.5e-4, eps=1e-5, max_grad_norm=0.5))]

A2C_HP = dict(value_loss_coef=0.5, entropy_coef=0.01, lr=7e-4, eps=1e-5, alpha=0.99, max_grad_norm=0.5)
A2C_CASES = [('cnn', 'cnn', None, 6, 5, 16), ('mlp', 'mlp', 17, 6, 5, 16)]


def shapes_for(kind, od, A):
    from oracle import policy_ref
    return policy_ref.cnn_param_shapes(4, 512, A) if kind == 'cnn' else policy_ref.mlp_param_shapes(od, 64, A)


def golden_subset(G, prefix, arr):
    """(golden values, matching entries of arr) for a summarised tensor."""
    a = np.asarray(arr).reshape(-1)
    if prefix + '#full' in G.files:
        return G[prefix + '#full'].astype(np.float64), a.astype(np.float64)
    return G[prefix + '#sample'].astype(np.float64), a[::SAMPLE_STRIDE].astype(np.float64)


KFAC_CASES = [('cnn', 'cnn', None, 6, 20, 32), ('mlp', 'mlp', 17, 6, 5, 16),
              ('cnn_fast', 'cnn', None, 6, 20, 32)]          # tag *_fast: KFACOptimizer.fast_cnn = True (kfac.py:33-36,53-55)


def kfac_golden(tag):
    return golden('kfac_fast' if tag.endswith('_fast') else 'kfac')


def unsplit_name(n):
    """Parameter name after KFAC's split_bias (kfac.py:74-84) -> original state_dict name."""
    return n.replace('.module.weight', '.weight').replace('.add_bias._bias', '.bias')


def check_update_vector(G, prefix_fmt, names, get, tol=1e-3, what=''):
    """KFAC-preconditioned update parity (north_star: 1e-3 relative).  The update is ONE vector over all
    parameters; per-tensor errors are measured against max(tensor norm, 10% of the whole-update norm) so
    that a tiny, ill-conditioned block (e.g. the critic head when the feature factor is rank deficient at
    small B) is held to the vector-level bar rather than to its own near-zero norm."""
    whole = np.sqrt(sum(float(G[prefix_fmt % n + '#norm']) ** 2 for n in names))
    tot = 0.0
    cnt = 0
    for n in names:
        gold, mine = golden_subset(G, prefix_fmt % n, get(n))
        gn = float(G[prefix_fmt % n + '#norm'])
        full = np.asarray(get(n)).size
        frac = gold.size / full                      # sampled fraction of this tensor
        err = np.linalg.norm(mine - gold)
        ref_scale = max(gn, 0.1 * whole) * np.sqrt(frac)
        assert err <= tol * ref_scale + 1e-12, '%s %s: err %g vs scale %g' % (what, n, err, ref_scale)
        tot += err ** 2 / frac
    assert np.sqrt(tot) <= tol * whole, '%s whole-update rel err %g' % (what, np.sqrt(tot) / whole)


def update_vector_errors(G, prefix_fmt, names, get_mine, get_truth):
    """Whole-update-vector relative errors (reference golden vs truth, mine vs truth, mine vs golden), estimated
    on the golden's stored subset of every tensor (full tensors or strided samples, rescaled by the sampled
    fraction)."""
    e_ref = e_mine = e_mg = nt = 0.0
    for n in names:
        gold, mine = golden_subset(G, prefix_fmt % n, get_mine(n))
        _, truth = golden_subset(G, prefix_fmt % n, get_truth(n))
        frac = gold.size / float(np.asarray(get_mine(n)).size)
        e_ref += np.sum((gold - truth) ** 2) / frac
        e_mine += np.sum((mine - truth) ** 2) / frac
        e_mg += np.sum((mine - gold) ** 2) / frac
        nt += np.sum(truth ** 2) / frac
    nt = max(nt, 1e-300)
    return np.sqrt(e_ref / nt), np.sqrt(e_mine / nt), np.sqrt(e_mg / nt)


def reference_self_distance(prefix_fmt, names, other='kfac_t8', base='kfac', other_prefix_fmt=None):
    """|| ref(1 thread) - ref(8 threads) || / || ref(1 thread) || over the whole vector made of the tensors `names`
    (estimated on the goldens' stored subsets like update_vector_errors): how far the REFERENCE moves against itself
    when only the intra-op thread count of its CPU run changes (tests/golden/make_golden_kfac.py --threads 8 against
    the committed 1-thread goldens, main.py:38)."""
    G1, G8 = golden(base), golden(other)
    num = den = 0.0
    for n in names:
        key = prefix_fmt % n
        suf = '#full' if key + '#full' in G1.files else '#sample'
        a, b = G1[key + suf].astype(np.float64), G8[(other_prefix_fmt or prefix_fmt) % n + suf].astype(np.float64)
        frac = a.size / (a.size if suf == '#full' else a.size * float(SAMPLE_STRIDE))
        num += np.sum((a - b) ** 2) / frac
        den += np.sum(a ** 2) / frac
    return float(np.sqrt(num / max(den, 1e-300)))


# ------------------------------------------------------------------------------------------ GAIL
GAIL_CASES = [('hc', 17, 6, 100, 128, 8, 64), ('small', 11, 3, 64, 32, 5, 16)]     # tag, obs, act, hidden, batch, T, N


def gail_inputs(ci):
    """Discriminator parameters, a rollout and two fixed expert batches per update (mirrors make_golden_gail.py)."""
    from oracle import gail_ref
    tag, od, A, hid, B, T, N = GAIL_CASES[ci]
    P = synth.make_params(gail_ref.disc_param_shapes(od + A, hid), 300 + ci)
    ro = synth.make_rollouts(T, N, (od,), 'box', A, salt=700 + ci)
    expert = [(synth.normalish((B, od), 710 + 10 * ci + k), synth.normalish((B, A), 715 + 10 * ci + k)) for k in range(2)]
    return P, ro, expert


def gail_expert_file_tensors(ci):
    """A synthetic expert file in the gail_experts format (convert_to_pytorch.py:28-48): 6 trajectories x 60 steps."""
    tag, od, A, hid, B, T, N = GAIL_CASES[ci]
    nt, L = 6, 60
    return OrderedDict([('states', t(synth.normalish((nt, L, od), 730 + ci))), ('actions', t(synth.normalish((nt, L, A), 731 + ci))),
                        ('rewards', t(synth.normalish((nt, L), 732 + ci))),
                        ('lengths', torch.tensor([60, 60, 40, 60, 20, 60], dtype=torch.int64))])


# ------------------------------------------------------------------------------------------ recurrent path
# mirrors tests/golden/make_golden_rec.py
REC_EVAL_CASES = [('mlp', 'mlp', 11, 3, 6, 4), ('cnn', 'cnn', None, 6, 5, 3)]
REC_PPO_CASES = [('mlp', 'mlp', 11, 3, 8, 4, dict(clip_param=0.2, ppo_epoch=2, num_mini_batch=2, value_loss_coef=0.5,
                                                   entropy_coef=0.0, lr=3e-4, eps=1e-5, max_grad_norm=0.5)),
                 ('cnn', 'cnn', None, 6, 6, 4, dict(clip_param=0.1, ppo_epoch=2, num_mini_batch=2, value_loss_coef=0.5,
                                                    entropy_coef=0.01, lr=2.5e-4, eps=1e-5, max_grad_norm=0.5))]
REC_A2C_CASES = [('cnn', 'cnn', None, 6, 5, 4), ('mlp', 'mlp', 17, 6, 5, 8)]


def rec_shapes_for(kind, od, A):
    from oracle import policy_ref
    if kind == 'cnn':
        return policy_ref.cnn_param_shapes(4, 512, A, recurrent=True)
    return policy_ref.mlp_param_shapes(od, 64, A, recurrent=True)


def rec_rollouts(kind, od, A, T, N, salt):
    obs_shape = (4, 84, 84) if kind == 'cnn' else (od,)
    return synth.make_rollouts(T, N, obs_shape, 'discrete' if kind == 'cnn' else 'box', A, salt=salt, p_mask=0.8,
                               hidden=512 if kind == 'cnn' else 64)


def rec_eval_inputs(ci):
    """numpy inputs of golden group rec/eval_* (obs, act, hxs, masks, vp, ret, adv) plus the raw rollout."""
    tag, kind, od, A, T, N = REC_EVAL_CASES[ci]
    ro = rec_rollouts(kind, od, A, T, N, 1900 + ci)
    B = T * N
    return dict(obs=ro['obs'][:-1].reshape((B,) + ro['obs'].shape[2:]), act=ro['actions'].reshape(B, -1),
                hxs=ro['recurrent_hidden_states'][0], masks=ro['masks'][:-1].reshape(B, 1),
                vp=ro['value_preds'][:-1].reshape(B, 1), ret=synth.normalish((B, 1), 1950 + ci),
                adv=synth.normalish((B, 1), 1960 + ci)), ro


# ------------------------------------------------------------------------------------------ rollout side + base x head pairings
# mirrors tests/golden/make_golden_act.py: tag, base, obs dim, categorical head?, A, B
ACT_CASES = [('cnn_cat', 'cnn', None, True, 6, 9), ('mlp_gauss', 'mlp', 11, False, 3, 16),
             ('cnn_gauss', 'cnn', None, False, 4, 5), ('mlp_cat', 'mlp', 17, True, 5, 12)]


def act_shapes(kind, od, cat, A):
    from oracle import policy_ref
    base = policy_ref.cnn_param_shapes(4, 512, A) if kind == 'cnn' else policy_ref.mlp_param_shapes(od, 64, A)
    return policy_ref.with_head(base, 512 if kind == 'cnn' else 64, A, cat)


def act_case_inputs(ci):
    """(params dict, obs, rollout dict) of ACT_CASES[ci], regenerated from synth."""
    tag, kind, od, cat, A, B = ACT_CASES[ci]
    obs_shape = (4, 84, 84) if kind == 'cnn' else (od,)
    P = synth.make_params(act_shapes(kind, od, cat, A), 140 + ci)
    ro = synth.make_rollouts(1, B, obs_shape, 'discrete' if cat else 'box', A, salt=1400 + ci)
    return P, ro['obs'][0], ro
