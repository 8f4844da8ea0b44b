"""Golden vectors for the rollout-side API and for the base x head combinations (build container only).

    python tests/golden/make_golden_act.py

Reference calls: Policy.act(deterministic=True) and Policy.get_value (model.py:54-70) for the four pairings of
{CNNBase, MLPBase} x {Discrete, Box} the reference's Policy.__init__ allows (model.py:20-38), plus evaluate_actions and
the PPO-minibatch gradients (the loss block of ppo.py:57-81 driven through the reference's own modules) for the two
pairings that the other goldens do not cover (CNNBase + Box, MLPBase + Discrete).  Inputs come from synth.py; the
fixture is tests/golden/act.npz.
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
from a2c_ppo_acktr.model import Policy                      # noqa: E402  (reference)
from oracle import policy_ref                               # noqa: E402  (shape tables only)
from make_golden import ppo_loss_ref, summarize, t          # noqa: E402

torch.set_num_threads(1)

# tag, base, obs dim, categorical head?, A, B   (mirrored by tests/helpers.py ACT_CASES)
ACT_CASES = [('cnn_cat', 'cnn', None, True, 6, 9), ('mlp_gauss', 'mlp', 11, False, 3, 16),
             ('cnn_gauss', 'cnn', None, False, 4, 5), ('mlp_cat', 'mlp', 17, True, 5, 12)]


def act_shapes(kind, od, cat, A):
    base = policy_ref.cnn_param_shapes(4, 512, A) if kind == 'cnn' else policy_ref.mlp_param_shapes(od, 64, A)
    return policy_ref.with_head(base, 512 if kind == 'cnn' else 64, A, cat)


def main():
    out = {}
    for ci, (tag, kind, od, cat, A, B) in enumerate(ACT_CASES):
        obs_shape = (4, 84, 84) if kind == 'cnn' else (od,)
        pol = Policy(obs_shape, ref_harness.Discrete(A) if cat else ref_harness.Box(A))
        shapes = act_shapes(kind, od, cat, A)
        P = synth.make_params(shapes, 140 + ci)
        pol.load_state_dict(OrderedDict((k, t(v)) for k, v in P.items()), strict=True)
        assert [n for n, _ in pol.named_parameters()] == list(shapes.keys()), 'parameter order differs'
        ro = synth.make_rollouts(1, B, obs_shape, 'discrete' if cat else 'box', A, salt=1400 + ci)
        obs = t(ro['obs'][0])
        with torch.no_grad():
            value, action, logp, _ = pol.act(obs, None, None, deterministic=True)
            gv = pol.get_value(obs, None, None)
        out[tag + '/act/value'] = value.numpy().copy()
        out[tag + '/act/action'] = action.numpy().copy()
        out[tag + '/act/logp'] = logp.numpy().copy()
        out[tag + '/get_value'] = gv.numpy().copy()
        # sampled actions (model.py:58-63 with deterministic=False): two consecutive calls behind one torch.manual_seed,
        # drawn by the reference's own dist.sample() on the CPU default generator -- the stream a drop-in has to consume
        # in the same order -- and a wider batch (256 rows) for the heads' statistics
        torch.manual_seed(77 + ci)
        with torch.no_grad():
            for k in range(2):
                _, action, logp, _ = pol.act(obs, None, None)
                out[tag + '/sample/action%d' % k] = action.numpy().copy()
                out[tag + '/sample/logp%d' % k] = logp.numpy().copy()
            if kind == 'mlp':
                wide = t(synth.make_rollouts(1, 256, obs_shape, 'discrete' if cat else 'box', A, salt=1480 + ci)['obs'][0])
                _, action, logp, _ = pol.act(wide, None, None)
                out[tag + '/sample/wide_action'] = action.numpy().copy()
                out[tag + '/sample/wide_logp'] = logp.numpy().copy()
        if tag in ('cnn_gauss', 'mlp_cat'):
            act = t(ro['actions'][0])
            vp = t(ro['value_preds'][0]); ret = t(synth.normalish((B, 1), 1450 + ci)); adv = t(synth.normalish((B, 1), 1460 + ci))
            with torch.no_grad():
                _, lp0, _, _ = pol.evaluate_actions(obs, None, None, act)
            olp = lp0 + 0.3 * t(synth.normalish((B, 1), 1470 + ci))
            values, lp, ent, vl, al = ppo_loss_ref(pol, obs, act, vp, ret, olp, adv, 0.2, 0.5, 0.01)
            out[tag + '/old_logp'] = olp.numpy().copy()
            out[tag + '/values'] = values.detach().numpy().copy()
            out[tag + '/logp'] = lp.detach().numpy().copy()
            out[tag + '/entropy'] = np.float64(ent.item())
            out[tag + '/value_loss'] = np.float64(vl.item())
            out[tag + '/action_loss'] = np.float64(al.item())
            for n, p in pol.named_parameters():
                summarize(tag + '/grad/' + n, p.grad.numpy(), out)
    np.savez_compressed(os.path.join(HERE, 'act.npz'), **out)
    print('wrote act.npz with', len(out), 'arrays')


if __name__ == '__main__':
    main()
