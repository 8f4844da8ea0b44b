"""Pins oracle/gail_ref.py against goldens produced by the reference's own gail.py (tests/golden/make_golden_gail.py).
`RunningMeanStd` is the one parity-unpinned call (sb3 is not in the container; the golden harness and the oracle share
the restatement) -- everything else (trunk, BCE losses, gradient penalty, Adam, dataset indexing) is the reference's."""
from collections import OrderedDict

import numpy as np
import pytest
import torch

from helpers import GAIL_CASES, gail_expert_file_tensors, gail_inputs, golden, t
from oracle import algo_ref, gail_ref


def policy_batches(ro, B):
    """storage.py:107-143 with advantages=None: (obs, actions) rows for each index batch of one randperm."""
    T, N = ro['rewards'].shape[:2]
    obs = t(ro['obs'][:-1]).reshape(T * N, -1)
    act = t(ro['actions']).reshape(T * N, -1)
    for idx in algo_ref.minibatch_indices(T * N, mini_batch_size=B):
        i = torch.as_tensor(idx)
        yield obs[i], act[i]


@pytest.mark.parametrize('ci', range(len(GAIL_CASES)))
def test_discriminator_update_oracle_vs_reference(ci):
    tag, od, A, hid, B, T, N = GAIL_CASES[ci]
    G = golden('gail')
    P, ro, expert = gail_inputs(ci)
    P = OrderedDict((k, t(v).clone()) for k, v in P.items())
    st = gail_ref.adam_init(P)
    for it in range(2):
        torch.manual_seed(900 + it)
        gen = policy_batches(ro, B)
        batches = []
        for es, ea in expert:                         # zip(expert_loader, generator): expert first, then policy, then alpha
            ps, pa = next(gen)
            alpha = torch.rand(B, 1)
            batches.append((torch.cat([ps, pa], 1), torch.cat([t(es), t(ea)], 1), alpha))
            # the reference interleaves draw and step; alpha does not depend on the parameters, so drawing here is the same stream
        # NB: randperm is drawn by the first next(gen) above, before the first alpha -- same order as gail.py:60-66,35
        loss = gail_ref.disc_update(P, st, batches)
        assert abs(loss - float(G[tag + '/update%d/loss' % it])) <= 2e-5 * abs(float(G[tag + '/update%d/loss' % it]))
        for n in P:
            np.testing.assert_allclose(P[n].numpy(), G[tag + '/update%d/param/%s' % (it, n)], rtol=2e-5, atol=2e-7)


@pytest.mark.parametrize('ci', range(len(GAIL_CASES)))
def test_predict_reward_oracle_vs_reference(ci):
    tag, od, A, hid, B, T, N = GAIL_CASES[ci]
    G = golden('gail')
    # the golden's predict_reward ran after two updates: take the reference's post-update parameters
    P = OrderedDict((k, t(G[tag + '/update1/param/' + k])) for k in gail_ref.disc_param_shapes(od + A, hid))
    _, ro, _ = gail_inputs(ci)
    rms, ret = gail_ref.RunningMeanStd(shape=()), None
    for step in range(3):
        r, ret = gail_ref.predict_reward(P, t(ro['obs'][step]), t(ro['actions'][step]), 0.99, t(ro['masks'][step]), ret, rms)
        np.testing.assert_allclose(r.numpy(), G[tag + '/reward%d' % step], rtol=2e-5, atol=1e-6)
        np.testing.assert_allclose(ret.numpy(), G[tag + '/returns%d' % step], rtol=2e-5, atol=1e-6)
        assert abs(float(rms.var[0]) - float(G[tag + '/rms_var%d' % step])) <= 1e-9 * max(1.0, float(rms.var[0]))


@pytest.mark.parametrize('ci', range(len(GAIL_CASES)))
def test_expert_dataset_oracle_vs_reference(ci):
    tag = GAIL_CASES[ci][0]
    G = golden('gail')
    torch.manual_seed(950 + ci)
    tr, get_idx = gail_ref.expert_dataset(gail_expert_file_tensors(ci), 4, 20)
    assert len(get_idx) == int(G[tag + '/ds/len'])
    assert np.array_equal(np.array(get_idx, dtype=np.int64), G[tag + '/ds/get_idx'])
    for k in ('states', 'actions', 'lengths'):
        assert np.array_equal(tr[k].numpy(), G[tag + '/ds/' + k])
