"""GPU parity: RolloutStorage on the HIP path vs the oracle and the reference goldens."""
import numpy as np
import pytest
import torch

import synth
from helpers import golden, t
from oracle import algo_ref, returns_ref

pytestmark = pytest.mark.gpu


class Box(object):
    def __init__(self, d):
        self.shape = (d,)


def make_storage(ro, obs_shape, A, dev='cuda'):
    from a2c_ppo_acktr.storage import RolloutStorage
    T, N = ro['rewards'].shape[:2]
    rs = RolloutStorage(T, N, obs_shape, Box(A), 1)
    for k in ('obs', 'rewards', 'value_preds', 'returns', 'action_log_probs', 'actions', 'masks', 'bad_masks'):
        getattr(rs, k).copy_(t(ro[k]))
    rs.to(dev)
    return rs


@pytest.mark.parametrize('T,N', [(5, 16), (128, 8), (2048, 1), (257, 33)])
@pytest.mark.parametrize('use_gae', [0, 1])
@pytest.mark.parametrize('proper', [0, 1])
def test_compute_returns_bit_exact_vs_reference_golden(T, N, use_gae, proper):
    G = golden('returns')
    ro = synth.make_rollouts(T, N, (3,), 'box', 2, salt=7 * T + N, p_mask=0.95, p_bad=0.96)
    rs = make_storage(ro, (3,), 2)
    rs.compute_returns(t(ro['next_value']).cuda(), bool(use_gae), 0.99, 0.95, bool(proper))
    key = 'T%d_N%d_gae%d_proper%d' % (T, N, use_gae, proper)
    got = rs.returns.cpu().numpy()
    assert np.array_equal(got.view(np.uint32), G[key + '/returns'].view(np.uint32))
    assert np.array_equal(rs.value_preds[-1].cpu().numpy().view(np.uint32),
                          G[key + '/value_preds_last'].view(np.uint32))


# (N <= 16 and N >= 4065 run the 32-environment form of the kernel, everything between the 16-environment form; 4100 / 4099
# cover its 16-byte and scalar loaders with a ragged last workgroup, 130 / 300 / 700 the chunk tails of both forms)
@pytest.mark.parametrize('T,N', [(1, 1), (63, 31), (64, 32), (65, 100), (700, 257), (130, 4100), (67, 4099), (300, 12)])
@pytest.mark.parametrize('mode', [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_compute_returns_bit_exact_vs_oracle_ragged(T, N, mode):
    use_gae, proper = mode
    ro = synth.make_rollouts(T, N, (1,), 'box', 1, salt=31 * T + N, p_mask=0.9, p_bad=0.9)
    rs = make_storage(ro, (1,), 1)
    rs.compute_returns(t(ro['next_value']).cuda(), bool(use_gae), 0.97, 0.9, bool(proper))
    vp, ret = ro['value_preds'].copy(), ro['returns'].copy()
    returns_ref.compute_returns(ro['rewards'], vp, ret, ro['masks'], ro['bad_masks'], ro['next_value'],
                                bool(use_gae), 0.97, 0.9, bool(proper))
    assert np.array_equal(rs.returns.cpu().numpy().view(np.uint32), ret.view(np.uint32))
    assert np.array_equal(rs.value_preds.cpu().numpy().view(np.uint32), vp.view(np.uint32))


def test_compute_returns_full_size_properties():
    """BASELINE config 5 size (T=2048, N=1024): properties that need no CPU scan of the full tensor:
    (i) envs are independent -> a column subset scanned by the oracle must match bit for bit;
    (ii) with masks == 0 everywhere, GAE returns collapse to r + (V - V) rows exactly as the oracle on the subset."""
    T, N = 2048, 1024
    g = torch.Generator().manual_seed(5)
    ro = dict(rewards=torch.randn(T, N, 1, generator=g), value_preds=torch.randn(T + 1, N, 1, generator=g),
              returns=torch.zeros(T + 1, N, 1), masks=(torch.rand(T + 1, N, 1, generator=g) < 0.99).float(),
              bad_masks=(torch.rand(T + 1, N, 1, generator=g) < 0.995).float())
    nv = torch.randn(N, 1, generator=g)
    from a2c_ppo_acktr.storage import RolloutStorage
    rs = RolloutStorage(T, N, (17,), Box(6), 1)
    for k, v in ro.items():
        getattr(rs, k).copy_(v)
    rs.to('cuda')
    for use_gae, proper in [(1, 1), (1, 0), (0, 1), (0, 0)]:
        rs.value_preds.copy_(ro['value_preds'])
        rs.compute_returns(nv.cuda(), bool(use_gae), 0.99, 0.95, bool(proper))
        cols = [0, 1, 31, 32, 33, 500, 1022, 1023]
        sub = {k: v[:, cols].numpy().copy() for k, v in ro.items()}
        returns_ref.compute_returns(sub['rewards'], sub['value_preds'], sub['returns'], sub['masks'],
                                    sub['bad_masks'], nv[cols].numpy(), bool(use_gae), 0.99, 0.95, bool(proper))
        got = rs.returns[:, cols].cpu().numpy()
        assert np.array_equal(got.view(np.uint32), sub['returns'].view(np.uint32))


def test_adv_normalize_and_sampler_vs_reference_golden():
    G = golden('sampler')
    T, N = 128, 8
    ro = synth.make_rollouts(T, N, (11,), 'box', 3, salt=501)
    rs = make_storage(ro, (11,), 3)
    rs.compute_returns(t(ro['next_value']).cuda(), True, 0.99, 0.95, True)
    adv = rs.normalized_advantages()
    np.testing.assert_allclose(adv.cpu().numpy(), G['adv'], rtol=2e-5, atol=2e-6)   # fp32, 1e-4 budget
    adv_ref = t(G['adv']).cuda()
    torch.manual_seed(1234)
    for e in range(2):
        for b, s in enumerate(rs.feed_forward_generator(adv_ref, 4)):
            obs_b, rhs_b, act_b, vp_b, ret_b, m_b, olp_b, adv_b = s
            assert np.array_equal(adv_b.cpu().numpy(), G['e%d_b%d/adv' % (e, b)])      # gathers are exact
            assert np.array_equal(ret_b.cpu().numpy(), G['e%d_b%d/ret' % (e, b)])
            assert np.array_equal(act_b.cpu().numpy(), G['e%d_b%d/act' % (e, b)])
            np.testing.assert_allclose(obs_b.double().sum(1).cpu().numpy(), G['e%d_b%d/obs_sum' % (e, b)],
                                       rtol=1e-12)
            assert rhs_b.shape == (256, 1) and m_b.shape == (256, 1) and vp_b.shape == (256, 1)
    torch.manual_seed(99)
    nb = 0
    for b, s in enumerate(rs.feed_forward_generator(None, mini_batch_size=100)):
        assert s[7] is None
        assert np.array_equal(s[6].cpu().numpy(), G['gail_b%d/olp' % b])
        nb += 1
    assert nb == int(G['gail_nb'])


def test_gather_discrete_actions_and_images():
    """int64 actions and 4x84x84 fp32 frames (112,896-B rows) gather exactly."""
    from a2c_ppo_acktr.storage import RolloutStorage

    class Discrete(object):
        n = 6
    T, N = 6, 5
    rs = RolloutStorage(T, N, (4, 84, 84), Discrete(), 1)
    g = torch.Generator().manual_seed(3)
    rs.obs.copy_(torch.randint(0, 256, rs.obs.shape, generator=g).float())
    rs.actions.copy_(torch.randint(0, 6, rs.actions.shape, generator=g))
    cpu_obs, cpu_act = rs.obs.clone(), rs.actions.clone()
    rs.to('cuda')
    torch.manual_seed(7)
    perm = torch.randperm(T * N)
    torch.manual_seed(7)
    for b, s in enumerate(rs.feed_forward_generator(None, 3)):
        idx = perm[b * 10:(b + 1) * 10]
        assert torch.equal(s[0].cpu(), cpu_obs[:-1].view(T * N, 4, 84, 84)[idx])
        assert s[2].dtype == torch.int64 and torch.equal(s[2].cpu(), cpu_act.view(T * N, 1)[idx])
