"""GPU parity of the GAIL discriminator through the C-ABI against goldens produced by the
reference's own gail.py (tests/golden/make_golden_gail.py).  Tolerance: losses and post-update parameters 1e-4 rel.
`RunningMeanStd` (sb3, not in the container) is the one parity-unpinned call: the golden harness used the same
restatement that ships in a2c_ppo_acktr/algo/gail.py."""
import os
import tempfile
from collections import OrderedDict

import numpy as np
import pytest
import torch

from helpers import GAIL_CASES, gail_expert_file_tensors, gail_inputs, golden, rel_err, t
from test_policy_gpu import Box, make_storage

pytestmark = pytest.mark.gpu


class FixedLoader(object):
    def __init__(self, batches, batch_size):
        self.batches, self.batch_size = batches, batch_size

    def __iter__(self):
        return iter(self.batches)


def make_disc(ci):
    from a2c_ppo_acktr.algo import gail
    tag, od, A, hid, B, T, N = GAIL_CASES[ci]
    P, ro, expert = gail_inputs(ci)
    disc = gail.Discriminator(od + A, hid, torch.device('cuda'))
    disc.load_state_dict(OrderedDict((k, t(v)) for k, v in P.items()), strict=True)
    return disc, P, ro, expert


@pytest.mark.parametrize('ci', range(len(GAIL_CASES)))
def test_discriminator_update_vs_reference(ci):
    tag, od, A, hid, B, T, N = GAIL_CASES[ci]
    G = golden('gail')
    disc, P, ro, expert = make_disc(ci)
    assert [n for n, _ in disc.named_parameters()] == list(P.keys())
    rs = make_storage(ro, (od,), Box(A))
    loader = FixedLoader([(t(s), t(a)) for s, a in expert], B)
    for it in range(2):
        torch.manual_seed(900 + it)
        loss = disc.update(loader, rs, obsfilt=lambda x, update=False: x)
        ref = float(G[tag + '/update%d/loss' % it])
        assert abs(loss - ref) <= 1e-4 * abs(ref), (it, loss, ref)
        for n, p in disc.named_parameters():
            e = rel_err(p.detach().cpu().numpy(), G[tag + '/update%d/param/%s' % (it, n)])
            assert e <= 1e-4, (it, n, e)


class OnlyGenerator(object):
    """A rollout storage that offers nothing but the reference's generator (storage.py:107): update() then takes the
    gathered 8-tuples instead of this package's index batches."""

    def __init__(self, rs):
        self.feed_forward_generator = rs.feed_forward_generator


@pytest.mark.parametrize('ci', range(len(GAIL_CASES)))
def test_discriminator_update_index_batches_equal_gathered_batches(ci):
    """update() over index batches (gather fused into the first kernel) == update() over gathered tuples, bit for bit,
    eager and replayed (four updates: two warm steps, capture, replay)."""
    tag, od, A, hid, B, T, N = GAIL_CASES[ci]
    out = []
    for light in (True, False):
        disc, P, ro, expert = make_disc(ci)
        rs = make_storage(ro, (od,), Box(A))
        loader = FixedLoader([(t(s), t(a)) for s, a in expert], B)
        losses = []
        for it in range(4):
            torch.manual_seed(900 + it)
            losses.append(disc.update(loader, rs if light else OnlyGenerator(rs), obsfilt=lambda x, update=False: x))
        out.append((losses, disc.flat_params.clone()))
    assert out[0][0] == out[1][0]
    assert torch.equal(out[0][1], out[1][1])


@pytest.mark.parametrize('ci', range(len(GAIL_CASES)))
def test_discriminator_gradients_vs_oracle(ci):
    """Both loss terms and every parameter gradient of one minibatch against the oracle's autograd (fp32 CPU)."""
    from oracle import gail_ref
    tag, od, A, hid, B, T, N = GAIL_CASES[ci]
    disc, P, ro, expert = make_disc(ci)
    g = torch.Generator().manual_seed(5)
    flat_obs = t(ro['obs'][:-1].reshape(-1, od))[:B].contiguous()
    flat_act = t(ro['actions'].reshape(-1, A))[:B].contiguous()
    es, ea = t(expert[0][0]), t(expert[0][1])
    alpha = torch.rand(B, 1, generator=g)
    loss = disc._losses_and_grads(flat_obs.cuda(), flat_act.cuda(), es.cuda(), ea.cuda(), alpha)
    Pg = OrderedDict((k, t(v).clone().requires_grad_(True)) for k, v in P.items())
    gl, gp = gail_ref.disc_losses(Pg, torch.cat([flat_obs, flat_act], 1), torch.cat([es, ea], 1), alpha)
    (gl + gp).backward()
    got = loss.cpu().numpy()
    assert abs(got[0] - gl.item()) <= 1e-4 * abs(gl.item()) and abs(got[1] - gp.item()) <= 1e-4 * abs(gp.item()), (got, gl, gp)
    for i, (n, p) in enumerate(Pg.items()):
        e = rel_err(disc._gview(i).cpu().numpy(), p.grad.numpy())
        assert e <= 1e-4, (n, e)


@pytest.mark.parametrize('od,A,hid,B', [(10, 3, 50, 33), (5, 2, 36, 7), (17, 6, 100, 300)])
def test_discriminator_gradients_odd_shapes_vs_oracle(od, A, hid, B):
    """Shapes where the row blocks of the stacked [4B, D] operands start at unaligned addresses (odd D, odd B)."""
    from a2c_ppo_acktr.algo import gail
    from oracle import gail_ref
    import synth
    P = synth.make_params(gail_ref.disc_param_shapes(od + A, hid), 320 + B)
    disc = gail.Discriminator(od + A, hid, torch.device('cuda'))
    disc.load_state_dict(OrderedDict((k, t(v)) for k, v in P.items()), strict=True)
    ps, pa, es, ea = (t(synth.normalish(shp, 800 + i + B)) for i, shp in enumerate(((B, od), (B, A), (B, od), (B, A))))
    alpha = torch.rand(B, 1, generator=torch.Generator().manual_seed(B))
    loss = disc._losses_and_grads(ps.cuda(), pa.cuda(), es.cuda(), ea.cuda(), alpha)
    Pg = OrderedDict((k, t(v).clone().requires_grad_(True)) for k, v in P.items())
    gl, gp = gail_ref.disc_losses(Pg, torch.cat([ps, pa], 1), torch.cat([es, ea], 1), alpha)
    (gl + gp).backward()
    got = loss.cpu().numpy()
    assert abs(got[0] - gl.item()) <= 1e-4 * abs(gl.item()) and abs(got[1] - gp.item()) <= 1e-4 * abs(gp.item()), (got, gl, gp)
    for i, (n, p) in enumerate(Pg.items()):
        e = rel_err(disc._gview(i).cpu().numpy(), p.grad.numpy())
        assert e <= 1e-4, (n, e)


@pytest.mark.parametrize('ci', range(len(GAIL_CASES)))
def test_predict_reward_vs_reference(ci):
    tag, od, A, hid, B, T, N = GAIL_CASES[ci]
    G = golden('gail')
    disc, P, ro, _ = make_disc(ci)
    disc.load_state_dict(OrderedDict((k, t(G[tag + '/update1/param/' + k])) for k in P), strict=True)
    for step in range(3):
        r = disc.predict_reward(t(ro['obs'][step]).cuda(), t(ro['actions'][step]).cuda(), 0.99, t(ro['masks'][step]).cuda())
        np.testing.assert_allclose(r.cpu().numpy(), G[tag + '/reward%d' % step], rtol=1e-4, atol=2e-5)
        np.testing.assert_allclose(disc.returns.cpu().numpy(), G[tag + '/returns%d' % step], rtol=1e-4, atol=2e-5)
        assert abs(float(disc.ret_rms.var[0]) - float(G[tag + '/rms_var%d' % step])) <= 1e-4 * float(G[tag + '/rms_var%d' % step])
    # the batched form gives the same per-step results
    disc2, _, _, _ = make_disc(ci)
    disc2.load_state_dict(OrderedDict((k, t(G[tag + '/update1/param/' + k])) for k in P), strict=True)
    rb = disc2.predict_reward_batched(t(ro['obs'][:3]).cuda(), t(ro['actions'][:3]).cuda(), 0.99, t(ro['masks'][:3]).cuda())
    for step in range(3):
        np.testing.assert_allclose(rb[step].cpu().numpy(), G[tag + '/reward%d' % step], rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize('ci', range(len(GAIL_CASES)))
def test_expert_dataset_vs_reference(ci):
    from a2c_ppo_acktr.algo import gail
    tag = GAIL_CASES[ci][0]
    G = golden('gail')
    with tempfile.TemporaryDirectory() as d:
        fn = os.path.join(d, 'trajs_synth.pt')
        torch.save(dict(gail_expert_file_tensors(ci)), fn)
        torch.manual_seed(950 + ci)
        ds = gail.ExpertDataset(fn, num_trajectories=4, subsample_frequency=20)
    assert len(ds) == int(G[tag + '/ds/len'])
    assert np.array_equal(np.array(ds.get_idx, dtype=np.int64), G[tag + '/ds/get_idx'])
    for k in ('states', 'actions', 'lengths'):
        assert np.array_equal(ds.trajectories[k].numpy(), G[tag + '/ds/' + k])
    assert np.array_equal(ds[len(ds) - 1][0].numpy(), G[tag + '/ds/last_state'])
