"""Pin the ACKTR/KFAC oracle against goldens from the reference (CPU).

Parity metric for the preconditioned update.  The Kronecker factors of the wide layers are rank deficient
at these batch sizes (e.g. the 1568x1568 FC factor from 640 samples), so the eigenbasis of their noise-level
null space -- and with it 1e-3 of the update -- changes with fp32 summation order; the reference run with a
different thread count differs from itself by that much.  The factors, losses and parameters are therefore
compared directly (2e-5 / 1e-4), and the update VECTOR is held to: distance to an fp64 run of the same
algorithm <= 2x the reference's own distance to it + 1e-3 (BASELINE.json: 1e-3 relative).
"""
from collections import OrderedDict

import numpy as np
import pytest
import torch

import synth
from helpers import (KFAC_CASES, check_summary, golden, kfac_golden, params_torch, reference_self_distance, shapes_for, t,
                     unsplit_name, update_vector_errors)
from oracle import kfac_ref, returns_ref


def kfac_inputs(ci):
    tag, kind, od, A, T, N = KFAC_CASES[ci]
    P = params_torch(synth.make_params(shapes_for(kind, od, A), 90 + ci))
    obs_shape = (4, 84, 84) if kind == 'cnn' else (od,)
    ro = synth.make_rollouts(T, N, obs_shape, 'discrete' if kind == 'cnn' else 'box', A, salt=820 + ci)
    returns_ref.compute_returns(ro['rewards'], ro['value_preds'], ro['returns'], ro['masks'], ro['bad_masks'],
                                ro['next_value'], False, 0.99, 0.95, False)
    return P, ro


def fp64_truth(ci, n_updates=3):
    """The oracle algorithm in double precision with the same CPU noise stream: list of per-update dicts."""
    tag, kind, od, A, T, N = KFAC_CASES[ci]
    P, ro = kfac_inputs(ci)
    P64 = OrderedDict((k, v.double()) for k, v in P.items())
    ro64 = {k: (t(v).double() if v.dtype == np.float32 else t(v)) for k, v in ro.items()}
    st = kfac_ref.new_state(Tf=2, fast_cnn=tag.endswith('_fast'))
    out = []
    for it in range(n_updates):
        torch.manual_seed(555 + it)
        noise = torch.randn(T, N, 1).double()
        _, info = kfac_ref.acktr_update(P64, st, ro64, 0.5, 0.01, noise=noise)
        d = {k: v.numpy().copy() for k, v in info['grads'].items()}
        d.update({'param:' + k: v.detach().numpy().copy() for k, v in P64.items()})
        out.append(d)
    return out


@pytest.mark.parametrize('ci', range(len(KFAC_CASES)))
def test_acktr_update_oracle_vs_reference(ci):
    tag, kind, od, A, T, N = KFAC_CASES[ci]
    G = kfac_golden(tag)
    P, ro = kfac_inputs(ci)
    ro = {k: t(v) for k, v in ro.items()}
    truth = fp64_truth(ci)
    st = kfac_ref.new_state(Tf=2, fast_cnn=tag.endswith('_fast'))
    mods = kfac_ref.module_list(P)
    assert len(mods) == len(G[tag + '/module_classes'])
    names = [str(n) for n in G[tag + '/param_names']]
    for it in range(3):
        torch.manual_seed(555 + it)
        losses, info = kfac_ref.acktr_update(P, st, ro, 0.5, 0.01)
        np.testing.assert_allclose(np.array(losses), G[tag + '/update%d/losses' % it], rtol=2e-5, atol=1e-7)
        for i, (key, kindm, pname, lname) in enumerate(mods):
            check_summary(G, tag + '/update%d/m_aa/%d' % (it, i), st['m_aa'][key].numpy(), 2e-5, tag)
            check_summary(G, tag + '/update%d/m_gg/%d' % (it, i), st['m_gg'][key].numpy(), 2e-5, tag)
        e_ref, e_mine, e_mg = update_vector_errors(G, tag + '/update%d/kfgrad/' % it + '%s', names,
                                                   lambda n: info['grads'][unsplit_name(n)].numpy(),
                                                   lambda n: truth[it][unsplit_name(n)])
        assert e_mine <= 2.0 * e_ref + 1e-3, (it, e_ref, e_mine, e_mg)
        p_ref, p_mine, _ = update_vector_errors(G, tag + '/update%d/param/' % it + '%s', names,
                                                lambda n: P[unsplit_name(n)].detach().numpy(),
                                                lambda n: truth[it]['param:' + unsplit_name(n)])
        assert p_mine <= 2.0 * p_ref + 1e-4, (it, p_ref, p_mine)


def test_reference_self_distance():
    """Pins the claim the KFAC parity bars rest on: the reference's preconditioned update for the NatureCNN case moves by
    MORE than north_star's 1e-3 against itself when nothing but the intra-op thread count of its CPU run changes
    (tests/golden/kfac.npz: torch.set_num_threads(1) as main.py:38; tests/golden/kfac_t8.npz: 8 threads; both written by
    make_golden_kfac.py from the imported reference), and both runs sit further than 1e-3 from the fp64 run of the same
    algorithm -- the rank-deficient 1568 x 1568 factor (640 samples) amplifies fp32 summation-order noise.  The MLP case
    (every factor full rank) reproduces itself exactly, and so do the post-update parameters (to 1e-5) and the A2C /
    RMSprop parameters (to 1e-6): for those the direct 1e-3 / 1e-4 bars against the golden apply unchanged.
    Recorded values (update 0, 1, 2): 1-vs-8 threads 1.53e-3, 1.30e-3, 3.91e-3; 1 thread vs fp64 2.12e-3, 1.88e-3,
    6.84e-3; 8 threads vs fp64 1.69e-3, 1.70e-3, 5.79e-3."""
    G1, G8 = golden('kfac'), golden('kfac_t8')
    assert int(G8['threads']) == 8
    truth = fp64_truth(0)
    for tag in ('cnn', 'mlp'):
        names = [str(n) for n in G1[tag + '/param_names']]
        for it in range(3):
            pf = tag + '/update%d/kfgrad/' % it + '%s'
            d = reference_self_distance(pf, names)
            dp = reference_self_distance(tag + '/update%d/param/' % it + '%s', names)
            print(tag, 'update', it, 'reference 1-vs-8 threads: update vector %.3g, parameters %.3g' % (d, dp))
            assert dp <= 1e-5
            if tag == 'mlp':
                assert d <= 1e-6
                continue
            assert d > 1e-3, 'the reference now agrees with itself to 1e-3: tighten the bars in test_kfac_gpu.py'
            for Gx, lo in ((G1, 1.5e-3), (G8, 1.5e-3)):
                e, _, _ = update_vector_errors(Gx, pf, names, lambda n: truth[it][unsplit_name(n)],
                                               lambda n: truth[it][unsplit_name(n)])
                assert e > lo, (it, e)
    Ga = golden('a2c')
    for tag in ('cnn', 'mlp'):
        names = [k[len(tag + '/update0/param/'):-5] for k in Ga.files
                 if k.startswith(tag + '/update0/param/') and k.endswith('#norm')]
        for it in range(3):
            d = reference_self_distance(tag + '/update%d/param/' % it + '%s', names, base='a2c',
                                        other_prefix_fmt='a2c/' + tag + '/update%d/param/' % it + '%s')
            assert d <= 1e-6, (tag, it, d)
