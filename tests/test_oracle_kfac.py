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
from helpers import (KFAC_CASES, check_summary, kfac_golden, params_torch, shapes_for, t, unsplit_name,
                     update_vector_errors)
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
