"""GPU parity of the KFAC pieces and of the full ACKTR update (HIP path through the C-ABI)."""
import ctypes
from collections import OrderedDict

import numpy as np
import pytest
import torch

import synth
from helpers import (KFAC_CASES, check_summary, golden, kfac_golden, reference_self_distance, t, unsplit_name,
                     update_vector_errors)
from oracle import kfac_ref
from test_policy_gpu import Box, Discrete, make_policy, make_storage

pytestmark = pytest.mark.gpu


def rel(a, b):
    a = np.asarray(a, np.float64).reshape(-1)
    b = np.asarray(b, np.float64).reshape(-1)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


# (B >= 64 on the 84x84x4, 20x20x32 and 9x9x64 layers runs the stationary-accumulator kernels of kfac_cov_sa.hpp, ragged ranges
# at 70, 300 and 333)
@pytest.mark.parametrize('H,K,S,C,B', [(84, 8, 4, 4, 7), (20, 4, 2, 32, 9), (9, 3, 1, 64, 40), (84, 8, 4, 4, 70), (84, 8, 4, 4, 300),
                                       (20, 4, 2, 32, 70), (20, 4, 2, 32, 333), (9, 3, 1, 64, 70), (9, 3, 1, 64, 333)])
def test_cov_a_conv_vs_oracle(H, K, S, C, B):
    from a2c_ppo_acktr import _hip, ops
    g = torch.Generator().manual_seed(H + B)
    x = torch.rand(B, C, H, H, generator=g)
    ref = kfac_ref.cov_a(x, 'conv', (K, K), S)
    L = _hip.lib()
    Kc = C * K * K
    OH = (H - K) // S + 1
    out = torch.empty(Kc, Kc, device='cuda')
    ws = ops.workspace('cuda', L.acktr_kfac_cov_workspace_bytes(Kc, B * OH * OH), 'kfac')
    xd = x.cuda()
    rc = L.acktr_kfac_cov_a_conv_f32(_hip.dev_ptr(xd), None, B, C, H, H, K, K, S, 1.0 / (B * float(OH * OH) ** 2),
                                     _hip.dev_ptr(out), _hip.dev_ptr(ws), ws.numel(), _hip.cur_stream())
    _hip.check(rc, 'cov_a_conv')
    assert rel(out.cpu().numpy(), ref.numpy()) < 2e-5
    assert torch.equal(out, out.t()), 'factor must be exactly symmetric'


@pytest.mark.parametrize('OC,SP,B', [(32, 400, 5), (64, 81, 33), (32, 49, 100)])
def test_cov_g_conv_and_bias_vs_oracle(OC, SP, B):
    from a2c_ppo_acktr import _hip, ops
    g = torch.Generator().manual_seed(OC + SP)
    side = int(round(SP ** 0.5))
    dy = torch.randn(B, OC, side, side, generator=g) / B
    L = _hip.lib()
    out = torch.empty(OC, OC, device='cuda')
    ws = ops.workspace('cuda', L.acktr_kfac_cov_workspace_bytes(OC, B * SP), 'kfac')
    dyd = dy.cuda()
    rc = L.acktr_kfac_cov_g_conv_f32(_hip.dev_ptr(dyd), B, OC, SP, float(SP) * B, _hip.dev_ptr(out), _hip.dev_ptr(ws),
                                     ws.numel(), _hip.cur_stream())
    _hip.check(rc, 'cov_g_conv')
    assert rel(out.cpu().numpy(), kfac_ref.cov_g(dy, 'conv').numpy()) < 2e-5
    ssum = torch.empty(B, OC, device='cuda')
    _hip.check(L.acktr_spatial_sum_f32(_hip.dev_ptr(dyd), B * OC, SP, _hip.dev_ptr(ssum), _hip.cur_stream()), 'ssum')
    _hip.check(L.acktr_kfac_cov_dense_f32(_hip.dev_ptr(ssum), OC, B, OC, float(B), _hip.dev_ptr(out), _hip.dev_ptr(ws),
                                          ws.numel(), _hip.cur_stream()), 'cov_dense')
    assert rel(out.cpu().numpy(), kfac_ref.cov_g(dy, 'bias').numpy()) < 2e-5


def test_running_stat_bit_exact():
    from a2c_ppo_acktr import _hip
    g = torch.Generator().manual_seed(1)
    m = torch.randn(1000, generator=g)
    x = torch.randn(1000, generator=g)
    ref = m.clone()
    kfac_ref.running_stat(x, ref, 0.99)
    md = m.cuda()
    _hip.check(_hip.lib().acktr_kfac_running_stat_f32(_hip.dev_ptr(md), _hip.dev_ptr(x.cuda()), 1000, 0.99 / (1 - 0.99), 1 - 0.99, 0,
                                                      _hip.cur_stream()), 'rs')
    assert torch.equal(md.cpu(), ref)
    ref0 = x.clone()
    kfac_ref.running_stat(x, ref0, 0.99)
    _hip.check(_hip.lib().acktr_kfac_running_stat_f32(_hip.dev_ptr(md), _hip.dev_ptr(x.cuda()), 1000, 0.99 / (1 - 0.99), 1 - 0.99, 1,
                                                      _hip.cur_stream()), 'rs')
    assert torch.equal(md.cpu(), ref0)


def _psd(n, rank, gen):
    a = torch.randn(rank, n, generator=gen)
    return (a.t() @ a / rank).contiguous()


@pytest.mark.parametrize('sizes', [[1, 6, 32, 64, 7, 33], [128, 100, 1, 2]])
def test_jacobi_batched(sizes):
    """Eigenvalues ascending, thresholded, Q Lambda Q^T reconstructs the input, rows orthonormal (never compare Q)."""
    from a2c_ppo_acktr import _hip
    from a2c_ppo_acktr.algo.kfac import _JacobiDesc
    g = torch.Generator().manual_seed(sum(sizes))
    mats = [_psd(n, max(1, n // 2) if i % 2 else n + 3, g).cuda() for i, n in enumerate(sizes)]
    ds = [torch.empty(n, device='cuda') for n in sizes]
    qts = [torch.empty(n, n, device='cuda') for n in sizes]
    arr = (_JacobiDesc * len(sizes))(*[_JacobiDesc(m.data_ptr(), d.data_ptr(), q.data_ptr(), n)
                                        for m, d, q, n in zip(mats, ds, qts, sizes)])
    descs = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).cuda()
    rc = _hip.lib().acktr_sym_eig_jacobi_batched_f32(_hip.dev_ptr(descs), len(sizes), max(sizes), 1e-6,
                                                     _hip.cur_stream())
    _hip.check(rc, 'jacobi')
    for m, d, q, n in zip(mats, ds, qts, sizes):
        A = m.cpu().double().numpy()
        w = np.linalg.eigvalsh(A)
        dd = d.cpu().double().numpy()
        Q = q.cpu().double().numpy().T
        assert np.all(np.diff(dd) >= -1e-7 * max(w.max(), 1e-6)), 'ascending'
        w_thr = np.where(w > 1e-6, w, 0.0)
        assert np.abs(dd - w_thr).max() <= 3e-5 * max(w.max(), 1.0) + 2e-6   # fp32 Jacobi: ~10 sweeps of rotations
        assert np.abs(Q.T @ Q - np.eye(n)).max() < 3e-5
        dd_raw = np.where(dd > 0, dd, 0.0)
        recon = Q @ np.diag(dd_raw) @ Q.T
        assert np.linalg.norm(recon - A) <= 5e-5 * np.linalg.norm(A) + 2e-6 * n


def _block_jacobi(mats, max_sweeps=0, thresh=1e-6):
    from a2c_ppo_acktr import _hip
    L = _hip.lib()
    ns = [m.shape[0] for m in mats]
    ds = [torch.full((n,), float('nan'), device='cuda') for n in ns]
    qts = [torch.full((n, n), float('nan'), device='cuda') for n in ns]
    jobs = (_hip.JacobiJob * len(ns))(*[_hip.JacobiJob(m.data_ptr(), d.data_ptr(), q.data_ptr(), n)
                                         for m, d, q, n in zip(mats, ds, qts, ns)])
    cns = (ctypes.c_int * len(ns))(*ns)
    nbytes = L.acktr_sym_eig_jacobi_block_workspace_bytes(cns, len(ns))
    assert nbytes > 0
    ws = torch.empty(nbytes, dtype=torch.uint8, device='cuda')
    sweeps = (ctypes.c_int * len(ns))()
    rc = L.acktr_sym_eig_jacobi_block_f32(jobs, len(ns), thresh, _hip.dev_ptr(ws), ws.numel(), max_sweeps, sweeps,
                                          _hip.cur_stream())
    _hip.check(rc, 'jacobi_block')
    torch.cuda.synchronize()
    return ds, qts, list(sweeps)


@pytest.mark.parametrize('sizes', [[256, 512, 576, 1024], [129, 200, 33, 64], [1568], [2048, 1000]])
def test_jacobi_block(sizes):
    """Blocked one-sided Jacobi (csrc/jacobi_block.hip) for 32 < n <= 2048, several factors per call: eigenvalues
    ascending and thresholded, Q Lambda Q^T reconstructs the input, rows orthonormal (never compare Q itself).
    Mix of full-rank and rank-deficient PSD inputs (the KFAC factors are both)."""
    g = torch.Generator().manual_seed(sum(sizes))
    mats = [_psd(n, max(1, n // 3) if i % 2 == 0 else n + 3, g).cuda() for i, n in enumerate(sizes)]
    keep = [m.clone() for m in mats]
    ds, qts, sweeps = _block_jacobi(mats, max_sweeps=48)
    print('sizes', sizes, 'sweeps', sweeps)
    # (random rank-deficient inputs converge slowly: 1568 columns in a 522-dimensional range take ~30 sweeps)
    assert all(0 < s_ <= 48 for s_ in sweeps), sweeps
    for m, k_, d, q, n in zip(mats, keep, ds, qts, sizes):
        assert torch.equal(m, k_), 'the input factor must not be modified'
        A = m.cpu().double().numpy()
        w = np.linalg.eigvalsh(A)
        dd = d.cpu().double().numpy()
        Q = q.cpu().double().numpy().T
        assert np.isfinite(dd).all() and np.isfinite(Q).all()
        assert np.all(np.diff(dd) >= 0), 'ascending'
        w_thr = np.where(w > 1e-6, w, 0.0)
        assert np.abs(dd - w_thr).max() <= 2e-5 * max(w.max(), 1.0) + 2e-6
        assert np.abs(Q.T @ Q - np.eye(n)).max() < 5e-5
        recon = Q @ np.diag(dd) @ Q.T
        assert np.linalg.norm(recon - A) <= 2e-5 * np.linalg.norm(A) + 2e-6 * n


def test_jacobi_block_relative_accuracy_of_small_eigenvalues():
    """The property the KFAC update needs: on a GRADED positive definite matrix A = D C D (C well
    conditioned, D spanning 2.5 decades) one-sided Jacobi with a relative rotation criterion gets every eigenvalue to a
    small RELATIVE error (Demmel-Veselic), where a tridiagonalisation-based fp32 solver is only accurate to
    eps * lambda_max in absolute terms and loses the small ones."""
    n = 384
    g = torch.Generator().manual_seed(5)
    X = torch.randn(n, 2 * n, generator=g, dtype=torch.float64)
    C = X @ X.t() / (2 * n)
    dC = C.diagonal().rsqrt()
    C = C * dC[:, None] * dC[None, :]
    D = torch.logspace(0, -2.5, n, dtype=torch.float64)[torch.randperm(n, generator=g)]
    A = (C * D[:, None] * D[None, :]).float().cuda()
    (d,), (q,), sweeps = _block_jacobi([A], thresh=0.0)
    w = torch.linalg.eigvalsh(A.cpu().double())
    assert float(w.min()) > 0
    rel_mine = float(((d.cpu().double() - w).abs() / w).max())
    w32 = torch.linalg.eigvalsh(A.cpu())
    rel_lap = float(((w32.double() - w).abs() / w).max())
    print('sweeps', sweeps, 'max relative eigenvalue error: blocked Jacobi %.2e, LAPACK fp32 %.2e; lambda range %.1e .. %.1e'
          % (rel_mine, rel_lap, float(w.min()), float(w.max())))
    assert rel_mine < 1e-4 and rel_mine < rel_lap


@pytest.mark.parametrize('n', [256, 576, 1568, 3137])      # 3137 = 64*7*7+1: the OpenAI-baselines FC factor north_star names
def test_rocsolver_eig(n):
    from a2c_ppo_acktr import _hip
    g = torch.Generator().manual_seed(n)
    A = _psd(n, n // 3, g)
    q = A.clone().cuda()
    d = torch.empty(n, device='cuda')
    e = torch.empty(n, device='cuda')
    info = torch.zeros(1, dtype=torch.int32, device='cuda')
    rc = _hip.lib().acktr_sym_eig_rocsolver_f32(_hip.dev_ptr(q), n, _hip.dev_ptr(d), _hip.dev_ptr(e),
                                                _hip.dev_ptr(info, torch.int32), 1e-6, _hip.cur_stream())
    _hip.check(rc, 'rocsolver')
    assert int(info.item()) == 0
    w = np.linalg.eigvalsh(A.double().numpy())
    dd = d.cpu().double().numpy()
    assert np.abs(dd - np.where(w > 1e-6, w, 0.0)).max() <= 1e-5 * w.max()
    Q = q.cpu().double().numpy().T            # rows of qt are eigenvectors
    assert np.abs(Q.T @ Q - np.eye(n)).max() < 1e-4
    assert np.linalg.norm(Q @ np.diag(dd) @ Q.T - A.double().numpy()) <= 1e-4 * np.linalg.norm(A.numpy())


@pytest.mark.parametrize('OUT,IN', [(32, 256), (512, 1568), (6, 512), (64, 1), (1, 512)])
def test_precondition_vs_torch(OUT, IN):
    from a2c_ppo_acktr import _hip
    g = torch.Generator().manual_seed(OUT * 7 + IN)
    Ag, Aa = _psd(OUT, OUT + 5, g), _psd(IN, IN + 5, g)
    dg, Qg = torch.linalg.eigh(Ag)
    da, Qa = torch.linalg.eigh(Aa)
    G = torch.randn(OUT, IN, generator=g)
    ref = Qg @ ((Qg.t() @ G @ Qa) / (dg.unsqueeze(1) * da.unsqueeze(0) + 0.01)) @ Qa.t()
    dev = lambda x: x.contiguous().cuda()
    qtg, qta, dgd, dad, Gd = dev(Qg.t()), dev(Qa.t()), dev(dg), dev(da), dev(G)
    t1, t2, v = torch.empty(OUT * IN, device='cuda'), torch.empty(OUT * IN, device='cuda'), torch.empty(OUT, IN, device='cuda')
    rc = _hip.lib().acktr_kfac_precondition_f32(_hip.dev_ptr(Gd), _hip.dev_ptr(qtg), _hip.dev_ptr(dgd), _hip.dev_ptr(qta),
                                                _hip.dev_ptr(dad), OUT, IN, 0.01, _hip.dev_ptr(t1), _hip.dev_ptr(t2),
                                                _hip.dev_ptr(v), _hip.cur_stream())
    _hip.check(rc, 'precondition')
    assert rel(v.cpu().numpy(), ref.numpy()) < 5e-5


@pytest.mark.parametrize('H,K,S,C,B', [(84, 8, 4, 4, 5), (20, 4, 2, 32, 7), (9, 3, 1, 64, 3), (12, 5, 3, 2, 4)])
def test_patch_mean_vs_torch(H, K, S, C, B):
    """acktr_kfac_patch_mean_f32 (compute_cov_a with fast_cnn=True, kfac.py:33-36) against unfold + mean, with and
    without a gather index."""
    from a2c_ppo_acktr import _hip
    g = torch.Generator().manual_seed(H + B)
    x = torch.rand(B + 3, C, H, H, generator=g)
    idx = torch.tensor([(3 * i + 1) % (B + 3) for i in range(B)], dtype=torch.int64)
    for use_idx in (False, True):
        xs = x[idx] if use_idx else x[:B]
        p = xs.double().unfold(2, K, S).unfold(3, K, S).permute(0, 2, 3, 1, 4, 5).contiguous()
        ref = (p.view(B, -1, C * K * K).mean(1) * 0.25).float()
        out = torch.empty(B, C * K * K, device='cuda')
        rc = _hip.lib().acktr_kfac_patch_mean_f32(_hip.dev_ptr(x.cuda()), _hip.dev_ptr(idx.cuda()) if use_idx else None, B, C, H,
                                                 H, K, K, S, 0.25, _hip.dev_ptr(out), _hip.cur_stream())
        _hip.check(rc, 'patch_mean')
        assert rel(out.cpu().numpy(), ref.numpy()) < 2e-6


@pytest.mark.parametrize('with_q', [False, True])
def test_precondition_batched_vs_torch(with_q):
    """acktr_kfac_precondition_batched_f32: every layer shape class of the two policies in ONE call (16-byte and scalar
    loads, 1-column bias modules, 1- and 6-row heads, more jobs than one group holds) against torch fp64 -- on the grouped
    tile GEMM (jobs without q_g / q_a) and on the stationary-accumulator stages (with them)."""
    from a2c_ppo_acktr import _hip
    shapes = [(512, 1568), (64, 512), (32, 576), (32, 256), (1, 512), (6, 512), (512, 1), (64, 1), (32, 1), (6, 1), (1, 1),
              (64, 64), (64, 11), (3, 64), (64, 17), (130, 70)] + [(8, 12)] * 6
    g = torch.Generator().manual_seed(5)
    keep, refs, outs = [], [], []
    jobs = (_hip.PrecondJob * len(shapes))()
    for i, (OUT, IN) in enumerate(shapes):
        Ag, Aa = _psd(OUT, OUT + 5, g).double(), _psd(IN, IN + 5, g).double()
        dg, Qg = torch.linalg.eigh(Ag)
        da, Qa = torch.linalg.eigh(Aa)
        G = torch.randn(OUT, IN, generator=g).double()
        refs.append(Qg @ ((Qg.t() @ G @ Qa) / (dg.unsqueeze(1) * da.unsqueeze(0) + 0.01)) @ Qa.t())
        dev = lambda x: x.float().contiguous().cuda()
        ts = [dev(G), dev(Qg.t()), dev(dg), dev(Qa.t()), dev(da), torch.empty(OUT * IN, device='cuda'),
              torch.empty(OUT * IN, device='cuda'), torch.empty(OUT, IN, device='cuda')]
        qs = [dev(Qg), dev(Qa)]
        keep.append((ts, qs))
        outs.append(ts[7])
        jobs[i] = _hip.PrecondJob(*[_hip.dev_ptr(x) for x in ts], OUT, IN,
                                  *([_hip.dev_ptr(x) for x in qs] if with_q else [None, None]))
    _hip.check(_hip.lib().acktr_kfac_precondition_batched_f32(jobs, len(shapes), 0.01, _hip.cur_stream()), 'precondition_batched')
    for (OUT, IN), v, ref in zip(shapes, outs, refs):
        assert rel(v.cpu().numpy(), ref.numpy()) < 5e-5, (OUT, IN)


@pytest.mark.parametrize('ci', range(len(KFAC_CASES)))
def test_acktr_update_vs_reference(ci):
    """Full A2C_ACKTR(acktr=True).update x3 (eigendecompositions at steps 0 and 2).  Losses, factors and
    parameters against the reference goldens; the preconditioned update against the fp64 truth with the
    reference's own fp32 error as yardstick (see tests/test_oracle_kfac.py for why): the HIP path must be at least as
    close to fp64 as the reference's fp32 CPU run is (k = 1, + the north_star 1e-3).  Round 1 needed k = 4: the error
    sat in the eigensolver stage (tools/kfac_stage_err.py; rocSOLVER ssyevd 3.5e-3 against LAPACK's 1.6e-3 on the same
    factors); the blocked one-sided Jacobi (csrc/jacobi_block.hip) brings that stage to 4e-5..9e-5.  Measured
    e_mine / e_ref: 1.0e-3 / 2.1e-3, 9.1e-4 / 1.9e-3, 2.7e-3 / 6.8e-3 (1568 factor on rocSOLVER, the default);
    4.6e-4, 1.1e-4, 2.5e-3 with ACKTR_EIG_BLOCK_MAX_N=2048 (every factor on the Jacobi)."""
    from a2c_ppo_acktr import algo
    from test_oracle_kfac import fp64_truth
    tag, kind, od, A, T, N = KFAC_CASES[ci]
    G = kfac_golden(tag)
    pol = make_policy(kind, od, A, 90 + ci)
    obs_shape = (4, 84, 84) if kind == 'cnn' else (od,)
    ro = synth.make_rollouts(T, N, obs_shape, 'discrete' if kind == 'cnn' else 'box', A, salt=820 + ci)
    rs = make_storage(ro, obs_shape, Discrete(A) if kind == 'cnn' else Box(A))
    rs.compute_returns(t(ro['next_value']).cuda(), False, 0.99, 0.95, False)
    truth = fp64_truth(ci)
    agent = algo.A2C_ACKTR(pol, value_loss_coef=0.5, entropy_coef=0.01, acktr=True)
    agent.optimizer.Tf = 2
    agent.optimizer.fast_cnn = tag.endswith('_fast')       # public attribute, as the golden harness sets it (kfac.py:134)
    names = [str(n) for n in G[tag + '/param_names']]
    assert [n for n, _ in pol.named_parameters()] == names, 'split_bias must rename parameters like the reference'
    assert [m.__class__.__name__ for m in agent.optimizer.modules] == [str(c) for c in G[tag + '/module_classes']]
    params = dict(pol.named_parameters())
    for it in range(3):
        torch.manual_seed(555 + it)
        res = agent.update(rs)
        np.testing.assert_allclose(np.array(res), G[tag + '/update%d/losses' % it], rtol=1e-4, atol=1e-6)
        # factor bar: 1e-4.  CNN: 5e-4, because a ReLU pre-activation below fp32 resolution (|pre| ~ 1e-7, about one
        # per 10^6 activations, see helpers.cnn_gate_aware_grads) has no determined sign -- ours and the reference's
        # conv may gate it differently, and one such gate moves a grad-output factor by ~2e-4 (measured: m_gg of conv2)
        ftol = 1e-4 if kind == 'mlp' else 5e-4
        for i, m in enumerate(agent.optimizer.modules):
            check_summary(G, tag + '/update%d/m_aa/%d' % (it, i), agent.optimizer.m_aa[m].cpu().numpy(), ftol, tag)
            check_summary(G, tag + '/update%d/m_gg/%d' % (it, i), agent.optimizer.m_gg[m].cpu().numpy(), ftol, tag)
        e_ref, e_mine, e_mg = update_vector_errors(G, tag + '/update%d/kfgrad/' % it + '%s', names,
                                                   lambda n: params[n].grad.detach().cpu().numpy(),
                                                   lambda n: truth[it][unsplit_name(n)])
        print(tag, 'update', it, 'e_ref %.3g e_mine %.3g mine-vs-golden %.3g' % (e_ref, e_mine, e_mg))
        for n in names:       # per-tensor breakdown (diagnostic)
            r1, r2, r3 = update_vector_errors(G, tag + '/update%d/kfgrad/' % it + '%s', [n],
                                              lambda n: params[n].grad.detach().cpu().numpy(),
                                              lambda n: truth[it][unsplit_name(n)])
            print('    %-36s ref %.2e mine %.2e norm %.3g' % (n, r1, r2, float(G[tag + '/update%d/kfgrad/%s#norm' % (it, n)])))
        assert e_mine <= 1.0 * e_ref + 1e-3, (it, e_ref, e_mine, e_mg)
        # Directly against the reference golden (north_star: 1e-3 relative).  How well the reference pins its OWN answer
        # is a committed fixture: the same reference code run with 8 intra-op threads instead of 1 (tests/golden/
        # kfac_t8.npz; self-distances recorded in tests/test_oracle_kfac.py::test_reference_self_distance) and the
        # fp64 run of the same algorithm.  Where both agree with the golden to 5e-4 (MLP: every factor full rank) the
        # bar is 1e-3; elsewhere it is the reference's own uncertainty + 1e-3 (an EXACT solver sits at e_ref from the
        # golden, so no implementation can do better than that).
        if tag.endswith('_fast'):
            self_d = None                      # no 8-thread fixture for the fast_cnn variant: fp64 anchor only
            unc = e_ref
        else:
            self_d = reference_self_distance(tag + '/update%d/kfgrad/' % it + '%s', names)
            unc = max(self_d, e_ref)
        bar = 1e-3 if unc <= 5e-4 else unc + 1e-3        # (pinned to half the tolerance, or not)
        print(tag, 'update', it, 'ours-vs-golden %.3g  bar %.3g  (reference 1-vs-8 threads %s, reference-vs-fp64 %.3g, '
              'ours-vs-fp64 %.3g)' % (e_mg, bar, 'n/a' if self_d is None else '%.3g' % self_d, e_ref, e_mine))
        assert e_mg <= bar, (it, e_mg, bar, self_d, e_ref, e_mine)
        p_ref, p_mine, p_mg = update_vector_errors(G, tag + '/update%d/param/' % it + '%s', names,
                                                   lambda n: params[n].detach().cpu().numpy(),
                                                   lambda n: truth[it]['param:' + unsplit_name(n)])
        print(tag, 'update', it, 'params: e_ref %.3g e_mine %.3g mine-vs-golden %.3g' % (p_ref, p_mine, p_mg))
        assert p_mine <= 1.0 * p_ref + 1e-4 and p_mg <= 1e-3, (it, p_ref, p_mine, p_mg)


@pytest.mark.parametrize('ci', range(len(KFAC_CASES)))
def test_acktr_graph_replay_matches_eager(ci):
    """The hipGraph-replayed ACKTR update (every step that does not eigendecompose, from the third on) against the
    same updates launched one by one: same CPU noise stream, same losses, same parameters."""
    from a2c_ppo_acktr import algo
    tag, kind, od, A, T, N = KFAC_CASES[ci]
    obs_shape = (4, 84, 84) if kind == 'cnn' else (od,)
    ro = synth.make_rollouts(T, N, obs_shape, 'discrete' if kind == 'cnn' else 'box', A, salt=820 + ci)
    out = []
    for use_graph in (False, True):
        pol = make_policy(kind, od, A, 90 + ci)
        rs = make_storage(ro, obs_shape, Discrete(A) if kind == 'cnn' else Box(A))
        rs.compute_returns(t(ro['next_value']).cuda(), False, 0.99, 0.95, False)
        agent = algo.A2C_ACKTR(pol, value_loss_coef=0.5, entropy_coef=0.01, acktr=True)
        agent.use_graph = use_graph
        agent.optimizer.fast_cnn = tag.endswith('_fast')
        agent.optimizer.Tf = 4                       # eig at optimizer steps 0 and 4; replays at 2, 3, 5, 6
        agent.optimizer.lr = 0.02                    # keep seven updates on one rollout well behaved
        torch.manual_seed(77)
        losses = [agent.update(rs) for _ in range(7)]
        if use_graph:
            assert len(agent._kgraphs) == 1
        out.append((np.array(losses), pol.flat_params.detach().cpu().numpy().copy()))
    np.testing.assert_allclose(out[1][0], out[0][0], rtol=1e-5, atol=1e-7)
    err = np.linalg.norm(out[1][1] - out[0][1]) / np.linalg.norm(out[0][1])
    assert err < 1e-6, err


@pytest.mark.parametrize('ci', [0, 1])
def test_kfac_reference_call_pattern(ci):
    """KFACOptimizer driven the way the reference drives it -- the body of a2c_acktr.py:33-80 written out as user code over
    the drop-in API: evaluate_actions (forward-pre hooks -> m_aa), `acc_stats = True; fisher_loss.backward(retain_graph=
    True)` (backward hooks -> m_gg), the real loss's backward, optimizer.step().  Factors, losses and parameters against
    the reference goldens, like test_acktr_update_vs_reference; and step() without statistics must raise."""
    from a2c_ppo_acktr.algo.kfac import KFACOptimizer
    tag, kind, od, A, T, N = KFAC_CASES[ci]
    G = kfac_golden(tag)
    pol = make_policy(kind, od, A, 90 + ci)
    obs_shape = (4, 84, 84) if kind == 'cnn' else (od,)
    ro = synth.make_rollouts(T, N, obs_shape, 'discrete' if kind == 'cnn' else 'box', A, salt=820 + ci)
    rs = make_storage(ro, obs_shape, Discrete(A) if kind == 'cnn' else Box(A))
    rs.compute_returns(t(ro['next_value']).cuda(), False, 0.99, 0.95, False)
    optimizer = KFACOptimizer(pol)
    optimizer.Tf = 2
    with pytest.raises(RuntimeError, match='no Kronecker-factor statistics'):
        optimizer.step()
    assert optimizer.steps == 0
    names = [str(n) for n in G[tag + '/param_names']]
    params = dict(pol.named_parameters())
    value_loss_coef, entropy_coef = 0.5, 0.01
    for it in range(3):
        torch.manual_seed(555 + it)
        # ---- a2c_acktr.py:34-51
        num_steps, num_processes, _ = rs.rewards.size()
        values, action_log_probs, dist_entropy, _ = pol.evaluate_actions(
            rs.obs[:-1].view(-1, *obs_shape), rs.recurrent_hidden_states[0].view(-1, 1), rs.masks[:-1].view(-1, 1),
            rs.actions.view(-1, rs.actions.size(-1)))
        values = values.view(num_steps, num_processes, 1)
        action_log_probs = action_log_probs.view(num_steps, num_processes, 1)
        advantages = rs.returns[:-1] - values
        value_loss = advantages.pow(2).mean()
        action_loss = -(advantages.detach() * action_log_probs).mean()
        # ---- a2c_acktr.py:53-68
        if optimizer.steps % optimizer.Ts == 0:
            pol.zero_grad()
            pg_fisher_loss = -action_log_probs.mean()
            value_noise = torch.randn(values.size()).cuda()
            sample_values = values + value_noise
            vf_fisher_loss = -(values - sample_values.detach()).pow(2).mean()
            fisher_loss = pg_fisher_loss + vf_fisher_loss
            optimizer.acc_stats = True
            fisher_loss.backward(retain_graph=True)
            optimizer.acc_stats = False
        # ---- a2c_acktr.py:70-80
        optimizer.zero_grad()
        (value_loss * value_loss_coef + action_loss - dist_entropy * entropy_coef).backward()
        optimizer.step()
        res = (value_loss.item(), action_loss.item(), dist_entropy.item())
        np.testing.assert_allclose(np.array(res), G[tag + '/update%d/losses' % it], rtol=1e-4, atol=1e-6)
        ftol = 1e-4 if kind == 'mlp' else 5e-4
        for i, m in enumerate(optimizer.modules):
            check_summary(G, tag + '/update%d/m_aa/%d' % (it, i), optimizer.m_aa[m].cpu().numpy(), ftol, tag)
            check_summary(G, tag + '/update%d/m_gg/%d' % (it, i), optimizer.m_gg[m].cpu().numpy(), ftol, tag)
        _, _, p_mg = update_vector_errors(G, tag + '/update%d/param/' % it + '%s', names,
                                          lambda n: params[n].detach().cpu().numpy(),
                                          lambda n: params[n].detach().cpu().numpy())
        print(tag, 'update', it, 'hook-pattern params vs golden %.3g' % p_mg)
        assert p_mg <= 1e-3, (it, p_mg)
    assert optimizer.steps == 3
