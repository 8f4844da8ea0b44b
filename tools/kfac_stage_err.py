"""Where does the KFAC-preconditioned update pick up its fp32 error?  (VERDICT r1, next #1c)

ACKTR step 0 of the C4 case (NatureCNN, 32 procs x 20 steps, B = 640).  Truth = the oracle algorithm in fp64.  The
update v = Q_g [(Q_g^T G Q_a) / (d_g d_a^T + lambda)] Q_a^T is then recomputed in fp64 with ONE stage at a time
replaced by the fp32 result of (a) the HIP path and (b) torch CPU fp32 (what the reference runs):

    grads      the loss gradient G
    factors    m_aa / m_gg (SYRKs + running average)
    eigen      eigendecomposition of the (same, fp32-rounded truth) factors
    gemm       the four preconditioner GEMMs on the same (fp32-rounded truth) operands

Prints whole-update-vector relative errors per stage and per module.  Runs on the GPU box:
    python tools/kfac_stage_err.py
"""
import ctypes
import os
import sys
from collections import OrderedDict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'pytorch-a2c-ppo-acktr-gail_amd'), os.path.join(ROOT, 'tests'),
          os.path.join(ROOT, 'tests', 'golden')):
    sys.path.insert(0, p)

import numpy as np                      # noqa: E402
import torch                            # noqa: E402

import synth                            # noqa: E402
from helpers import KFAC_CASES, params_torch, shapes_for, t, unsplit_name   # noqa: E402
from oracle import kfac_ref, returns_ref                                    # noqa: E402

LA = 1e-2


def precond64(Gm, dg, Qg, da, Qa):
    v1 = Qg.t() @ Gm @ Qa
    v2 = v1 / (dg.unsqueeze(1) * da.unsqueeze(0) + LA)
    return Qg @ v2 @ Qa.t()


def stage_inputs(dtype):
    """factors, grads of step 0 computed by the oracle in `dtype`; dict per module key."""
    ci = 0
    tag, kind, od, A, T, N = KFAC_CASES[ci]
    P = params_torch(synth.make_params(shapes_for(kind, od, A), 90 + ci))
    ro = synth.make_rollouts(T, N, (4, 84, 84), 'discrete', A, salt=820 + ci)
    returns_ref.compute_returns(ro['rewards'], ro['value_preds'], ro['returns'], ro['masks'], ro['bad_masks'],
                                ro['next_value'], False, 0.99, 0.95, False)
    Pd = OrderedDict((k, v.to(dtype)) for k, v in P.items())
    rod = {k: (t(v).to(dtype) if v.dtype == np.float32 else t(v)) for k, v in ro.items()}
    torch.manual_seed(555)
    noise = torch.randn(T, N, 1).to(dtype)
    st = kfac_ref.new_state(Tf=2)
    # acktr_update, but keep the pieces: run it and read the state back (factors after the running average, the
    # eigen results it used, the final grads = v * nu)
    P0 = OrderedDict((k, v.clone()) for k, v in Pd.items())
    losses, info = kfac_ref.acktr_update(Pd, st, rod, 0.5, 0.01, noise=noise)
    # the raw loss gradient G: recompute from P0 (acktr_update does not return it)
    for v in P0.values():
        v.requires_grad_(True)
    obs = rod['obs'][:-1].reshape(T * N, 4, 84, 84)
    act = rod['actions'].reshape(T * N, -1)
    values, logp, ent, caps = kfac_ref.forward_capture(P0, obs, act)
    adv = rod['returns'][:-1] - values.view(T, N, 1)
    loss = adv.pow(2).mean() * 0.5 + (-(adv.detach() * logp.view(T, N, 1)).mean()) - ent * 0.01
    names = list(P0.keys())
    grads = dict(zip(names, torch.autograd.grad(loss, [P0[k] for k in names])))
    return P, ro, st, grads, info, kfac_ref.module_list(P)


def vec_err(vs, truth):
    num = sum(float(((a.double() - b.double()) ** 2).sum()) for a, b in zip(vs, truth))
    den = sum(float((b.double() ** 2).sum()) for b in truth)
    return (num / den) ** 0.5


def main():
    from a2c_ppo_acktr import _hip, algo
    from a2c_ppo_acktr.model import Policy
    torch.set_num_threads(16)
    P, ro, st64, g64, info64, mods = stage_inputs(torch.float64)
    _, _, st32, g32, info32, _ = stage_inputs(torch.float32)
    keys = [m[0] for m in mods]
    pname = {m[0]: m[2] for m in mods}
    gm = lambda g, k: g[pname[k]].reshape(g[pname[k]].shape[0], -1)

    def full_v(G, fa, fg, eig, gemm):
        """fp64 composition: G, factors (dicts), eig = fn(matrix) -> (d, Q) fp64 tensors, gemm = fn(Gm,dg,Qg,da,Qa)"""
        out = []
        for k in keys:
            da, Qa = eig(fa[k], k, 'a')
            dg, Qg = eig(fg[k], k, 'g')
            out.append(gemm(gm(G, k), dg, Qg, da, Qa))
        return out

    def eig64(M, k=None, side=None):
        d, Q = torch.linalg.eigh(M.double(), UPLO='U')
        return d * (d > 1e-6).double(), Q

    def eig32cpu(M, k=None, side=None):
        d, Q = torch.linalg.eigh(M.float(), UPLO='U')
        d = d * (d > 1e-6).float()
        return d.double(), Q.double()

    truth = full_v(g64, st64['m_aa'], st64['m_gg'], eig64, precond64)
    whole = sum(float((b ** 2).sum()) for b in truth) ** 0.5
    print('truth |v| = %.6g' % whole)

    # ------------------------------------------------------------------ HIP run of the same step
    class Discrete(object):
        n = 6
    pol = Policy((4, 84, 84), Discrete())
    pol.load_state_dict(OrderedDict((k, v.clone()) for k, v in P.items()), strict=True)
    pol.to('cuda')
    from a2c_ppo_acktr.storage import RolloutStorage
    T, N = ro['rewards'].shape[:2]
    rs = RolloutStorage(T, N, (4, 84, 84), Discrete(), 1)
    for k in ('obs', 'rewards', 'value_preds', 'returns', 'action_log_probs', 'actions', 'masks', 'bad_masks'):
        getattr(rs, k).copy_(t(ro[k]))
    rs.to('cuda')
    agent = algo.A2C_ACKTR(pol, value_loss_coef=0.5, entropy_coef=0.01, acktr=True)
    agent.use_graph = False
    opt = agent.optimizer
    opt.Tf = 2
    snap = {}
    inner = opt.step

    def step_hook():
        snap['g'] = pol.flat_grads.clone()
        return inner()
    opt.step = step_hook
    torch.manual_seed(555)
    agent.update(rs)
    opt.step = inner
    L = _hip.lib()
    stream = _hip.cur_stream(pol.flat_params.device)
    hip_mods = opt._mods
    assert len(hip_mods) == len(keys)
    names_hip = [n for n, _ in pol.named_parameters()]
    byparam = {id(p): n for n, p in pol.named_parameters()}
    for m, k in zip(hip_mods, keys):
        assert unsplit_name(byparam[id(m.param)]) == pname[k], (byparam[id(m.param)], pname[k])

    g_hip = {pname[k]: pol.grad_view(m.param, snap['g']).detach().cpu().double().reshape(g64[pname[k]].shape)
             for m, k in zip(hip_mods, keys)}
    fa_hip = {k: opt.m_aa[m.module].detach().cpu().double() for m, k in zip(hip_mods, keys)}
    fg_hip = {k: opt.m_gg[m.module].detach().cpu().double() for m, k in zip(hip_mods, keys)}
    v_hip = [pol.grad_view(m.param, opt.v_flat).detach().cpu().double().reshape(gm(g64, k).shape)
             for m, k in zip(hip_mods, keys)]

    def hip_eig_all(fa, fg):
        """run the HIP eigensolvers (batched Jacobi + rocSOLVER, whatever _eigendecompose dispatches) on given factors"""
        for m, k in zip(hip_mods, keys):
            opt.aa_run[m.oa:m.oa + m.na * m.na].copy_(fa[k].float().reshape(-1).cuda())
            opt.gg_run[m.og:m.og + m.ng * m.ng].copy_(fg[k].float().reshape(-1).cuda())
        import time
        for _ in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            opt._eigendecompose()
            torch.cuda.synchronize()
            print('   _eigendecompose: %.2f ms' % ((time.perf_counter() - t0) * 1e3), 'block sweeps',
                  list(getattr(opt, '_blk_sweeps', [])))
        res = {}
        for m, k in zip(hip_mods, keys):
            res[(k, 'a')] = (opt.d_a[m.module].cpu().double(), opt.Q_a[m.module].cpu().double().contiguous())
            res[(k, 'g')] = (opt.d_g[m.module].cpu().double(), opt.Q_g[m.module].cpu().double().contiguous())
        return res

    def hip_gemm(Gm, dg, Qg, da, Qa):
        OUT, IN = Gm.shape
        dev = lambda x: x.float().contiguous().cuda()
        qtg, qta, dgd, dad, Gd = dev(Qg.t()), dev(Qa.t()), dev(dg), dev(da), dev(Gm)
        t1, t2 = torch.empty(OUT * IN, device='cuda'), torch.empty(OUT * IN, device='cuda')
        v = torch.empty(OUT, IN, device='cuda')
        rc = L.acktr_kfac_precondition_f32(_hip.dev_ptr(Gd), _hip.dev_ptr(qtg), _hip.dev_ptr(dgd), _hip.dev_ptr(qta),
                                           _hip.dev_ptr(dad), OUT, IN, LA, _hip.dev_ptr(t1), _hip.dev_ptr(t2),
                                           _hip.dev_ptr(v), stream)
        _hip.check(rc, 'precondition')
        return v.cpu().double()

    def cpu32_gemm(Gm, dg, Qg, da, Qa):
        Gm, dg, Qg, da, Qa = (x.float() for x in (Gm, dg, Qg, da, Qa))
        v1 = Qg.t() @ Gm @ Qa
        v2 = v1 / (dg.unsqueeze(1) * da.unsqueeze(0) + LA)
        return (Qg @ v2 @ Qa.t()).double()

    r32 = lambda d: {k: v.float().double() for k, v in d.items()}
    fa64r, fg64r = r32(st64['m_aa']), r32(st64['m_gg'])          # truth factors rounded to fp32 (eigen-stage input)
    hip_eigs = hip_eig_all(fa64r, fg64r)
    base_r = full_v(g64, fa64r, fg64r, eig64, precond64)          # effect of merely rounding the factors to fp32
    rows = [
        ('end to end            HIP', v_hip),
        ('end to end            cpu fp32', [info32['grads'][pname[k]].double().reshape(gm(g64, k).shape) / info32['nu'] for k in keys]),
        ('factors rounded to fp32 only', base_r),
        ('grads                 HIP', full_v(g_hip, st64['m_aa'], st64['m_gg'], eig64, precond64)),
        ('grads                 cpu fp32', full_v({k: v.double() for k, v in g32.items()}, st64['m_aa'], st64['m_gg'], eig64, precond64)),
        ('factors               HIP', full_v(g64, fa_hip, fg_hip, eig64, precond64)),
        ('factors               cpu fp32', full_v(g64, {k: v.double() for k, v in st32['m_aa'].items()},
                                                   {k: v.double() for k, v in st32['m_gg'].items()}, eig64, precond64)),
        ('eigen (of fp32 truth) HIP', full_v(g64, fa64r, fg64r, lambda M, k, s: hip_eigs[(k, s)], precond64)),
        ('eigen (of fp32 truth) cpu fp32', full_v(g64, fa64r, fg64r, eig32cpu, precond64)),
        ('gemm                  HIP', full_v(g64, fa64r, fg64r, eig64, hip_gemm)),
        ('gemm                  cpu fp32', full_v(g64, fa64r, fg64r, eig64, cpu32_gemm)),
    ]
    print('%-34s %10s   per-module errors relative to |v| (modules with > 1e-4)' % ('stage replaced', 'rel err'))
    for name, vs in rows:
        e = vec_err(vs, truth)
        per = []
        for k, a, b in zip(keys, vs, truth):
            pe = float(((a.double() - b) ** 2).sum()) ** 0.5 / whole
            if pe > 1e-4:
                per.append('%s %.1e' % (k.replace('base.', ''), pe))
        print('%-34s %10.3e   %s' % (name, e, ', '.join(per)))
    # eigen-stage detail: residual / orthogonality of the HIP eigensolves per factor size
    print('eigen quality (HIP): n, ||QDQ^T - A||/||A||, ||Q^TQ - I||_max')
    for m, k in zip(hip_mods, keys):
        for side, F in (('a', fa64r), ('g', fg64r)):
            d, Q = hip_eigs[(k, side)]
            A = F[k]
            n = A.shape[0]
            if n < 32:
                continue
            d64, Q64 = torch.linalg.eigh(A, UPLO='U')
            res = float((Q @ torch.diag(d) @ Q.t() - A * 1.0).norm() / A.norm())
            res_t = float((Q64 @ torch.diag(d64 * (d64 > 1e-6)) @ Q64.t() - A).norm() / A.norm())
            print('   %-22s %s n=%4d  resid %.2e (fp64 thresholded: %.2e)  orth %.2e  max|d-d64| %.2e' % (
                k, side, n, res, res_t, float((Q.t() @ Q - torch.eye(n, dtype=torch.float64)).abs().max()),
                float((d - d64 * (d64 > 1e-6)).abs().max())))


if __name__ == '__main__':
    main()
