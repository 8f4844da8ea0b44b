"""CPU prototype (numpy fp32) of the blocked one-sided Jacobi eigensolver that csrc/jacobi_block.hip implements:
convergence (sweeps), residual / orthogonality, and -- on the saved C4 factors (/tmp/kfac_c4_step0.pt written by
tools/kfac_stage_err.py's stage_inputs) -- the error of the KFAC update against the fp64 truth.

Algorithm: W = A (columns), V = I.  Column blocks of width b; round-robin pairing of the nb blocks; for a pair
(I, J): P = [W_I W_J] (n x 2b), G = P^T P, G = U L U^T (small symmetric eigenproblem), P <- P U, V_IJ <- V_IJ U.
At convergence the columns of W are orthogonal: w_i = lambda_i v_i, lambda_i = v_i . w_i.
"""
import sys
import time

import numpy as np

f32 = np.float32


_PAIRS = {}
INNER = int(__import__('os').environ.get('INNER', '8'))
RTOL = float(__import__('os').environ.get('RTOL', '1e-7'))


def _pairs(m):
    if m not in _PAIRS:
        rounds = []
        for r in range(m - 1):
            pq = [(m - 1, r)] + [((r + t) % (m - 1), (r - t + m - 1) % (m - 1)) for t in range(1, m // 2)]
            pq = [(min(a, b), max(a, b)) for a, b in pq]
            rounds.append((np.array([a for a, b in pq]), np.array([b for a, b in pq])))
        _PAIRS[m] = rounds
    return _PAIRS[m]


def small_eig_jacobi(G, sweeps=8, rtol=1e-7, floor2=0.0):
    """parallel-order cyclic two-sided Jacobi on a small symmetric PSD matrix in fp32 (what one workgroup does in
    LDS); rotation (p,q) only when |g_pq| > rtol*sqrt(g_pp g_qq) + floor2 (relative criterion -> small columns keep
    their relative accuracy).  Returns (U, number of rotations applied) with columns by descending eigenvalue."""
    G = G.astype(f32).copy()
    m = G.shape[0]
    U = np.eye(m, dtype=f32)
    nrot = 0
    for _ in range(sweeps):
        rot_this = 0
        for p, q in _pairs(m):
            apq = G[p, q]
            app, aqq = G[p, p], G[q, q]
            act = np.abs(apq) > f32(rtol) * np.sqrt(np.abs(app * aqq)) + f32(floor2)
            if not act.any():
                continue
            rot_this += int(act.sum())
            apq_s = np.where(act, apq, f32(1))
            tau = (aqq - app) / (f32(2) * apq_s)
            t = np.where(tau >= 0, f32(1), f32(-1)) / (np.abs(tau) + np.sqrt(f32(1) + tau * tau))
            c = (f32(1) / np.sqrt(f32(1) + t * t)).astype(f32)
            sn = (t * c).astype(f32)
            c = np.where(act, c, f32(1)).astype(f32)
            sn = np.where(act, sn, f32(0)).astype(f32)
            J = np.zeros((m, m), f32)
            J[p, p] = c; J[q, q] = c; J[p, q] = sn; J[q, p] = -sn
            G = (J.T @ G @ J).astype(f32)
            U = (U @ J).astype(f32)
        nrot += rot_this
        if rot_this == 0:
            break
    order = np.argsort(-np.diag(G), kind='stable')
    return U[:, order], nrot


def block_jacobi(A, b=16, max_sweeps=int(__import__('os').environ.get('SWEEPS', '12')), tol=1e-6, small='lapack', verbose=False, Q0=None, w64=False):
    A = np.asarray(A, f32)
    n = A.shape[0]
    nb = -(-n // b)
    if nb % 2:
        nb += 1
    npad = nb * b
    W = np.zeros((n, npad), f32)
    W[:, :n] = np.triu(A) + np.triu(A, 1).T
    V = np.zeros((n, npad), f32)
    V[:n, :n] = np.eye(n, dtype=f32)
    if Q0 is not None:                       # refinement of a given (fp32) eigenbasis: W = A Q0, V = Q0
        Q0 = Q0[:, ::-1]                     # descending eigenvalues: big columns first
        V[:n, :n] = Q0
        As = np.triu(A) + np.triu(A, 1).T
        W[:, :n] = (As.astype(np.float64) @ Q0.astype(np.float64)).astype(f32) if w64 else (As @ Q0).astype(f32)
    players = list(range(nb))
    sweeps_done = 0
    nu2 = f32((float(__import__('os').environ.get('NU', '1e-7'))) ** 2 * (W.astype(np.float64) ** 2).sum(0).max())       # (1e-6 * largest column norm)^2
    for sweep in range(max_sweeps):
        worst = 0.0
        nrot = 0
        for rnd in range(nb - 1):
            for k in range(nb // 2):
                I, J = players[k], players[nb - 1 - k]
                if I > J:
                    I, J = J, I
                cols = np.r_[I * b:(I + 1) * b, J * b:(J + 1) * b]
                P = W[:, cols]
                G = (P.T @ P).astype(f32)
                dg = np.sqrt(np.maximum(np.diag(G), f32(1e-37)))
                offn = np.abs(G / np.outer(dg, dg))
                np.fill_diagonal(offn, 0)
                # only columns that are not numerically zero count for convergence
                sc = np.abs(G) / (np.outer(dg, dg) + nu2)
                np.fill_diagonal(sc, 0)
                worst = max(worst, float(sc.max()))
                if small == 'lapack':
                    lam, U = np.linalg.eigh(G.astype(f32))
                    U = U[:, ::-1].astype(f32)
                else:
                    U, nr = small_eig_jacobi(G, sweeps=INNER, rtol=RTOL, floor2=nu2)
                    nrot += nr
                W[:, cols] = (P @ U).astype(f32)
                V[:, cols] = (V[:, cols] @ U).astype(f32)
            players = [players[0]] + [players[-1]] + players[1:-1]
        sweeps_done = sweep + 1
        if verbose:
            print('   sweep %d worst normalised off-diagonal %.2e rotations %d' % (sweep, worst, nrot))
        if worst < tol or (small != 'lapack' and nrot == 0):
            break
    lam = np.einsum('ij,ij->j', V.astype(np.float64), W.astype(np.float64))
    nrm = np.sqrt(np.einsum('ij,ij->j', V.astype(np.float64), V.astype(np.float64)))
    keep = nrm > 0.5                                   # padded (zero) columns have V = 0
    lam, Vk = (lam[keep] / nrm[keep] ** 2), (V[:, keep] / nrm[keep]).astype(f32)
    order = np.argsort(lam, kind='stable')
    return lam[order].astype(f32), Vk[:, order], sweeps_done


def quality(A, d, Q):
    A = np.asarray(A, np.float64)
    A = np.triu(A) + np.triu(A, 1).T
    Q = Q.astype(np.float64)
    d = d.astype(np.float64)
    n = A.shape[0]
    return (np.linalg.norm(Q @ np.diag(d) @ Q.T - A) / np.linalg.norm(A), np.abs(Q.T @ Q - np.eye(n)).max(),
            np.abs(d - np.linalg.eigvalsh(A)).max())


if __name__ == '__main__':
    import torch
    blob = torch.load('/tmp/kfac_c4_step0.pt')
    sys.path.insert(0, 'tools')
    keys = [x[0] for x in blob['mods']]
    pname = {x[0]: x[2] for x in blob['mods']}
    fa = {k: v.float().double() for k, v in blob['m_aa'].items()}
    fg = {k: v.float().double() for k, v in blob['m_gg'].items()}
    g64 = blob['g']
    gm = lambda k: g64[pname[k]].reshape(g64[pname[k]].shape[0], -1)
    LA = 1e-2

    def precond(Gm, dg, Qg, da, Qa):
        v1 = Qg.T @ Gm @ Qa
        return Qg @ (v1 / (dg[:, None] * da[None, :] + LA)) @ Qa.T

    def e64(M, tag=None):
        d, Q = np.linalg.eigh(np.asarray(M, np.float64))
        return d * (d > 1e-6), Q

    def e32(M, tag=None):
        d, Q = torch.linalg.eigh(torch.from_numpy(np.asarray(M, np.float32)), UPLO='U')
        d, Q = d.numpy(), Q.numpy()
        return (d * (d > 1e-6)).astype(np.float64), Q.astype(np.float64)

    b = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    small = sys.argv[2] if len(sys.argv) > 2 else 'lapack'
    cache = {}
    MODE = __import__('os').environ.get('MODE', 'plain')

    def ebj(M, tag):
        key = tag if not tag.endswith('#b/a') else 'bias/a'
        if key not in cache:
            if M.shape[0] < 200:
                return e32(M)
            else:
                t0 = time.time()
                Q0 = None
                if MODE != 'plain':
                    Q0 = torch.linalg.eigh(torch.from_numpy(np.asarray(M, np.float32)), UPLO='U')[1].numpy()
                d, Q, sw = block_jacobi(np.asarray(M, np.float32), b=b, small=small, verbose=True, Q0=Q0, w64=(MODE == 'refine64'))
                print('  n=%d: %d sweeps, %.1f s; resid %.2e orth %.2e max|d-d64| %.2e' % ((M.shape[0], sw, time.time() - t0) + quality(M, d, Q)))
                cache[key] = ((d * (d > 1e-6)).astype(np.float64), Q.astype(np.float64))
        return cache[key]
    only = sys.argv[3].split(',') if len(sys.argv) > 3 else None
    if only:
        keys = [k for k in keys if any(o in k for o in only)]
    truth = {k: precond(gm(k).numpy(), *e64(fg[k].numpy()), *e64(fa[k].numpy())) for k in keys}
    whole = np.sqrt(sum((v ** 2).sum() for v in truth.values()))
    for name, fn in (('lapack fp32', e32), ('block jacobi fp32', ebj)):
        errs = {}
        for k in keys:
            A_, G_ = fa[k].numpy(), fg[k].numpy()
            v = precond(gm(k).numpy(), *fn(G_, k + '/g'), *fn(A_, k + '/a'))
            errs[k] = np.sqrt(((v - truth[k]) ** 2).sum()) / whole
        print(name, 'whole %.3e' % np.sqrt(sum(e ** 2 for e in errs.values())), {k[5:]: '%.1e' % e for k, e in errs.items() if e > 2e-5})
