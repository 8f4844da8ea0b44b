"""Golden vectors for the ACKTR / KFAC path from the REFERENCE (build container only); see make_golden.py."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def main(fast=False, threads=None):
    """fast=False: kfac.npz (cases 0, 1).  fast=True: kfac_fast.npz, case 2 = the NatureCNN case with the optimizer's
    public `fast_cnn` attribute (kfac.py:98, read by the hooks at kfac.py:150,168) set to True.
    threads=K: kfac_t<K>.npz -- cases 0, 1 once more with torch.set_num_threads(K) instead of main.py:38's 1 (other
    intra-op summation orders in the reference's own GEMMs / eigh): only the preconditioned update vectors, the
    post-update parameters and the A2C-path RMSprop parameters are kept.  The distance between kfac.npz and kfac_t8.npz
    is the reference's SELF-distance that tests/test_oracle_kfac.py records and tests/test_kfac_gpu.py uses as its bar."""
    import make_golden as mg          # installs the reference import harness
    if threads:
        torch.set_num_threads(threads)
    import synth
    import ref_harness
    from a2c_ppo_acktr import algo    # reference
    out = {}
    cases = [('cnn', 'cnn', None, 6, 20, 32), ('mlp', 'mlp', 17, 6, 5, 16), ('cnn_fast', 'cnn', None, 6, 20, 32)]
    for ci, (tag, kind, od, A, T, N) in enumerate(cases):
        if fast != tag.endswith('_fast'):
            continue
        pol, P = mg.make_policy(kind, 90 + ci, od, A)
        obs_shape = (4, 84, 84) if kind == 'cnn' else (od,)
        space = ref_harness.Discrete(A) if kind == 'cnn' else ref_harness.Box(A)
        ro = synth.make_rollouts(T, N, obs_shape, 'discrete' if kind == 'cnn' else 'box', A, salt=820 + ci)
        rs = mg.fill_storage(ro, obs_shape, space)
        rs.compute_returns(mg.t(ro['next_value']), False, 0.99, 0.95, False)
        agent = algo.A2C_ACKTR(pol, value_loss_coef=0.5, entropy_coef=0.01, acktr=True)
        agent.optimizer.Tf = 2            # public attribute (kfac.py:137): eigendecompose at steps 0 and 2
        if fast:
            agent.optimizer.fast_cnn = True
        names = [n for n, _ in pol.named_parameters()]
        out[tag + '/param_names'] = np.array(names)
        for it in range(3):
            torch.manual_seed(555 + it)
            res = agent.update(rs)
            out[tag + '/update%d/losses' % it] = np.array(res, dtype=np.float64)
            for n, p in pol.named_parameters():
                mg.summarize(tag + '/update%d/param/%s' % (it, n), p.detach().numpy(), out)
                mg.summarize(tag + '/update%d/kfgrad/%s' % (it, n), p.grad.detach().numpy(), out)
            for i, m in enumerate(agent.optimizer.modules if not threads else []):
                for fam, d in (('m_aa', agent.optimizer.m_aa), ('m_gg', agent.optimizer.m_gg)):
                    mat = d[m].detach().numpy()
                    mg.summarize(tag + '/update%d/%s/%d' % (it, fam, i), mat, out)
                    out[tag + '/update%d/%s/%d#trace' % (it, fam, i)] = np.float64(np.trace(mat.astype(np.float64)))
        out[tag + '/module_classes'] = np.array([m.__class__.__name__ for m in agent.optimizer.modules])
    if threads:
        # the A2C (RMSprop) case of make_golden.g_a2c at the same thread count: parameters after each update
        hp = dict(value_loss_coef=0.5, entropy_coef=0.01, lr=7e-4, eps=1e-5, alpha=0.99, max_grad_norm=0.5)     # = g_a2c
        for ci, (tag, kind, od, A, T, N) in enumerate([('cnn', 'cnn', None, 6, 5, 16), ('mlp', 'mlp', 17, 6, 5, 16)]):
            pol, P = mg.make_policy(kind, 80 + ci, od, A)
            obs_shape = (4, 84, 84) if kind == 'cnn' else (od,)
            space = ref_harness.Discrete(A) if kind == 'cnn' else ref_harness.Box(A)
            ro = synth.make_rollouts(T, N, obs_shape, 'discrete' if kind == 'cnn' else 'box', A, salt=800 + ci)
            rs = mg.fill_storage(ro, obs_shape, space)
            rs.compute_returns(mg.t(ro['next_value']), False, 0.99, 0.95, False)
            agent = algo.A2C_ACKTR(pol, acktr=False, **hp)
            for it in range(3):
                res = agent.update(rs)
                out['a2c/' + tag + '/update%d/losses' % it] = np.array(res, dtype=np.float64)
                for n, p in pol.named_parameters():
                    mg.summarize('a2c/' + tag + '/update%d/param/%s' % (it, n), p.detach().numpy(), out)
        out['threads'] = np.int64(torch.get_num_threads())
    name = 'kfac_t%d.npz' % threads if threads else ('kfac_fast.npz' if fast else 'kfac.npz')
    np.savez_compressed(os.path.join(HERE, name), **out)


if __name__ == '__main__':
    sys.path.insert(0, HERE)
    thr = int(sys.argv[sys.argv.index('--threads') + 1]) if '--threads' in sys.argv else None
    main(fast='--fast' in sys.argv, threads=thr)
