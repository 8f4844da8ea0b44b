"""Time the blocked Jacobi (csrc/jacobi_block.hip) on NatureCNN-like factor sets and on the saved C4 factors if present."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'pytorch-a2c-ppo-acktr-gail_amd')):
    sys.path.insert(0, p)
import torch
from a2c_ppo_acktr import _hip


def run(mats, label, reps=3):
    L = _hip.lib()
    ns = [m.shape[0] for m in mats]
    ds = [torch.empty(n, device='cuda') for n in ns]
    qts = [torch.empty(n, n, device='cuda') for n in ns]
    jobs = (_hip.JacobiJob * len(ns))(*[_hip.JacobiJob(m.data_ptr(), d.data_ptr(), q.data_ptr(), n)
                                         for m, d, q, n in zip(mats, ds, qts, ns)])
    cns = (ctypes.c_int * len(ns))(*ns)
    ws = torch.empty(L.acktr_sym_eig_jacobi_block_workspace_bytes(cns, len(ns)), dtype=torch.uint8, device='cuda')
    sweeps = (ctypes.c_int * len(ns))()
    for r in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rc = L.acktr_sym_eig_jacobi_block_f32(jobs, len(ns), 1e-6, _hip.dev_ptr(ws), ws.numel(), 0, sweeps, _hip.cur_stream())
        _hip.check(rc, 'jb')
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
        print('%s: %.2f ms, sweeps %s' % (label, dt, list(sweeps)), flush=True)
    for m, d, q in zip(mats, ds, qts):
        A = m.double(); Q = q.double().t(); dd = d.double()
        print('   n=%4d resid %.2e orth %.2e' % (m.shape[0], float((Q @ torch.diag(dd) @ Q.t() - A).norm() / A.norm()),
                                                 float((Q.t() @ Q - torch.eye(m.shape[0], device='cuda', dtype=torch.float64)).abs().max())))


def psd(n, rank, g):
    a = torch.randn(rank, n, generator=g)
    return (a.t() @ a / rank).contiguous().cuda()


g = torch.Generator().manual_seed(0)
sizes = [256, 512, 576, 512, 512, 512, 512]
run([psd(n, 640, g) for n in sizes], 'NatureCNN factors <= 576 (random PSD, rank 640)')
run([psd(1568, 640, g)], '1568 alone (rank 640)')
run([psd(n, 640, g) for n in sizes + [1568]], 'all eight')
