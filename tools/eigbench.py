"""Time the symmetric eigensolvers of the KFAC step at the factor sizes of the NatureCNN policy."""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'pytorch-a2c-ppo-acktr-gail_amd'))
from a2c_ppo_acktr import _hip
L = _hip.lib()
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
for n in (257, 513, 577, 1569):
    x = torch.randn(n, 2 * n, generator=g)
    a = (x @ x.t() / (2 * n)).to(dev)
    q = a.clone(); d = torch.empty(n, device=dev); e = torch.empty(n, device=dev); info = torch.zeros(1, dtype=torch.int32, device=dev)
    for it in range(3):
        q.copy_(a)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = L.acktr_sym_eig_rocsolver_f32(_hip.dev_ptr(q), n, _hip.dev_ptr(d), _hip.dev_ptr(e), _hip.dev_ptr(info, torch.int32), 1e-6, _hip.cur_stream(dev))
        e1.record(); torch.cuda.synchronize()
        assert rc == 0
    print('rocsolver ssyevd n=%4d  %.2f ms' % (n, e0.elapsed_time(e1)), flush=True)
    # torch.linalg.eigh for comparison (also rocSOLVER/hipSOLVER underneath)
    torch.cuda.synchronize(); e0.record(); torch.linalg.eigh(a); e1.record(); torch.cuda.synchronize()
    e0.record(); torch.linalg.eigh(a); e1.record(); torch.cuda.synchronize()
    print('   torch.linalg.eigh      %.2f ms' % e0.elapsed_time(e1), flush=True)
