"""acktr_kfac_cov_dense_f32 on the ACKTR shapes (640 rows; D = 1568, 512), HIP-event times; ACKTR_COV_DENSE_TILE=1 = tile GEMM."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'pytorch-a2c-ppo-acktr-gail_amd')]
import torch
from a2c_ppo_acktr import _hip, ops
L = _hip.lib()
dev = torch.device('cuda:0')
for B, D in ((640, 1568), (640, 512), (640, 256), (80, 1568)):
    x = torch.randn(B, D, device=dev)
    out = torch.empty(D, D, device=dev)
    ws = ops.workspace(dev, L.acktr_kfac_cov_workspace_bytes(D, B), 'kfac')
    run = lambda: _hip.check(L.acktr_kfac_cov_dense_f32(_hip.dev_ptr(x), D, B, D, 1.0 / B, _hip.dev_ptr(out), _hip.dev_ptr(ws),
                                                        ws.numel(), _hip.cur_stream(dev)), 'cov')
    for _ in range(5):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        run()
    e1.record(); torch.cuda.synchronize()
    ref = (x.double().t() @ x.double() / B)
    err = float((out.double() - ref).norm() / ref.norm())
    print('B %4d D %4d  %.1f us  %.1f TFLOP/s (half count)  err %.1e  symmetric %s' % (B, D, e0.elapsed_time(e1) * 10, B * D * (D + 1.0) / (e0.elapsed_time(e1) * 1e-5) / 1e12, err, bool(torch.equal(out, out.t()))))
