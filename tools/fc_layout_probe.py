"""Does the FC dgrad want a k-contiguous weight operand?  Times dX = dY W as a plain GEMM with W row-contiguous
(what the layer has) and with a pre-transposed W^T (k-contiguous, one ds_read_b128 per fragment)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'pytorch-a2c-ppo-acktr-gail_amd'))
import torch
from a2c_ppo_acktr import ops
dev = torch.device('cuda:0')
B = 1024
g = torch.Generator().manual_seed(1)
dy = torch.randn(B, 512, generator=g).to(dev)
w = (torch.randn(512, 1568, generator=g) * 0.02).to(dev)
wt = w.t().contiguous()
x = torch.randn(B, 1568, generator=g).to(dev)
out = torch.empty(B, 1568, device=dev)
dw = torch.empty(512, 1568, device=dev)
dyt, xt = dy.t().contiguous(), x.t().contiguous()
cases = {
    'dgrad W row-contig (RCT)': lambda: ops.gemm(dy, w, out=out),
    'dgrad W^T k-contig (KCT)': lambda: ops.gemm(dy, wt, out=out, trans_b=True),
    'wgrad dY,X row-contig (RCT,RCT)': lambda: ops.gemm(dy, x, out=dw, trans_a=True),
    'wgrad dY^T,X^T k-contig (KCT,KCT)': lambda: ops.gemm(dyt, xt, out=dw, trans_b=True),
}
ref = dy @ w
for name, fn in cases.items():
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print('%-36s %7.2f us' % (name, e0.elapsed_time(e1) * 10), flush=True)
ops.gemm(dy, wt, out=out, trans_b=True)
print('check', float((out - ref).norm() / ref.norm()))
