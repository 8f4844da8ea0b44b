"""Quick numerical check of the conv ops against torch (fp64 reference) at one batch size."""
import os, sys
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'pytorch-a2c-ppo-acktr-gail_amd'))
from a2c_ppo_acktr import ops

def rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm()).item()

for B in [int(v) for v in (sys.argv[1:] or ['80'])]:
    g = torch.Generator().manual_seed(B)
    rn = lambda *s: torch.randn(*s, generator=g).cuda()
    for (C, H, OC, K, S) in [(4, 84, 32, 8, 4), (32, 20, 64, 4, 2), (64, 9, 32, 3, 1)]:
        x = torch.relu(rn(B, C, H, H)); w = rn(OC, C, K, K) * 0.05; b = rn(OC) * 0.1
        y = ops.conv2d_fwd(x, w, b, S, relu=True)
        yr = torch.relu(F.conv2d(x.double(), w.double(), b.double(), stride=S))
        dy = rn(*y.shape)
        dx = ops.conv2d_dgrad(dy, w, x, S)
        dxr = F.conv_transpose2d(dy.double(), w.double(), stride=S) * (x > 0)
        bad = ((dx.double() - dxr).abs() > 1e-3 * dxr.abs().max()).nonzero()
        print('B=%d k%ds%d fwd %.2e dgrad %.2e  bad=%d %s' % (B, K, S, rel(y, yr), rel(dx, dxr), bad.shape[0],
              bad[:6].tolist() if bad.shape[0] else ''), flush=True)
