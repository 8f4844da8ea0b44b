"""In-kernel clock and matrix-pipe occupancy of the conv1 weight-gradient ring kernel (diagnostic build):
    bash tools/build_variant.sh s9stamp "-DS9_STAMP" && ACKTR_HIP_LIB=tools/bin/libacktr_s9stamp.so python tools/probe/s9_clock.py
"""
import ctypes, os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'pytorch-a2c-ppo-acktr-gail_amd')]
import torch
from a2c_ppo_acktr import ops, _hip
B = 1024
g = torch.Generator().manual_seed(1)
x = torch.randint(0, 256, (B, 4, 84, 84), generator=g).float().cuda()
dy = (torch.randn(B, 32, 20, 20, generator=g) * (torch.rand(B, 32, 20, 20, generator=g) < 0.6).float()).cuda()
flat = torch.zeros(32 * 256 + 32, device='cuda')
dw, db = flat[:32 * 256].view(32, 4, 8, 8), flat[32 * 256:]
for mode in ('cold', 'hot'):
    n = 1 if mode == 'cold' else 3000
    for _ in range(n):
        ops.conv2d_wgrad(x, dy, (32, 4, 8, 8), 4, dw, db, in_scale=1 / 255.0)
    torch.cuda.synchronize()
    buf = (ctypes.c_longlong * (6 * 4096))()
    rc = _hip.lib().acktr_debug_s9_stamps(buf)
    assert rc == 0
    rows = [tuple(buf[6 * w + j] for j in range(6)) for w in range(2048) if buf[6 * w + 2] > 0]
    clk = [r[0] / r[1] * 0.1 for r in rows]
    busy = [r[2] * 80 * 64 * 2 / r[0] for r in rows]          # two waves share a SIMD's matrix pipe
    us = [r[1] / 100.0 for r in rows]
    t0 = min(r[3] for r in rows)
    print('%-4s waves %d  clock median %.3f GHz (min %.3f max %.3f)  loop %.1f us median (max %.1f)  matrix-pipe busy %.3f median'
          % (mode, len(rows), statistics.median(clk), min(clk), max(clk), statistics.median(us), max(us), statistics.median(busy)))
    print('     (us after the first wave entered)  last entry %.1f  loop start median %.1f  loop end median %.1f max %.1f  exit median %.1f max %.1f'
          % ((max(r[3] for r in rows) - t0) / 100, statistics.median((r[4] - r[1] - t0) for r in rows) / 100,
             statistics.median((r[4] - t0) for r in rows) / 100, (max(r[4] for r in rows) - t0) / 100,
             statistics.median((r[5] - t0) for r in rows) / 100, (max(r[5] for r in rows) - t0) / 100))
