"""compute_returns timing at the C5 stress size and a wide rollout: python tools/scan_bench.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'pytorch-a2c-ppo-acktr-gail_amd'))
import torch
from bench import Discrete
from a2c_ppo_acktr.storage import RolloutStorage
dev = torch.device('cuda:0')
for T, N in ((2048, 1024), (512, 65536), (2048, 1), (128, 256)):
    for mode in ((True, True), (False, False)):
        g = torch.Generator(device=dev).manual_seed(23)
        rs = RolloutStorage(1, 1, (1,), Discrete(6), 1)
        rs.num_steps = T
        rs.rewards = torch.randn(T, N, 1, device=dev, generator=g)
        rs.value_preds = torch.randn(T + 1, N, 1, device=dev, generator=g)
        rs.returns = torch.zeros(T + 1, N, 1, device=dev)
        rs.masks = (torch.rand(T + 1, N, 1, device=dev, generator=g) < 0.99).float()
        rs.bad_masks = (torch.rand(T + 1, N, 1, device=dev, generator=g) < 0.995).float()
        nv = torch.randn(N, 1, device=dev, generator=g)
        for _ in range(5):
            rs.compute_returns(nv, mode[0], 0.99, 0.95, mode[1])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            rs.compute_returns(nv, mode[0], 0.99, 0.95, mode[1])
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 50
        bpe = 20 if mode == (True, True) else 12
        print('T=%5d N=%6d gae=%d proper=%d: %.2f us  %.0f GB/s  %.1f ns/step' % (T, N, mode[0], mode[1], ms * 1e3, T * N * bpe / ms / 1e6, ms * 1e6 / T))
