"""Rollout-side latency of Policy.act / get_value (model.py:54-70) on the device: python tools/act_bench.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'pytorch-a2c-ppo-acktr-gail_amd'))
import torch
import bench
from a2c_ppo_acktr.model import Policy
dev = torch.device('cuda:0')


def timeit(f, n=200):
    for _ in range(10):
        f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for name, shape, space, kw, N in (('NatureCNN Categorical(6)', (4, 84, 84), bench.Discrete(6), {}, 32),
                                  ('NatureCNN + GRU', (4, 84, 84), bench.Discrete(6), {'recurrent': True}, 32),
                                  ('NatureCNN Categorical(6)', (4, 84, 84), bench.Discrete(6), {}, 256),
                                  ('MLP(64) DiagGaussian(6)', (17,), bench.Box(6), {}, 32),
                                  ('MLP(64) DiagGaussian(6)', (17,), bench.Box(6), {}, 1024)):
    torch.manual_seed(1)
    pol = Policy(shape, space, base_kwargs=kw); pol.to(dev)
    obs = torch.rand(N, *shape, device=dev) * (255.0 if len(shape) == 3 else 1.0)
    hxs = torch.zeros(N, pol.recurrent_hidden_state_size, device=dev)
    masks = torch.ones(N, 1, device=dev)
    with torch.no_grad():
        a = timeit(lambda: pol.act(obs, hxs, masks))
        v = timeit(lambda: pol.get_value(obs, hxs, masks))
    print('%-28s N=%5d: act %7.1f us   get_value %7.1f us' % (name, N, a, v))
