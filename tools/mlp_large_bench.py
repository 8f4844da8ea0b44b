"""PPO.update on the HalfCheetah-shaped MLP policy at configs[4]'s rollout size (obs 17, act 6, T=2048 x 1024 procs,
10 epochs x 32 minibatches of 65536 rows): python tools/mlp_large_bench.py"""
import os, sys, statistics, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'pytorch-a2c-ppo-acktr-gail_amd'))
import torch
import bench
from a2c_ppo_acktr import algo
from a2c_ppo_acktr.model import Policy
from a2c_ppo_acktr.storage import RolloutStorage
dev = torch.device('cuda:0')
T, N, OD, A = 2048, 1024, 17, 6
g = torch.Generator(device=dev).manual_seed(5)
torch.manual_seed(3)
pol = Policy((OD,), bench.Box(A)); pol.to(dev)
agent = algo.PPO(pol, clip_param=0.2, ppo_epoch=10, num_mini_batch=32, value_loss_coef=0.5, entropy_coef=0.0, lr=3e-4,
                 eps=1e-5, max_grad_norm=0.5)
rs = RolloutStorage(1, 1, (1,), bench.Box(A), 1)
rs.num_steps = T
rs.obs = torch.randn(T + 1, N, OD, device=dev, generator=g)
rs.actions = torch.randn(T, N, A, device=dev, generator=g)
rs.rewards = torch.randn(T, N, 1, device=dev, generator=g)
rs.value_preds = torch.randn(T + 1, N, 1, device=dev, generator=g)
rs.action_log_probs = 0.1 * torch.randn(T, N, 1, device=dev, generator=g) - 5.0
rs.masks = (torch.rand(T + 1, N, 1, device=dev, generator=g) < 0.99).float()
rs.bad_masks = torch.ones(T + 1, N, 1, device=dev)
rs.returns = torch.zeros(T + 1, N, 1, device=dev)
rs.recurrent_hidden_states = torch.zeros(T + 1, N, 1, device=dev)
torch.cuda.synchronize(); t0 = time.perf_counter()
rs.compute_returns(torch.randn(N, 1, device=dev, generator=g), True, 0.99, 0.95, True)
torch.cuda.synchronize()
print('compute_returns (first call): %.3f ms' % ((time.perf_counter() - t0) * 1e3))
ms = bench._timed_updates(agent, rs, 2, 400)
ms = bench._timed_updates(agent, rs, 3, 410)
m = statistics.median(ms)
print('PPO.update MLP 2048 x 1024: ms median %.1f (%.1f M samples/s, %.0f us per 65536-row minibatch step)'
      % (m, T * N * 10 / m / 1e3, m * 1e3 / 320))
