"""PPO update on the Hopper-shaped MLP (configs[0]) alone, for rocprofv3: python tools/mlp_bench.py [updates]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'pytorch-a2c-ppo-acktr-gail_amd'))
import torch
from bench import Box, _timed_updates
from a2c_ppo_acktr import algo
from a2c_ppo_acktr.model import Policy
from a2c_ppo_acktr.storage import RolloutStorage
dev = torch.device('cuda:0')
T, N = 2048, 1
g = torch.Generator(device=dev).manual_seed(24)
torch.manual_seed(10)
pol = Policy((11,), Box(3)); pol.to(dev)
agent = algo.PPO(pol, 0.2, 10, 32, 0.5, 0.0, lr=3e-4, eps=1e-5, max_grad_norm=0.5)
rs = RolloutStorage(1, 1, (1,), Box(3), 1)
rs.num_steps = T
rs.obs = torch.randn(T + 1, N, 11, device=dev, generator=g)
rs.actions = torch.randn(T, N, 3, device=dev, generator=g)
rs.rewards = torch.randn(T, N, 1, device=dev, generator=g)
rs.value_preds = torch.randn(T + 1, N, 1, device=dev, generator=g)
rs.action_log_probs = 0.1 * torch.randn(T, N, 1, device=dev, generator=g) - 2
rs.masks = (torch.rand(T + 1, N, 1, device=dev, generator=g) < 0.99).float()
rs.bad_masks = (torch.rand(T + 1, N, 1, device=dev, generator=g) < 0.995).float()
rs.returns = torch.zeros(T + 1, N, 1, device=dev)
rs.recurrent_hidden_states = torch.zeros(T + 1, N, 1, device=dev)
rs.compute_returns(torch.randn(N, 1, device=dev, generator=g), True, 0.99, 0.95, True)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
_timed_updates(agent, rs, 2, 600)
ms = _timed_updates(agent, rs, n, 610)
print('ms per update', ['%.2f' % m for m in ms], 'us per minibatch step %.1f' % (sorted(ms)[len(ms) // 2] * 1e3 / 320))
