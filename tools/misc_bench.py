"""Update latency of the less-travelled configurations (anything far off its neighbours is a path that fell off the
fast kernels): python tools/misc_bench.py"""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'pytorch-a2c-ppo-acktr-gail_amd'))
import torch
import bench
from a2c_ppo_acktr import algo
from a2c_ppo_acktr.model import Policy
from a2c_ppo_acktr.storage import RolloutStorage
dev = torch.device('cuda:0')


def storage(T, N, shape, space, H=1):
    g = torch.Generator(device=dev).manual_seed(5)
    rs = RolloutStorage(1, 1, (1,), space, 1)
    rs.num_steps = T
    img = len(shape) == 3
    rs.obs = torch.rand(T + 1, N, *shape, device=dev, generator=g) * (255.0 if img else 1.0)
    if space.__class__.__name__ == 'Discrete':
        rs.actions = torch.randint(0, space.n, (T, N, 1), device=dev, generator=g)
    else:
        rs.actions = torch.randn(T, N, space.shape[0], device=dev, generator=g)
    rs.rewards = torch.randn(T, N, 1, device=dev, generator=g)
    rs.value_preds = torch.randn(T + 1, N, 1, device=dev, generator=g)
    rs.action_log_probs = 0.1 * torch.randn(T, N, 1, device=dev, generator=g) - 2.0
    rs.masks = (torch.rand(T + 1, N, 1, device=dev, generator=g) < 0.98).float()
    rs.bad_masks = torch.ones(T + 1, N, 1, device=dev)
    rs.returns = torch.zeros(T + 1, N, 1, device=dev)
    rs.recurrent_hidden_states = 0.1 * torch.randn(T + 1, N, H, device=dev, generator=g)
    rs.compute_returns(torch.randn(N, 1, device=dev, generator=g), True, 0.99, 0.95, True)
    return rs


CNN, MLP = (4, 84, 84), (17,)
cases = [('A2C   CNN+GRU  Discrete 16x5', 'a2c', CNN, bench.Discrete(6), True, 5, 16),
         ('A2C   MLP      Box      16x5', 'a2c', MLP, bench.Box(6), False, 5, 16),
         ('ACKTR MLP      Box      32x20', 'acktr', MLP, bench.Box(6), False, 20, 32),
         ('PPO   MLP      Discrete 2048x1 (32 mb)', 'ppo32', MLP, bench.Discrete(6), False, 2048, 1),
         ('PPO   MLP+GRU  Box      128x32 (4 mb)', 'ppo', MLP, bench.Box(6), True, 128, 32),
         ('PPO   CNN      Box      128x32 (4 mb)', 'ppo', CNN, bench.Box(6), False, 128, 32)]
for name, kind, shape, space, rec, T, N in cases:
    torch.manual_seed(2)
    pol = Policy(shape, space, base_kwargs={'recurrent': rec}); pol.to(dev)
    rs = storage(T, N, shape, space, pol.recurrent_hidden_state_size)
    if kind == 'a2c':
        agent = algo.A2C_ACKTR(pol, 0.5, 0.01, lr=7e-4, eps=1e-5, alpha=0.99, max_grad_norm=0.5)
    elif kind == 'acktr':
        agent = algo.A2C_ACKTR(pol, 0.5, 0.01, acktr=True)
    else:
        agent = algo.PPO(pol, clip_param=0.2, ppo_epoch=4 if kind == 'ppo' else 10, num_mini_batch=4 if kind == 'ppo' else 32,
                         value_loss_coef=0.5, entropy_coef=0.01, lr=3e-4, eps=1e-5, max_grad_norm=0.5)
    bench._timed_updates(agent, rs, 3, 400)
    ms = bench._timed_updates(agent, rs, 12, 410)
    print('%-44s update ms median %8.3f  (min %.3f)' % (name, statistics.median(ms), min(ms)), flush=True)
