"""Recurrent PPO update (NatureCNN + GRU 512, T=128 x 32 procs, 4 epochs x 4 minibatches of 8 whole trajectories) next to
the feed-forward update of the same size: python tools/rec_bench.py [--trace]"""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'pytorch-a2c-ppo-acktr-gail_amd'))
import torch
import bench
from a2c_ppo_acktr import algo
from a2c_ppo_acktr.model import Policy
dev = torch.device('cuda:0')
T, N = bench.T_STEPS, bench.PROCS_PER_GPU
ro = bench.synth_rollouts(T, N, dev, seed=31)
for rec in (False, True):
    torch.manual_seed(7)
    pol = Policy((4, 84, 84), bench.Discrete(bench.ACTIONS), base_kwargs={'recurrent': rec}); pol.to(dev)
    agent = algo.PPO(pol, **bench.PPO_HP)
    rs = bench.fill_storage(ro, dev)
    if rec:
        rs.recurrent_hidden_states = 0.1 * torch.randn(T + 1, N, 512, device=dev)
    rs.compute_returns(ro['next_value'], True, 0.99, 0.95, True)
    bench._timed_updates(agent, rs, 3, 400)
    ms = bench._timed_updates(agent, rs, 10, 410)
    print('%s PPO update: ms median %.3f (%.0f samples/s)' % ('recurrent  ' if rec else 'feedforward', statistics.median(ms),
          T * N * bench.PPO_HP['ppo_epoch'] / (statistics.median(ms) * 1e-3)))
