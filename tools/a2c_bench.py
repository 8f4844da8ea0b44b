"""A2C update (configs[1]: NatureCNN, 16 procs x 5 steps, B = 80) timing: python tools/a2c_bench.py"""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'pytorch-a2c-ppo-acktr-gail_amd'))
import torch
import bench
from a2c_ppo_acktr import algo
from a2c_ppo_acktr.model import Policy
dev = torch.device('cuda:0')
T, N = 5, 16
ro = bench.synth_rollouts(T, N, dev, seed=21)
torch.manual_seed(8)
pol = Policy((4, 84, 84), bench.Discrete(bench.ACTIONS)); pol.to(dev)
agent = algo.A2C_ACKTR(pol, 0.5, 0.01, lr=7e-4, eps=1e-5, alpha=0.99, max_grad_norm=0.5)
rs = bench.fill_storage(ro, dev)
rs.compute_returns(ro['next_value'], False, 0.99, 0.95, False)
bench._timed_updates(agent, rs, 3, 400)
ms = bench._timed_updates(agent, rs, 40, 410)
print('A2C update ms median %.3f min %.3f' % (statistics.median(ms), min(ms)))
