"""One-rank rehearsal of the data-parallel A2C / ACKTR updates on RCCL: plain single process, data-parallel path replayed
as graph pieces around the collectives, data-parallel path launch by launch.   python tools/dp_rehearsal.py"""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'pytorch-a2c-ppo-acktr-gail_amd'))
import torch
import torch.distributed as dist
import bench
from a2c_ppo_acktr import algo
from a2c_ppo_acktr.model import Policy
from a2c_ppo_acktr.parallel import DataParallel

os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29541')
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
dev = torch.device('cuda:0')
torch.cuda.set_device(dev)
sys.stdout.flush(); fd = os.dup(1); os.dup2(2, 1)          # RCCL prints its banner on fd 1
dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)


def say(s):
    os.write(fd, (s + '\n').encode())


def run(acktr, mode):
    T, N = (20, 32) if acktr else (5, 16)
    ro = bench.synth_rollouts(T, N, dev, seed=22 if acktr else 21)
    torch.manual_seed(9)
    pol = Policy((4, 84, 84), bench.Discrete(bench.ACTIONS)); pol.to(dev)
    if acktr:
        agent = algo.A2C_ACKTR(pol, 0.5, 0.01, acktr=True)
    else:
        agent = algo.A2C_ACKTR(pol, 0.5, 0.01, lr=7e-4, eps=1e-5, alpha=0.99, max_grad_norm=0.5)
    if mode != 'plain':
        agent.dp = DataParallel(force_active=True)
        if acktr:
            agent.optimizer.dp = agent.dp
    agent.use_graph = mode != 'dp_eager'
    rs = bench.fill_storage(ro, dev)
    rs.compute_returns(ro['next_value'], False, 0.99, 0.95, False)
    bench._timed_updates(agent, rs, 4, 500)
    ms = bench._timed_updates(agent, rs, 36, 510)
    return statistics.median(ms)


for acktr in (False, True):
    r = {m: run(acktr, m) for m in ('plain', 'dp_graph_pieces', 'dp_eager')}
    say('%s update, ms median (ACKTR: updates without eigendecomposition dominate the median): %s'
        % ('ACKTR B=640' if acktr else 'A2C B=80', ', '.join('%s %.3f' % kv for kv in r.items())))
dist.barrier(); dist.destroy_process_group()
