"""Where the HOST time of a data-parallel (eager-launch) PPO update goes: cProfile of a few updates on one RCCL rank."""
import cProfile, os, pstats, sys, io, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'pytorch-a2c-ppo-acktr-gail_amd')]
import torch
import torch.distributed as dist
import bench
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29541')
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
dev = torch.device('cuda:0')
ro = bench.synth_rollouts(bench.T_STEPS, bench.PROCS_PER_GPU, dev, seed=1234)
pol, agent = bench.build_agent(dev, seed=7)
from a2c_ppo_acktr.parallel import DataParallel
agent.dp = DataParallel(force_active=True)
rs = bench.fill_storage(ro, dev)
rs.compute_returns(ro['next_value'], True, 0.99, 0.95, True)
for i in range(5):
    agent.update(rs)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(10):
    agent.update(rs)
torch.cuda.synchronize()
print('ms per update', (time.perf_counter() - t0) * 100)
# host-only pace: the same without waiting for the device between updates is what the loop above already does; profile it
pr = cProfile.Profile()
pr.enable()
for i in range(10):
    agent.update(rs)
pr.disable()
torch.cuda.synchronize()
out = io.StringIO()
pstats.Stats(pr, stream=out).sort_stats('tottime').print_stats(28)
print(out.getvalue()[:6000])
dist.destroy_process_group()
