"""Where a GAIL discriminator update spends its time (configs[4] shapes): python tools/gail_bench.py"""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'pytorch-a2c-ppo-acktr-gail_amd'))
import torch
from bench import Box
from a2c_ppo_acktr.algo import gail
from a2c_ppo_acktr.storage import RolloutStorage

dev = torch.device('cuda:0')
T, N, Ds, Da = 2048, 1024, 17, 6
g = torch.Generator(device=dev).manual_seed(25)
torch.manual_seed(11)
disc = gail.Discriminator(Ds + Da, 100, dev)
rs = RolloutStorage(1, 1, (1,), Box(Da), 1)
rs.num_steps = T
rs.obs = torch.randn(T + 1, N, Ds, device=dev, generator=g)
rs.actions = torch.randn(T, N, Da, device=dev, generator=g)
for k, shp in (('rewards', (T, N, 1)), ('value_preds', (T + 1, N, 1)), ('action_log_probs', (T, N, 1)), ('returns', (T + 1, N, 1)),
               ('recurrent_hidden_states', (T + 1, N, 1))):
    setattr(rs, k, torch.zeros(*shp, device=dev))
rs.masks = torch.ones(T + 1, N, 1, device=dev)
rs.bad_masks = torch.ones(T + 1, N, 1, device=dev)


def timeit(f, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


B = 128
ps, pa, es, ea = (torch.randn(B, d, device=dev) for d in (Ds, Da, Ds, Da))
al = torch.rand(B, 1)
for _ in range(4):
    disc._step(ps, pa, es, ea, al)
print('_step (graph replay, fixed inputs)      : %8.1f us' % timeit(lambda: disc._step(ps, pa, es, ea, al), 200))
os.environ['ACKTR_HIP_GRAPH'] = '0'
print('_step (launch by launch)                : %8.1f us' % timeit(lambda: disc._step(ps, pa, es, ea, al), 100))
os.environ['ACKTR_HIP_GRAPH'] = '1'
with tempfile.TemporaryDirectory() as td:
    gcpu = torch.Generator().manual_seed(26)
    fn = os.path.join(td, 'trajs.pt')
    torch.save({'states': torch.randn(64, 1000, Ds, generator=gcpu), 'actions': torch.randn(64, 1000, Da, generator=gcpu),
                'rewards': torch.randn(64, 1000, generator=gcpu), 'lengths': torch.full((64,), 1000, dtype=torch.int64)}, fn)
    ds = gail.ExpertDataset(fn, num_trajectories=64, subsample_frequency=20)
loader = torch.utils.data.DataLoader(ds, batch_size=128, shuffle=True, drop_last=True)
nb = len(loader)
t0 = time.perf_counter()
for _ in range(3):
    for eb in loader:
        pass
print('expert DataLoader, per batch            : %8.1f us (%d batches)' % ((time.perf_counter() - t0) / 3 / nb * 1e6, nb))
torch.cuda.synchronize(); t0 = time.perf_counter()
gen = rs.feed_forward_generator(None, mini_batch_size=128)
first = next(gen)
torch.cuda.synchronize(); t1 = time.perf_counter()
for i, pb in zip(range(nb - 1), gen):
    pass
torch.cuda.synchronize(); t2 = time.perf_counter()
print('feed_forward_generator: first batch     : %8.1f us (CPU randperm over %d rows)' % ((t1 - t0) * 1e6, T * N))
print('feed_forward_generator: per later batch : %8.1f us' % ((t2 - t1) / (nb - 1) * 1e6))
t0 = time.perf_counter()
for _ in range(3):
    list(zip(range(nb), rs.minibatch_index_batches(mini_batch_size=128)))
torch.cuda.synchronize()
rp = (time.perf_counter() - t0) / 3 * 1e6
print('minibatch_index_batches, warmed         : %8.1f us per call (randperm + one copy to the device)' % rp)
t0 = time.perf_counter()
for _ in range(3):
    for eb in gail._expert_batches(loader):
        pass
print('expert batches, dataset fast path       : %8.1f us per batch' % ((time.perf_counter() - t0) / 3 / nb * 1e6))
obsfilt = lambda x, update=False: x
for _ in range(2):
    disc.update(loader, rs, obsfilt)
up = timeit(lambda: disc.update(loader, rs, obsfilt), 4)
print('Discriminator.update, %d batches        : %8.1f us per batch (%.1f net of the permutation)' % (nb, up / nb, (up - rp) / nb))
