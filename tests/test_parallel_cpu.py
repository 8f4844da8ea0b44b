"""CPU (gloo, world_size 2): the data-parallel sharding math of a2c_ppo_acktr/parallel.py.

The HIP engine cannot run here, so the per-rank "local gradient" is produced by the CPU oracle; what is
under test is the host logic that surrounds it on the GPU path: identical permutations on every rank,
contiguous balanced shards, 1/B_global loss scaling, SUM all-reduce of one flat bucket, KFAC factor
scalings with the global batch, and the layer-wise eigensolve assignment."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import synth
from helpers import params_torch, t
from oracle import algo_ref, kfac_ref, policy_ref


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(1)
    from a2c_ppo_acktr.parallel import DataParallel
    dp = DataParallel()
    assert dp.active and dp.world == world and dp.rank == rank
    B, od, A = 50, 11, 3                       # B not divisible by world*anything special: unequal shards
    P = params_torch(synth.make_params(policy_ref.mlp_param_shapes(od, 64, A), 5))
    ro = synth.make_rollouts(1, B, (od,), 'box', A, salt=77)
    obs, act = t(ro['obs'][0]), t(ro['actions'][0])
    vp, olp = t(ro['value_preds'][0]), t(ro['action_log_probs'][0])
    ret, adv = t(synth.normalish((B, 1), 1)), t(synth.normalish((B, 1), 2))
    torch.manual_seed(123)                     # same seed on every rank -> same permutation (F7)
    perm = torch.randperm(B)
    idx = dp.shard(perm)
    lo, hi = dp.bounds(B)
    assert idx.numel() == hi - lo
    # local loss terms scaled by 1/B_global (what the fused loss kernel does with inv_batch)
    for v in P.values():
        v.requires_grad_(True)
    values, logp, ent = policy_ref.evaluate_actions(P, obs[idx], act[idx])
    ratio = torch.exp(logp - olp[idx])
    s1, s2 = ratio * adv[idx], torch.clamp(ratio, 0.8, 1.2) * adv[idx]
    al = -torch.min(s1, s2).sum() / B
    vpc = vp[idx] + (values - vp[idx]).clamp(-0.2, 0.2)
    vl = 0.5 * torch.max((values - ret[idx]).pow(2), (vpc - ret[idx]).pow(2)).sum() / B
    en = ent * idx.numel() / B                 # entropy mean over the shard -> contribution to the global mean
    loss = vl * 0.5 + al - en * 0.01
    grads = torch.autograd.grad(loss, list(P.values()))
    flat = torch.cat([g.reshape(-1) for g in grads])
    dp.all_reduce_sum_(flat)
    # KFAC factor partial sums with GLOBAL scalings
    g_local = torch.randn(B, 7, generator=torch.Generator().manual_seed(9))[perm][lo:hi]
    gg = (float(B) * g_local.t() @ g_local)    # alpha = B_global, summed over ranks
    dp.all_reduce_sum_(gg)
    if rank == 0:
        torch.save({'flat': flat, 'gg': gg, 'perm': perm}, os.path.join(out_dir, 'dp.pt'))
    dist.barrier()
    dist.destroy_process_group()


def test_dp_gradient_and_factor_sharding(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    res = torch.load(os.path.join(str(tmp_path), 'dp.pt'))
    B, od, A = 50, 11, 3
    P = params_torch(synth.make_params(policy_ref.mlp_param_shapes(od, 64, A), 5))
    ro = synth.make_rollouts(1, B, (od,), 'box', A, salt=77)
    perm = res['perm']
    sel = lambda x: t(x)[perm]
    losses, G = algo_ref.ppo_minibatch_grads(P, sel(ro['obs'][0]), sel(ro['actions'][0]), sel(ro['value_preds'][0]),
                                             sel(synth.normalish((B, 1), 1)), sel(ro['action_log_probs'][0]),
                                             sel(synth.normalish((B, 1), 2)), 0.2, 0.5, 0.01)
    full = torch.cat([g.reshape(-1) for g in G.values()])
    assert float((res['flat'] - full).norm() / full.norm()) < 1e-5
    g_all = torch.randn(B, 7, generator=torch.Generator().manual_seed(9))[perm]
    assert float((res['gg'] - kfac_ref.cov_g(g_all, 'linear')).norm() / kfac_ref.cov_g(g_all, 'linear').norm()) < 1e-5


def test_shard_bounds_partition():
    from a2c_ppo_acktr.parallel import DataParallel
    dp = DataParallel()
    assert not dp.active and dp.shard(torch.arange(5)).numel() == 5
    for world in (2, 3, 8):
        for n in (1, 7, 64, 1023):
            cover = []
            for r in range(world):
                dp.world, dp.rank = world, r
                lo, hi = dp.bounds(n)
                cover += list(range(lo, hi))
            assert cover == list(range(n))


def test_eig_owner_assignment_balanced_and_deterministic():
    from a2c_ppo_acktr.parallel import assign_eig_owners
    sizes = [256, 32, 1, 32, 512, 64, 1, 64, 576, 32, 1, 32, 1568, 512, 1, 512, 512, 1, 1, 1, 512, 6, 1, 6]
    for world in (1, 2, 4, 8):
        own = assign_eig_owners(sizes, world)
        assert own == assign_eig_owners(list(sizes), world)
        assert set(own) <= set(range(world))
        loads = [sum(float(s) ** 3 for s, o in zip(sizes, own) if o == r) for r in range(world)]
        crit = max(float(s) ** 3 for s in sizes)
        assert max(loads) <= max(crit, sum(loads) / world) * 1.34 + 1    # LPT bound (4/3) against the 1568 factor
    assert assign_eig_owners(sizes, 8)[12] != assign_eig_owners(sizes, 8)[8]   # 1568 and 576 on different ranks


# ---------------------------------------------------------------------------------------------------------------
# bench.py's own N-rank launcher (the driver may run `python bench.py --gpus N` without torch.distributed.run)
# ---------------------------------------------------------------------------------------------------------------
BENCH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'bench.py')


def _run(cmd, env=None, timeout=240):
    import subprocess
    e = dict(os.environ)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run(cmd, env=e, capture_output=True, text=True, timeout=timeout)


@pytest.mark.parametrize('world', [2, 4])
def test_bench_self_launch_reaches_rendezvous(world):
    """`python bench.py --gpus N` with no launcher environment starts N fresh ranks itself; `--launch-check` stops
    before the first device call (gloo rendezvous + one all-reduce), so the launcher runs on a CPU box."""
    import json
    import sys
    r = _run([sys.executable, BENCH, '--gpus', str(world), '--launch-check'])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout                       # stdout carries exactly one line, rank 0's
    rec = json.loads(lines[0])
    assert rec == dict(launch_check=True, world=world, backend='gloo', sum=world * (world + 1) / 2)


def test_bench_under_torch_distributed_run():
    """The driver's documented N > 1 form: python -m torch.distributed.run ... bench.py --gpus N."""
    import json
    import sys
    r = _run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
              '127.0.0.1', '--master-port', str(_free_port()), BENCH, '--gpus', '2', '--launch-check'])
    assert r.returncode == 0, r.stderr[-2000:]
    rec = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert rec == [dict(launch_check=True, world=2, backend='gloo', sum=3.0)]


def test_bench_gpus_beyond_device_count_fails_cleanly():
    """More ranks than GPUs: a message naming the device count and exit code 2 before anything is started -- not an
    AssertionError from inside a rank."""
    import sys
    n = torch.cuda.device_count() + 2
    r = _run([sys.executable, BENCH, '--gpus', str(n)], timeout=120)
    assert r.returncode == 2
    assert 'needs %d GPUs' % n in r.stderr and 'exposes %d' % (n - 2) in r.stderr
    assert 'AssertionError' not in r.stderr and r.stdout.strip() == ''
    # and a launcher environment that disagrees with --gpus
    r = _run([sys.executable, BENCH, '--gpus', '2'], env=dict(WORLD_SIZE='3', RANK='0', LOCAL_RANK='0'), timeout=120)
    assert r.returncode == 2 and 'WORLD_SIZE=3' in r.stderr


def test_bench_self_launch_propagates_rank_failure():
    """A rank that dies takes the command down with a non-zero exit code (the other ranks are stopped, not left in
    the rendezvous)."""
    import sys
    import time
    t0 = time.time()
    r = _run([sys.executable, BENCH, '--gpus', '2', '--launch-check'], env=dict(ACKTR_BENCH_TEST_FAIL_RANK='1'))
    assert r.returncode != 0
    assert 'rank 1 exited' in r.stderr
    assert time.time() - t0 < 120


# ---------------------------------------------------------------------------------------------------------------
# ACKTR: owner broadcasts of the sharded eigendecompositions (algo/kfac.py::_exchange_eig), world 4 on gloo
# ---------------------------------------------------------------------------------------------------------------
def _eig_fill(i, n):
    """What 'rank owner[i] decomposed factor i' looks like in this test: values determined by the factor alone."""
    q = (torch.arange(n * n, dtype=torch.float32) * 1e-3 + float(i)).remainder(7.0) - 3.0
    d = torch.arange(n, dtype=torch.float32) + 0.5 * i
    return q, d


def _eig_worker(rank, world, port, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(1)
    from a2c_ppo_acktr.algo.kfac import KFACOptimizer
    from a2c_ppo_acktr.model import Policy
    from a2c_ppo_acktr.parallel import DataParallel

    class Discrete(object):
        n = 6
    torch.manual_seed(0)
    pol = Policy((4, 84, 84), Discrete())                  # CPU tensors: only the host logic runs here
    opt = KFACOptimizer(pol)
    opt.dp = DataParallel()
    assert opt.dp.world == world and opt.dp.rank == rank
    opt._setup(torch.device('cpu'))
    assert len(opt._eig_work) == 24 and sorted(set(opt._eig_owner)) == list(range(world))
    nan = float('nan')
    for b in (opt.qt_a, opt.qt_g, opt.d_a_flat, opt.d_g_flat):
        b.fill_(nan)                                       # foreign slots hold garbage until the owners' messages land
    for i, ((run, qt, dfl, off, doff, n), own) in enumerate(zip(opt._eig_work, opt._eig_owner)):
        if own == rank:
            q, d = _eig_fill(i, n)
            qt[off:off + n * n].copy_(q)
            dfl[doff:doff + n].copy_(d)
    opt._exchange_eig()
    ok = True
    for i, (run, qt, dfl, off, doff, n) in enumerate(opt._eig_work):
        q, d = _eig_fill(i, n)
        ok = ok and torch.equal(qt[off:off + n * n], q) and torch.equal(dfl[doff:doff + n], d)
    # the public views see the same storage
    m = opt.modules[6]                                     # Linear(1568 -> 512)
    ok = ok and tuple(opt.Q_a[m].shape) == (1568, 1568) and not bool(torch.isnan(opt.Q_a[m]).any())
    torch.save({'ok': bool(ok), 'owner': list(opt._eig_owner)}, os.path.join(out_dir, 'eig_r%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_eig_owner_broadcast_world4(tmp_path):
    world = 4
    mp.spawn(_eig_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(os.path.join(str(tmp_path), 'eig_r%d.pt' % r)) for r in range(world)]
    assert all(r['ok'] for r in res)
    assert all(r['owner'] == res[0]['owner'] for r in res)          # identical assignment on every rank
    sizes = [256, 32, 1, 32, 512, 64, 1, 64, 576, 32, 1, 32, 1568, 512, 1, 512, 512, 1, 1, 1, 512, 6, 1, 6]
    from a2c_ppo_acktr.parallel import assign_eig_owners
    assert res[0]['owner'] == assign_eig_owners(sizes, world)
