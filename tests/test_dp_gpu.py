"""The data-parallel update path END TO END on the HIP kernels: 2 ranks (both on cuda:0, `gloo` process group -- RCCL
refuses two ranks on one device) run PPO.update / A2C_ACKTR(acktr).update on their shards of every minibatch and must
end with the parameters of the single-process update on the full minibatches (same seeds).  Covers what the 8-GPU
bench relies on: index sharding through the fused gather, 1/B_global loss scaling, the flat-bucket all-reduce between
the captured forward/backward and optimiser graphs, factor all-reduce and sharded eigendecompositions."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir, algo_name, backend='gloo'):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.dirname(here), os.path.join(os.path.dirname(here), 'pytorch-a2c-ppo-acktr-gail_amd'), here,
              os.path.join(here, 'golden')):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    import synth
    from helpers import t
    from test_policy_gpu import Discrete, make_storage
    import datetime
    dev = rank if (backend == 'nccl' and world > 1) else 0      # real multi-GPU: one device per rank
    torch.cuda.set_device(dev)
    if world > 1 or backend == 'nccl':
        os.environ['MASTER_ADDR'] = '127.0.0.1'
        os.environ['MASTER_PORT'] = str(port)
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        kw = dict(device_id=torch.device('cuda', dev)) if backend == 'nccl' else {}
        dist.init_process_group(backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60), **kw)
    from a2c_ppo_acktr import algo
    from a2c_ppo_acktr.model import Policy
    T, N, A = 6, 12, 6                                   # 72 samples: minibatches of 36 -> shards of 18
    tag = algo_name
    if algo_name.endswith('_uneven'):                    # 75 samples: PPO minibatches of 37 -> shards of 18 / 19 (and the
        T, N = 3, 25                                     # 75th row dropped, drop_last); A2C / ACKTR shards of 37 / 38 rows
        algo_name = algo_name[:-len('_uneven')]
    torch.manual_seed(3)
    if algo_name == 'ppo_fullgraph':                     # the collectives captured into the epoch graph (opt-in mode)
        os.environ['ACKTR_DP_GRAPH'] = 'full'
        algo_name = tag = 'ppo'
    rec = algo_name.endswith('_rec')                     # recurrent: ranks split the ENVS of a minibatch
    pol = Policy((4, 84, 84), Discrete(A), base_kwargs={'recurrent': rec})
    pol.to('cuda')
    ro = synth.make_rollouts(T, N, (4, 84, 84), 'discrete', A, salt=610, p_mask=0.8 if rec else 0.97,
                             hidden=512 if rec else None)
    rs = make_storage(ro, (4, 84, 84), Discrete(A))
    if rec:
        rs.recurrent_hidden_states = t(ro['recurrent_hidden_states']).cuda()
    if algo_name in ('ppo', 'ppo_rec'):
        rs.compute_returns(t(ro['next_value']).cuda(), True, 0.99, 0.95, True)
        agent = algo.PPO(pol, clip_param=0.1, ppo_epoch=2, num_mini_batch=2, value_loss_coef=0.5, entropy_coef=0.01,
                         lr=2.5e-4, eps=1e-5, max_grad_norm=0.5)
        n_upd = 3                                        # the third update replays the captured graphs
    elif algo_name in ('a2c', 'a2c_rec'):
        rs.compute_returns(t(ro['next_value']).cuda(), False, 0.99, 0.95, False)
        agent = algo.A2C_ACKTR(pol, 0.5, 0.01, lr=7e-4, eps=1e-5, alpha=0.99, max_grad_norm=0.5)
        n_upd = 2 if rec else 4                          # feed-forward: updates 2.. replay the graph pieces
    else:
        rs.compute_returns(t(ro['next_value']).cuda(), False, 0.99, 0.95, False)
        agent = algo.A2C_ACKTR(pol, value_loss_coef=0.5, entropy_coef=0.01, acktr=True)
        agent.optimizer.Tf = 2
        n_upd = 6                                        # updates 3 and 5 replay graph pieces around the collectives
    if backend == 'nccl' and world == 1:                 # one rank, but through the data-parallel code path on RCCL
        from a2c_ppo_acktr.parallel import DataParallel
        agent.dp = DataParallel(force_active=True)
        if hasattr(agent.optimizer, 'dp'):
            agent.optimizer.dp = agent.dp
    losses = []
    for it in range(n_upd):
        torch.manual_seed(40 + it)                       # same CPU stream on every rank: permutation and Fisher noise
        losses.append(agent.update(rs))
        print('[dp test] %s world %d rank %d update %d done' % (algo_name, world, rank, it), flush=True)
    torch.cuda.synchronize()
    if rank == 0:
        torch.save({'flat': pol.flat_params.detach().cpu(), 'losses': torch.tensor(losses, dtype=torch.float64)},
                   os.path.join(out_dir, '%s_w%d%s.pt' % (tag, world, '_rccl' if backend == 'nccl' else '')))
    if world > 1 or backend == 'nccl':
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize('algo_name', ['ppo', 'acktr', 'a2c', 'ppo_rec', 'a2c_rec', 'ppo_uneven', 'acktr_uneven', 'a2c_uneven'])
def test_two_rank_update_matches_single_process(tmp_path, algo_name):
    out = str(tmp_path)
    mp.spawn(_worker, args=(1, 0, out, algo_name), nprocs=1, join=True)
    mp.spawn(_worker, args=(2, _free_port(), out, algo_name), nprocs=2, join=True)
    one = torch.load(os.path.join(out, '%s_w1.pt' % algo_name))
    two = torch.load(os.path.join(out, '%s_w2.pt' % algo_name))
    d = np.abs(two['losses'].numpy() - one['losses'].numpy())
    err = float((two['flat'] - one['flat']).norm() / one['flat'].norm())
    print('[dp test] %s: max |loss diff| %.3g (rel %.3g), params rel %.3g'
          % (algo_name, d.max(), (d / np.abs(one['losses'].numpy())).max(), err))
    np.testing.assert_allclose(two['losses'].numpy(), one['losses'].numpy(), rtol=2e-4, atol=1e-6)
    assert err < (1e-3 if algo_name.startswith('acktr') else 2e-5), err


@pytest.mark.parametrize('algo_name', ['ppo', 'acktr', 'a2c', 'ppo_fullgraph'])
def test_rccl_collectives_single_rank(tmp_path, algo_name):
    """The data-parallel code path on the REAL RCCL backend (one rank is all a one-GPU box can host): split graphs,
    the asynchronous late-bucket all-reduce on RCCL's stream, factor / eigen all-reduces.  With one rank every
    collective is an identity, so the result must equal the plain single-process update."""
    out = str(tmp_path)
    base = 'ppo' if algo_name == 'ppo_fullgraph' else algo_name
    mp.spawn(_worker, args=(1, 0, out, base), nprocs=1, join=True)
    mp.spawn(_worker, args=(1, _free_port(), out, algo_name, 'nccl'), nprocs=1, join=True)
    one = torch.load(os.path.join(out, '%s_w1.pt' % base))
    two = torch.load(os.path.join(out, '%s_w1_rccl.pt' % base))
    np.testing.assert_allclose(two['losses'].numpy(), one['losses'].numpy(), rtol=2e-4, atol=1e-6)
    err = float((two['flat'] - one['flat']).norm() / one['flat'].norm())
    assert err < (1e-3 if algo_name.startswith('acktr') else 2e-5), err


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs two GPUs: two ranks over real RCCL (xGMI)')
@pytest.mark.parametrize('algo_name', ['ppo', 'acktr', 'a2c', 'ppo_rec', 'ppo_uneven', 'acktr_uneven'])
def test_two_rank_update_matches_single_process_rccl(tmp_path, algo_name):
    """The same check with one GPU per rank and the `nccl` (= RCCL) backend: runs wherever the box has two GPUs (the
    driver's multi-GPU node), skips on a one-GPU box.  Exercises what gloo cannot: in-place device all-reduces on RCCL's
    stream, the asynchronous late-bucket all-reduce overlapped with the backward, owner broadcasts of the eigenbases."""
    out = str(tmp_path)
    mp.spawn(_worker, args=(1, 0, out, algo_name), nprocs=1, join=True)
    mp.spawn(_worker, args=(2, _free_port(), out, algo_name, 'nccl'), nprocs=2, join=True)
    one = torch.load(os.path.join(out, '%s_w1.pt' % algo_name))
    two = torch.load(os.path.join(out, '%s_w2_rccl.pt' % algo_name))
    np.testing.assert_allclose(two['losses'].numpy(), one['losses'].numpy(), rtol=2e-4, atol=1e-6)
    err = float((two['flat'] - one['flat']).norm() / one['flat'].norm())
    assert err < (1e-3 if algo_name.startswith('acktr') else 2e-5), err
