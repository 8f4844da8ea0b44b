"""GPU parity at the shapes bench.py times: the headline minibatch of 1024 rows read through
a gather index into a 128 x 32 storage, and the 8192-row minibatch over the 3.7 GB observation tensor of the
256-process side metric.  Grid sizes, split-K slab counts and tile choices at these sizes differ from the
golden cases (B <= 80), so they get their own checks against the CPU oracle (oracle/algo_ref.py) on identical
inputs.  Tolerance: north_star, losses and gradients within 1e-4 relative."""
from collections import OrderedDict

import numpy as np
import pytest
import torch

from helpers import cnn_gate_aware_grads, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-4
CLIP, C_V, C_E = 0.1, 0.5, 0.01


class Discrete(object):
    def __init__(self, n):
        self.n = n


def _policy(seed):
    from a2c_ppo_acktr.model import Policy
    torch.manual_seed(seed)
    pol = Policy((4, 84, 84), Discrete(6))
    P = OrderedDict((k, v.detach().clone()) for k, v in pol.state_dict().items())
    pol.to('cuda')
    return pol, P


def _rollout_cpu(T, N, seed):
    g = torch.Generator().manual_seed(seed)
    ro = {}
    ro['obs'] = torch.randint(0, 256, (T, N, 4, 84, 84), generator=g, dtype=torch.int32).float()
    ro['actions'] = torch.randint(0, 6, (T * N,), generator=g)
    ro['old_logp'] = 0.1 * torch.randn(T * N, generator=g) - 1.8
    ro['vp'] = torch.randn(T * N, generator=g)
    ro['ret'] = torch.randn(T * N, generator=g)
    ro['adv'] = torch.randn(T * N, generator=g)
    return ro


def _hip_step(pol, obs_dev, idx_dev, ro_dev, B):
    """What PPO.update runs per minibatch: forward(head=False) -> fused head/loss step -> backward."""
    from a2c_ppo_acktr import engine
    acc = torch.zeros(3, device='cuda')
    gflat = torch.full_like(pol.flat_params, 9.0)
    with torch.no_grad():
        assert engine.head_step_ok(pol, B)
        acts = engine.forward(pol, obs_dev, gather_idx=idx_dev, head=False)
        engine.head_step(pol, acts, engine.ALGO_PPO, ro_dev['actions'], acc, 1.0 / B, old_logp=ro_dev['old_logp'],
                         old_values=ro_dev['vp'], returns=ro_dev['ret'], adv=ro_dev['adv'], gather_idx=idx_dev,
                         clip=CLIP, value_coef=C_V, entropy_coef=C_E, gflat=gflat)
        engine.backward(pol, acts, gflat, keep=True)
    return acc.tolist(), gflat, acts


def test_ppo_minibatch_step_1024_gathered_vs_oracle():
    """One PPO minibatch step at B = 1024 through a gather index into a 128 x 32 storage (bench.py's headline shape)
    against oracle.algo_ref.ppo_minibatch_grads: three losses and every gradient at 1e-4."""
    from oracle import algo_ref
    T, N, B = 128, 32, 1024
    pol, P = _policy(7)
    ro = _rollout_cpu(T, N, 1234)
    torch.manual_seed(99)
    idx = torch.randperm(T * N)[:B].contiguous()
    flat_obs = ro['obs'].view(T * N, 4, 84, 84)
    dev = {k: v.cuda() for k, v in ro.items()}
    losses, gflat, acts = _hip_step(pol, dev['obs'].view(T * N, 4, 84, 84), idx.cuda(), dev, B)
    col = lambda k: ro[k][idx].view(B, 1)
    x = dict(obs=flat_obs[idx], act=ro['actions'][idx].view(B, 1), vp=col('vp'), ret=col('ret'), adv=col('adv'))
    ref_losses, G = algo_ref.ppo_minibatch_grads(P, x['obs'], x['act'], x['vp'], x['ret'], col('old_logp'), x['adv'],
                                                 CLIP, C_V, C_E)
    for a, b in zip(losses, ref_losses):
        assert abs(a - b) <= TOL * max(abs(b), 1e-3), ('losses', losses, ref_losses)
    grads = OrderedDict((n, pol.grad_view(p, gflat).cpu().numpy()) for n, p in pol.named_parameters())
    errs = {n: rel_err(g, G[n].numpy()) for n, g in grads.items()}
    print('B=1024 gathered step: loss', losses, 'worst grad rel err vs fp32 oracle %.2e' % max(errs.values()))
    if max(errs.values()) <= TOL:
        return
    # ReLU gates whose fp64 pre-activation is below fp32 accumulation noise are undetermined:
    # impose the HIP path's own gates on an fp64 forward/backward and require the disagreeing ones to be of that kind
    gates = [(v > 0).cpu().numpy() for v in (acts.y1, acts.y2, acts.y3, acts.feat)]
    xn = {k: v.numpy() for k, v in x.items()}
    ref64, flips = cnn_gate_aware_grads({k: v.detach().numpy() for k, v in P.items()}, gates, xn,
                                        col('old_logp').numpy(), CLIP, C_E)
    assert flips.size <= 8 * B // 80 and (flips.size == 0 or float(flips.max()) < 1e-5), \
        '%d gates disagree with fp64, largest |pre| %g' % (flips.size, float(flips.max()) if flips.size else 0)
    for n, g in grads.items():
        e = rel_err(g, ref64[n])
        assert e <= TOL, '%s gate-aware rel err %g (fp32 oracle rel err %g)' % (n, e, errs[n])
    print('   gate-aware fp64 check: %d undetermined gates' % flips.size)


def test_minibatch_8192_from_full_size_storage():
    """The 256-process side metric: minibatches of 8192 rows gathered from a [128*256, 4, 84, 84] fp32 tensor
    (3.7 GB; row offsets beyond 2^31 bytes).  Size-independent properties: rows gathered from the top of the big
    tensor give bit-identical conv1 output, conv1 weight gradient and step gradients as the same rows copied into a
    small tensor and read through the same kernels; the losses match the CPU oracle's forward on those rows."""
    from a2c_ppo_acktr import engine, ops
    from oracle import algo_ref
    T, N, B = 128, 256, 8192
    rows = T * N
    pol, P = _policy(7)
    g = torch.Generator(device='cuda').manual_seed(5)
    big = torch.randint(0, 256, (rows, 4, 84, 84), device='cuda', generator=g, dtype=torch.int32).float()
    gc = torch.Generator().manual_seed(6)
    # most rows from the last quarter of the tensor (byte offsets > 2.7 GB), some from the start, the very last row included
    top = rows - 1 - torch.randperm(rows // 4, generator=gc)[:B - 512]
    top[0] = rows - 1 if int(top.max()) != rows - 1 else top[0]
    low = torch.randperm(rows // 20, generator=gc)[:512]
    idx = torch.cat([top, low])[torch.randperm(B, generator=gc)].contiguous()
    assert int(idx.max()) == rows - 1 and idx.numel() == B and idx.unique().numel() == B
    ro = {'actions': torch.randint(0, 6, (rows,), generator=gc), 'old_logp': 0.1 * torch.randn(rows, generator=gc) - 1.8,
          'vp': torch.randn(rows, generator=gc), 'ret': torch.randn(rows, generator=gc),
          'adv': torch.randn(rows, generator=gc)}
    dev = {k: v.cuda() for k, v in ro.items()}
    idx_d = idx.cuda()
    losses_big, g_big, acts_big = _hip_step(pol, big, idx_d, dev, B)
    y1_big = acts_big.y1.clone()
    w1 = engine.cnn_layer_params(pol)['conv1'][0]
    dw_big = pol.grad_view(w1, g_big).clone()
    # the same rows in a small tensor, through the same (gathering) kernels with the identity index
    small = big[idx_d].contiguous()
    dev_small = {k: v[idx_d].contiguous() for k, v in dev.items()}
    ident = torch.arange(B, device='cuda')
    del big
    torch.cuda.empty_cache()
    losses_small, g_small, acts_small = _hip_step(pol, small, ident, dev_small, B)
    assert torch.equal(y1_big, acts_small.y1), 'conv1 output differs between the 3.7 GB tensor and the copied rows'
    assert torch.equal(dw_big, pol.grad_view(w1, g_small)), 'conv1 wgrad differs'
    assert torch.equal(g_big, g_small), 'step gradients differ'
    assert losses_big == losses_small
    # without an index at all (another loader path): same arithmetic up to summation order
    y1_plain = ops.conv2d_fwd(small, w1, engine.cnn_layer_params(pol)['conv1'][1].view(-1), 4, in_scale=1.0 / 255.0)
    assert rel_err(y1_plain.cpu().numpy(), y1_big.cpu().numpy()) < 2e-5
    # losses against the CPU oracle (fp32 forward of the same 8192 rows)
    from oracle import policy_ref
    with torch.no_grad():
        sc = small.cpu()
        values, logp, ent = policy_ref.evaluate_actions(P, sc, ro['actions'][idx].view(B, 1), None, None)
        col = lambda k: ro[k][idx].view(B, 1)
        vl, al, en = algo_ref.ppo_losses(values, logp, ent, col('vp'), col('ret'), col('old_logp'), col('adv'), CLIP)
    for a, b in zip(losses_big, (vl.item(), al.item(), en.item())):
        assert abs(a - b) <= TOL * max(abs(b), 1e-3), ('losses', losses_big, (vl.item(), al.item(), en.item()))
    assert bool(torch.isfinite(g_big).all()) and float(g_big.abs().max()) > 0


def test_ppo_update_is_run_to_run_deterministic():
    """Two fresh agents, same seeds, three updates each at the benchmarked minibatch size (1024 rows through the index; the
    third update replays the captured epoch graph): every reduction on the path -- split-K slabs, the slab sums that ride in
    the next launch as extra workgroups, the head step's finalize, the gradient norm -- runs in a fixed order, so losses and
    parameters must agree bit for bit."""
    from a2c_ppo_acktr import algo
    from a2c_ppo_acktr.model import Policy
    from a2c_ppo_acktr.storage import RolloutStorage

    class Discrete(object):
        n = 6
    T, N = 128, 32
    g = torch.Generator().manual_seed(77)
    obs = torch.randint(0, 256, (T + 1, N, 4, 84, 84), generator=g, dtype=torch.int32).to(torch.uint8)
    small = {k: torch.randn(*shape, generator=g) for k, shape in (('rewards', (T, N, 1)), ('value_preds', (T + 1, N, 1)))}
    alp = 0.1 * torch.randn(T, N, 1, generator=g) - 1.8
    act = torch.randint(0, 6, (T, N, 1), generator=g)
    masks = (torch.rand(T + 1, N, 1, generator=g) < 0.99).float()
    res = []
    for rep in range(2):
        torch.manual_seed(5)
        pol = Policy((4, 84, 84), Discrete())
        pol.to('cuda')
        rs = RolloutStorage(T, N, (4, 84, 84), Discrete(), 1)
        rs.to('cuda')
        rs.obs.copy_(obs.cuda())
        rs.rewards.copy_(small['rewards']); rs.value_preds.copy_(small['value_preds'])
        rs.action_log_probs.copy_(alp); rs.actions.copy_(act); rs.masks.copy_(masks)
        rs.compute_returns(torch.zeros(N, 1).cuda(), True, 0.99, 0.95, True)
        agent = algo.PPO(pol, 0.1, 2, 4, 0.5, 0.01, lr=2.5e-4, eps=1e-5, max_grad_norm=0.5)
        losses = []
        for it in range(3):
            torch.manual_seed(40 + it)
            losses.append(agent.update(rs))
        res.append((np.array(losses), pol.flat_params.detach().clone()))
        del rs, agent, pol
        torch.cuda.empty_cache()
    assert np.array_equal(res[0][0], res[1][0])
    assert torch.equal(res[0][1].view(torch.int32), res[1][1].view(torch.int32))
