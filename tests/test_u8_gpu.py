"""GPU: Atari frames kept as bytes on the device (RolloutStorage(obs_dtype=torch.uint8)).  The
reference stores observations as float32 holding 0..255 (storage.py:12, F6); the uint8 kernels read the same values as
bytes, convert exactly and multiply in the same order, so the bar is BIT-IDENTICAL results against the float storage:
operators (conv1 forward / weight gradient / KFAC input factor, with and without the minibatch index), the rollout-side
API, and whole PPO / A2C / ACKTR updates."""
from collections import OrderedDict

import numpy as np
import pytest
import torch

import synth
from helpers import t

pytestmark = pytest.mark.gpu


class Discrete(object):
    def __init__(self, n):
        self.n = n


def _frames(B, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, 256, (B, 4, 84, 84), generator=g, dtype=torch.int32).to(torch.uint8)


def _same(a, b):
    return torch.equal(a.view(torch.int32), b.view(torch.int32))


@pytest.mark.parametrize('B,use_idx', [(1, False), (37, False), (80, True), (300, True), (1024, False)])
def test_conv1_operators_bit_identical(B, use_idx):
    from a2c_ppo_acktr import _hip, ops
    g = torch.Generator().manual_seed(B)
    src = B + 5 if use_idx else B
    x8 = _frames(src, 11 + B).cuda()
    xf = x8.float()
    idx = torch.randperm(src, generator=g)[:B].cuda() if use_idx else None
    w = (torch.randn(32, 4, 8, 8, generator=g) * 0.05).cuda()
    bias = torch.randn(32, generator=g).cuda()
    y8 = ops.conv2d_fwd(x8, w, bias, 4, in_scale=1.0 / 255.0, relu=True, gather_idx=idx)
    yf = ops.conv2d_fwd(xf, w, bias, 4, in_scale=1.0 / 255.0, relu=True, gather_idx=idx)
    assert y8.shape == (B, 32, 20, 20) and _same(y8, yf)
    dy = torch.randn(B, 32, 20, 20, generator=g).cuda()
    out = []
    for x in (x8, xf):
        dwb = torch.zeros(32 * 256 + 32, device='cuda')
        ops.conv2d_wgrad(x, dy, (32, 4, 8, 8), 4, dwb[:32 * 256], dwb[32 * 256:], in_scale=1.0 / 255.0, gather_idx=idx)
        out.append(dwb)
    if B >= 256:        # the float path runs the same stationary-accumulator kernel from 256 rows on: bit-identical
        assert _same(out[0], out[1])
    else:               # below that the float path is the tile GEMM (another summation order): fp32 agreement
        ref = out[1].double()
        assert float((out[0].double() - ref).norm() / ref.norm()) < 2e-5
    # KFAC input factor (kfac.py:29-46)
    L = _hip.lib()
    ws = ops.workspace(x8.device, L.acktr_kfac_cov_workspace_bytes(256, B * 400), 'kfac')
    a8, af = torch.zeros(256, 256, device='cuda'), torch.zeros(256, 256, device='cuda')
    alpha = (1.0 / 255.0) ** 2 / (B * 400.0 ** 2)
    _hip.check(L.acktr_kfac_cov_a_conv1_u8(_hip.dev_ptr(x8, torch.uint8), _hip.dev_ptr(idx, torch.int64), B, alpha,
                                           _hip.dev_ptr(a8), _hip.dev_ptr(ws), ws.numel(), _hip.cur_stream()), 'cov u8')
    _hip.check(L.acktr_kfac_cov_a_conv_f32(_hip.dev_ptr(xf), _hip.dev_ptr(idx, torch.int64), B, 4, 84, 84, 8, 8, 4, alpha,
                                           _hip.dev_ptr(af), _hip.dev_ptr(ws), ws.numel(), _hip.cur_stream()), 'cov f32')
    assert torch.equal(a8, a8.t())
    if B >= 64:
        assert _same(a8, af)
    else:
        assert float((a8.double() - af.double()).norm() / af.double().norm()) < 2e-5


def _storage(T, N, dtype, seed):
    from a2c_ppo_acktr.storage import RolloutStorage
    ro = synth.make_rollouts(T, N, (1,), 'discrete', 6, salt=seed)
    rs = RolloutStorage(T, N, (4, 84, 84), Discrete(6), 1, obs_dtype=dtype)
    assert rs.obs.dtype == dtype
    frames = _frames((T + 1) * N, seed).view(T + 1, N, 4, 84, 84)
    rs.obs[0].copy_(frames[0].float())                  # main.py:99: the env's float frames, converted exactly
    for k in ('rewards', 'value_preds', 'returns', 'action_log_probs', 'actions', 'masks', 'bad_masks'):
        getattr(rs, k).copy_(t(ro[k]))
    rs.to('cuda')
    for s in range(T):                                   # main.py:130-134: insert() on device tensors
        rs.insert(frames[s + 1].float().cuda(), torch.zeros(N, 1).cuda(), rs.actions[s].clone(),
                  rs.action_log_probs[s].clone(), rs.value_preds[s].clone(), rs.rewards[s].clone(),
                  rs.masks[s + 1].clone(), rs.bad_masks[s + 1].clone())
    assert torch.equal(rs.obs.float().cpu(), frames.float())
    return rs, ro


def _run_updates(algo_name, dtype):
    from a2c_ppo_acktr import algo
    from a2c_ppo_acktr.model import Policy
    T, N = (8, 40) if algo_name == 'ppo' else (5, 16)      # PPO: minibatches of 160 rows through the index
    torch.manual_seed(5)
    pol = Policy((4, 84, 84), Discrete(6))
    pol.to('cuda')
    rs, ro = _storage(T, N, dtype, 900)
    if algo_name == 'ppo':
        rs.compute_returns(t(ro['next_value']).cuda(), True, 0.99, 0.95, True)
        agent = algo.PPO(pol, 0.1, 2, 2, 0.5, 0.01, lr=2.5e-4, eps=1e-5, max_grad_norm=0.5)
    else:
        rs.compute_returns(t(ro['next_value']).cuda(), False, 0.99, 0.95, False)
        agent = algo.A2C_ACKTR(pol, 0.5, 0.01, lr=7e-4, eps=1e-5, alpha=0.99, max_grad_norm=0.5,
                               acktr=algo_name == 'acktr')
    p0 = OrderedDict((k, v.detach().cpu().clone()) for k, v in pol.state_dict().items())
    losses = []
    for it in range(3):                                # the third update replays the captured graph
        torch.manual_seed(60 + it)
        losses.append(agent.update(rs))
    with torch.no_grad():
        v, a, lp, _ = pol.act(rs.obs[1], None, None, deterministic=True)
        gv = pol.get_value(rs.obs[2], None, None)
    return (np.array(losses), pol.flat_params.detach().clone(), v, a, lp, gv, p0, rs)


def _ppo_fp64(p0, rs):
    """The same three PPO updates by the oracle in float64 (CPU): the trajectory without rounding."""
    from oracle import algo_ref
    P = OrderedDict((k, v.double()) for k, v in p0.items())
    ro = {k: getattr(rs, k).detach().cpu() for k in ('obs', 'actions', 'value_preds', 'returns', 'action_log_probs')}
    ro = {k: (v.double() if v.dtype in (torch.float32, torch.uint8) else v) for k, v in ro.items()}
    ro['rewards'] = rs.rewards.detach().cpu().double()
    opt = algo_ref.adam_init(P)
    out = []
    for it in range(3):
        torch.manual_seed(60 + it)
        out.append(algo_ref.ppo_update(P, opt, ro, 0.1, 2, 2, 0.5, 0.01, 2.5e-4, 1e-5, 0.5))
    return np.array(out)


@pytest.mark.parametrize('algo_name', ['ppo', 'a2c', 'acktr'])
def test_updates_bit_identical_to_float_storage(algo_name):
    res = [_run_updates(algo_name, dtype) for dtype in (torch.uint8, torch.float32)]
    # A2C / ACKTR at 80 rows: the float path's conv1 weight gradient (and, ACKTR below 64 rows, its factor) runs on the
    # tile GEMM, another summation order -> fp32 agreement; PPO at 160-row minibatches: the same... also the tile GEMM
    # below 256 rows.  Everything the uint8 kernels share with the float path is bit-identical (operator test above).
    # The first update agrees to rounding.  Later ones may drift: Adam's element-wise normalisation turns a 1e-7 difference
    # on a small-gradient element into an O(lr) difference of that parameter (the same effect as RMSprop's in
    # test_policy_gpu.test_a2c_update_vs_reference), and which elements are that close to zero depends on the trajectory.
    # And a ReLU whose pre-activation sits within rounding of zero passes its gradient in one run and not in the other.
    # (Seen with this seed inside the whole GPU suite: the two runs agree to 1e-7 in every gradient for five minibatch
    # steps; in the sixth every gradient from the FC layer down differs by 1e-3 while the heads' agree -- one gate of the
    # 512 x 160 FC outputs -- and the third update's losses end 1.6e-4 apart.  Alone in a process the same seed gives 6e-8.)
    # So beyond rounding the bar is the loss / gradient budget of BASELINE.json against a float64 run of the same
    # updates (the oracle, CPU): 1e-3 on the losses, with the float64 distances of both storages printed.
    np.testing.assert_allclose(res[0][0][0], res[1][0][0], rtol=2e-5, atol=1e-7)
    err = float((res[0][1].double() - res[1][1].double()).norm() / res[1][1].double().norm())
    close = np.allclose(res[0][0], res[1][0], rtol=2e-5, atol=1e-7) and err < (1e-3 if algo_name == 'acktr' else 2e-5)
    if not close:
        assert algo_name == 'ppo', (res[0][0], res[1][0], err)
        truth = _ppo_fp64(res[1][6], res[1][7])
        e8, ef = np.abs(res[0][0] / truth - 1).max(), np.abs(res[1][0] / truth - 1).max()
        print('uint8 vs float storage beyond rounding: losses', res[0][0].tolist(), res[1][0].tolist(), 'float64', truth.tolist(),
              'distance to float64: uint8 %.3g, float %.3g; parameters uint8 vs float %.3g' % (e8, ef, err))
        assert e8 <= 1e-3 and ef <= 1e-3 and err < 1e-2, (e8, ef, err)
        return
    assert torch.equal(res[0][3], res[1][3])
    for k in (2, 4, 5):
        np.testing.assert_allclose(res[0][k].cpu().numpy(), res[1][k].cpu().numpy(), rtol=1e-4, atol=1e-5)


def test_ppo_bench_shape_bit_identical():
    """One PPO update at the benchmarked minibatch size (1024 rows through the index from a 128 x 32 storage): at this
    size both storages run the SAME kernels, so losses and parameters must agree bit for bit."""
    from a2c_ppo_acktr import algo
    from a2c_ppo_acktr.model import Policy
    res = []
    for dtype in (torch.uint8, torch.float32):
        torch.manual_seed(6)
        pol = Policy((4, 84, 84), Discrete(6))
        pol.to('cuda')
        rs, ro = _storage(128, 32, dtype, 901)
        rs.compute_returns(t(ro['next_value']).cuda(), True, 0.99, 0.95, True)
        agent = algo.PPO(pol, 0.1, 1, 4, 0.5, 0.01, lr=2.5e-4, eps=1e-5, max_grad_norm=0.5)
        losses = []
        for it in range(3):
            torch.manual_seed(70 + it)
            losses.append(agent.update(rs))
        res.append((np.array(losses), pol.flat_params.detach().clone()))
        del rs, agent, pol
        torch.cuda.empty_cache()
    assert np.array_equal(res[0][0], res[1][0])
    assert _same(res[0][1], res[1][1])


def test_u8_rejected_where_unsupported():
    from a2c_ppo_acktr.model import Policy
    from a2c_ppo_acktr.storage import RolloutStorage

    class Box(object):
        shape = (3,)
    with pytest.raises(ValueError):
        RolloutStorage(4, 2, (11,), Box(), 1, obs_dtype=torch.uint8)
    with pytest.raises(ValueError):
        RolloutStorage(4, 2, (4, 84, 84), Discrete(6), 1, obs_dtype=torch.float64)
    pol = Policy((11,), Box())
    pol.to('cuda')
    with pytest.raises(RuntimeError, match='float32'):
        pol.get_value(torch.zeros(2, 11, dtype=torch.uint8, device='cuda'), None, None)
