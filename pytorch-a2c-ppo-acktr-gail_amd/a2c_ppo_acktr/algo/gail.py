"""GAIL discriminator and expert dataset -- host-side mirror of reference a2c_ppo_acktr/algo/gail.py.

Same classes, constructor signatures, public attributes (`trunk`, `optimizer`, `returns`, `ret_rms`) and
`state_dict` names as the reference; the arithmetic runs on the MI355X through the C-ABI (include/acktr_hip.h): the
three linear layers on the GEMM / skinny kernels (forward, input gradients, weight gradients), everything else
(minibatch assembly, BCE losses, gradient penalty and the element-wise links of its double backward, reward /
running-return / moments) on the kernels of csrc/gail.hip.  There is no CPU path.

Differences a caller can see: none in results (parity tests: tests/test_gail_gpu.py).  `predict_reward_batched`
is an addition: all T steps of a rollout in one pass, same per-step semantics as T calls of `predict_reward`.
"""
import ctypes

import os

import numpy as np
import torch
import torch.nn as nn
import torch.utils.data

from .. import _hip, ops
from .ppo import _FlatAdam

f32 = torch.float32


class RunningMeanStd(object):
    """stable_baselines3.common.running_mean_std.RunningMeanStd (gail.py:9,27): the published algorithm restated
    (batch moments merged with Chan et al.'s parallel update, float64); sb3 itself is not a dependency here."""

    def __init__(self, epsilon=1e-4, shape=()):
        self.mean = np.zeros(shape, np.float64)
        self.var = np.ones(shape, np.float64)
        self.count = epsilon

    def update(self, arr):
        self.update_from_moments(np.mean(arr, axis=0), np.var(arr, axis=0), arr.shape[0])

    def update_from_moments(self, batch_mean, batch_var, batch_count):
        delta = batch_mean - self.mean
        tot = self.count + batch_count
        new_mean = self.mean + delta * batch_count / tot
        m_2 = self.var * self.count + batch_var * batch_count + np.square(delta) * self.count * batch_count / tot
        self.mean, self.var, self.count = new_mean, m_2 / tot, tot


class Discriminator(nn.Module):
    def __init__(self, input_dim, hidden_dim, device):
        super(Discriminator, self).__init__()
        self.device = device
        # reference gail.py:17-20 (same modules, same default initialisation order -> same-seed weights)
        self.trunk = nn.Sequential(
            nn.Linear(input_dim, hidden_dim), nn.Tanh(),
            nn.Linear(hidden_dim, hidden_dim), nn.Tanh(),
            nn.Linear(hidden_dim, 1)).to(device)
        self.trunk.train()
        self._flatten()
        # gail.py:24: torch.optim.Adam(self.trunk.parameters()) -- lr 1e-3, betas (.9, .999), eps 1e-8
        self.optimizer = _FlatAdam(self, lr=1e-3, eps=1e-8)
        self.returns = None
        self.ret_rms = RunningMeanStd(shape=())
        self._in, self._hid = input_dim, hidden_dim

    # ------------------------------------------------------------------ flat parameter / gradient buffers
    def _flatten(self):
        ps = list(self.trunk.parameters())
        dev = ps[0].device
        offs, off = [], 0
        for p in ps:
            off = (off + 3) // 4 * 4
            offs.append(off)
            off += p.numel()
        total = (off + 3) // 4 * 4
        flat = torch.zeros(total, dtype=f32, device=dev)
        for p, o in zip(ps, offs):
            flat[o:o + p.numel()].copy_(p.data.reshape(-1))
            p.data = flat[o:o + p.numel()].view(p.shape)
        self.__dict__['flat_params'] = flat
        self.__dict__['flat_grads'] = torch.zeros_like(flat)
        self.__dict__['_offs'] = offs

    def _apply(self, fn, *a, **k):
        r = super(Discriminator, self)._apply(fn, *a, **k)
        self._flatten()
        return r

    def load_state_dict(self, *a, **k):
        r = super(Discriminator, self).load_state_dict(*a, **k)   # copies in place: views stay bound
        return r

    def _gview(self, i):
        p = list(self.trunk.parameters())[i]
        o = self._offs[i]
        return self.flat_grads[o:o + p.numel()].view(p.shape)

    def _check_dev(self):
        if not self.flat_params.is_cuda:
            raise RuntimeError('Discriminator must live on the MI355X device; this package has no CPU path')

    # ------------------------------------------------------------------ trunk forward (gail.py:17-20)
    def _forward(self, x):
        W1, b1, W2, b2, w3, b3 = list(self.trunk.parameters())
        h1 = ops.linear_fwd(x, W1, b1, 'tanh')
        h2 = ops.linear_fwd(h1, W2, b2, 'tanh')
        d = ops.linear_fwd(h2, w3, b3, None)
        return h1, h2, d

    # ------------------------------------------------------------------ one minibatch: losses + all gradients
    def _losses_and_grads(self, policy_state, policy_action, expert_state, expert_action, alpha, lambda_=10.0,
                          policy_idx=None):
        """gail.py:67-93 for one minibatch: returns a device tensor [gail_loss, grad_pen] and leaves
        d(gail_loss + grad_pen)/d(parameters) in self.flat_grads.  With `policy_idx` (int64 device tensor [B]) the policy
        rows are rows policy_idx of the two [rows, D] arrays: the minibatch gather runs inside the first kernel."""
        L = _hip.lib()
        dev = self.flat_params.device
        st = _hip.cur_stream(dev)
        W1, b1, W2, b2, w3, b3 = list(self.trunk.parameters())
        Ds, Da = policy_state.shape[1], policy_action.shape[1]
        B = policy_state.shape[0] if policy_idx is None else policy_idx.numel()
        D, H = Ds + Da, W1.shape[0]
        c = lambda x_: x_.to(device=dev, dtype=f32).contiguous()
        ps, pa, es, ea, al = c(policy_state), c(policy_action), c(expert_state), c(expert_action), c(alpha).view(-1)
        assert es.shape == (B, Ds) and ea.shape == (B, Da) and al.numel() == B, 'expert / policy batch shapes differ'
        e = lambda *s: torch.empty(*s, dtype=f32, device=dev)
        # Row layout of every [4B, .] array: [0, B) policy, [B, 2B) expert, [2B, 3B) mixup, [3B, 4B) mixup again.  The
        # trunk runs over all 4B rows (these launches are latency-bound, the repeated rows are free); afterwards rows
        # [2B, 3B) of X / h1 / h2 are overwritten by the double backward's r / d v1 / t2'*d u2 and rows [3B, 4B) keep the
        # mixup activations, so that each layer's weight gradient -- ordinary backward + the double backward's outer
        # product + the penalty's backward (gail.py:93 on gail_loss + grad_pen) -- is ONE product over the 4B stacked rows.
        m, t_ = slice(2 * B, 3 * B), slice(3 * B, 4 * B)
        X = e(4 * B, D)
        _hip.check(L.acktr_gail_mix_concat_idx_f32(_hip.dev_ptr(ps), _hip.dev_ptr(pa), _hip.dev_ptr(policy_idx, torch.int64),
                                                   _hip.dev_ptr(es), _hip.dev_ptr(ea), _hip.dev_ptr(al), B, Ds, Da, 2,
                                                   _hip.dev_ptr(X), st), 'acktr_gail_mix_concat_idx_f32')
        h1, h2, d = self._forward(X)
        seed, seed_bce, loss = e(3 * B, 1), e(4 * B, 1), e(2)
        _hip.check(L.acktr_gail_bce_seed_f32(_hip.dev_ptr(d), B, 1, _hip.dev_ptr(seed), _hip.dev_ptr(seed_bce),
                                             _hip.dev_ptr(loss), st), 'acktr_gail_bce_seed_f32')
        # first backward, rows [0, 3B): policy / expert rows carry the BCE seeds, the mixup rows carry `ones` (gail.py:45-52)
        a2, a1 = e(4 * B, H), e(4 * B, H)
        ops.linear_dgrad(seed, w3, gate_src=h2[:3 * B], gate='tanh', out=a2[:3 * B])   # rows<2B: d a2 (BCE); mixup: u2 = t2' * w3
        v = ops.linear_dgrad(a2[:3 * B], W2)                            # rows<2B: d h1 (BCE); mixup rows: v1
        link = lambda mode, h, i0, i1, w, o0, o1: _hip.check(L.acktr_gail_tanh_link_f32(
            mode, _hip.dev_ptr(h), _hip.dev_ptr(i0), _hip.dev_ptr(i1), _hip.dev_ptr(w), i0.numel(), H, _hip.dev_ptr(o0),
            _hip.dev_ptr(o1), st), 'acktr_gail_tanh_link_f32')
        link(0, h1[:3 * B], v, None, None, a1[:3 * B], None)            # rows<2B: d a1 (BCE); mixup rows: u1 = t1' * v1
        v1, u1, h1m, h2m = v[m], a1[m], h1[t_], h2[t_]
        g = ops.linear_dgrad(u1, W1)                                    # d D / d x at the mixup points  [B, D]
        r = X[m]                                                        # the mixup inputs live on in X[3B:4B]
        _hip.check(L.acktr_gail_pen_seed_f32(_hip.dev_ptr(g), B, D, float(lambda_), _hip.dev_ptr(r),
                                             _hip.dev_ptr(e(B)) if B > 4096 else None, _hip.dev_ptr(loss), st),
                   'acktr_gail_pen_seed_f32')
        # double backward (autograd of gail.py:46-54 through create_graph=True), mixup rows only
        du1 = ops.linear_fwd(r, W1, None, None)                         # d u1 = r W1^T
        dv1, dt1 = h1[m], e(B, H)
        link(1, h1m, du1, v1, None, dv1, dt1)
        du2 = ops.linear_fwd(dv1, W2, None, None)                       # d u2 = d v1 W2^T
        prod, da2p = h2[m], a2[t_]
        link(2, h2m, du2, None, w3.view(-1), prod, da2p)                # t2' * d u2 (-> d w3) and d a2 of the penalty
        dh1p = ops.linear_dgrad(da2p, W2)
        link(3, h1m, dh1p, dt1, None, a1[t_], None)                     # d a1 of the penalty
        # parameter gradients: per layer  [BCE rows; u; d a (penalty)]^T [inputs; double-backward operand; mixup inputs];
        # the u rows [2B, 3B) take no part in the bias sums
        gW1, gb1, gW2, gb2, gw3, gb3 = [self._gview(i) for i in range(6)]
        _hip.check(L.acktr_gail_param_grads_f32(
            _hip.dev_ptr(a1), _hip.dev_ptr(X), _hip.dev_ptr(a2), _hip.dev_ptr(h1), _hip.dev_ptr(seed_bce), _hip.dev_ptr(h2),
            4 * B, D, H, 2 * B, 3 * B, _hip.dev_ptr(gW1), _hip.dev_ptr(gb1), _hip.dev_ptr(gW2), _hip.dev_ptr(gb2),
            _hip.dev_ptr(gw3), _hip.dev_ptr(gb3), st), 'acktr_gail_param_grads_f32')
        return loss

    # ------------------------------------------------------------------ one optimiser step, replayed from a hipGraph
    def _step(self, policy_state, policy_action, expert_state, expert_action, alpha, policy_idx=None):
        """losses + gradients + Adam for one minibatch (gail.py:67-93).  The 20 launches of a step are latency-bound,
        so from the third step of a given batch shape on they are replayed from a captured hipGraph whose inputs are
        persistent buffers refilled before every replay (ACKTR_HIP_GRAPH=0 disables it).  With `policy_idx` the policy
        operands are the whole [rows, D] rollout arrays (read in place) and only the index batch is refilled."""
        dev = self.flat_params.device
        call = lambda ps, pa, es, ea, al, ix: self._losses_and_grads(ps, pa, es, ea, al, policy_idx=ix)
        if os.environ.get('ACKTR_HIP_GRAPH', '1') == '0' or ops.profiler is not None:
            loss = call(policy_state, policy_action, expert_state, expert_action, alpha, policy_idx)
            self.optimizer.step(max_grad_norm=None)                     # flat Adam; grads were just overwritten
            return loss
        Ds, Da = policy_state.shape[1], policy_action.shape[1]
        B = policy_state.shape[0] if policy_idx is None else policy_idx.numel()
        key = (B, Ds, Da, self.flat_params.data_ptr(),
               None if policy_idx is None else (policy_state.data_ptr(), policy_action.data_ptr(), policy_state.shape[0]))
        st = self.__dict__.get('_gstate')
        if st is None or st['key'] != key:
            e = lambda *s: torch.empty(*s, dtype=f32, device=dev)
            if policy_idx is None:
                inp = [e(B, Ds), e(B, Da), e(B, Ds), e(B, Da), e(B, 1), None]
            else:                                                       # the rollout arrays themselves + an index buffer
                inp = [policy_state, policy_action, e(B, Ds), e(B, Da), e(B, 1), torch.zeros(B, dtype=torch.int64, device=dev)]
            st = dict(key=key, seen=0, graph=None, loss=None, inp=inp)
            self.__dict__['_gstate'] = st
        srcs = (policy_state, policy_action, expert_state, expert_action, alpha, policy_idx)
        for i, (dst, src) in enumerate(zip(st['inp'], srcs)):
            if dst is None or (policy_idx is not None and i < 2):
                continue
            dst.copy_(src.reshape(dst.shape), non_blocking=True)
        if st['graph'] is None:
            if st['seen'] < 2:                                          # warm workspaces / optimiser state launch by launch
                st['seen'] += 1
                loss = call(*st['inp'])
                self.optimizer.step(max_grad_norm=None)
                return loss
            self.optimizer.prepare()
            g = torch.cuda.CUDAGraph()
            with ops.graph_capture(g):
                st['loss'] = call(*st['inp'])
                self.optimizer.step(max_grad_norm=None, capturing=True)
                self.optimizer.step_count -= 1
            st['graph'] = g
        self.optimizer.prepare()                                        # host-side: lr / betas changes reach the device
        st['graph'].replay()
        self.optimizer.step_count += 1
        return st['loss'].clone()

    # ------------------------------------------------------------------ reference API
    def compute_grad_pen(self, expert_state, expert_action, policy_state, policy_action, lambda_=10):
        """gail.py:29-55: the penalty value (a 0-d device tensor; no autograd graph -- use `update`)."""
        self._check_dev()
        alpha = torch.rand(expert_state.size(0), 1)                     # CPU default generator, like the reference
        loss = self._losses_and_grads(policy_state, policy_action, expert_state, expert_action, alpha, lambda_)
        return loss[1]

    def update(self, expert_loader, rollouts, obsfilt=None):
        """gail.py:57-95"""
        self._check_dev()
        self.train()
        # gail.py:60 draws its policy batches from rollouts.feed_forward_generator(None, mini_batch_size=...).  Of the
        # eight gathered tensors of a batch only the states and actions are used, so for this package's RolloutStorage
        # the same index batches (same ONE randperm) are taken directly and the gather is fused into the first kernel.
        light = (hasattr(rollouts, 'minibatch_index_batches') and rollouts.obs.dim() == 3 and rollouts.obs.is_cuda
                 and rollouts.obs.dtype == f32 and rollouts.actions.dtype == f32)
        losses = []
        if light:
            T = rollouts.rewards.size(0)
            obs_rows = rollouts.obs[:T].reshape(-1, rollouts.obs.size(-1))
            act_rows = rollouts.actions.reshape(-1, rollouts.actions.size(-1))

            def batches():                                              # a generator, so that the permutation is drawn at the
                for ix in rollouts.minibatch_index_batches(mini_batch_size=expert_loader.batch_size):
                    yield ix                                            # first next(), after the expert loader's own draws
            batches = batches()
        else:
            batches = rollouts.feed_forward_generator(None, mini_batch_size=expert_loader.batch_size)
        for expert_batch, policy_batch in zip(_expert_batches(expert_loader), batches):
            expert_state, expert_action = expert_batch
            expert_state = obsfilt(expert_state.numpy(), update=False)
            expert_state = torch.FloatTensor(expert_state).to(self.device)
            expert_action = expert_action.to(self.device)
            alpha = torch.rand(expert_state.size(0), 1)                 # gail.py:35 (CPU RNG stream, F7)
            if light:
                losses.append(self._step(obs_rows, act_rows, expert_state, expert_action, alpha,
                                         policy_idx=policy_batch.contiguous()))
            else:
                losses.append(self._step(policy_batch[0], policy_batch[2], expert_state, expert_action, alpha))
        tot = torch.stack(losses).sum(dim=1).cpu()                      # ONE host sync per update (the reference: one per batch)
        return float(tot.double().sum().item()) / len(losses)

    def _reward(self, state, action):
        L = _hip.lib()
        dev = self.flat_params.device
        x = torch.cat([state.to(dev, f32), action.to(dev, f32)], dim=1).contiguous()
        _, _, d = self._forward(x)
        reward = torch.empty_like(d)
        _hip.check(L.acktr_gail_reward_f32(_hip.dev_ptr(d), d.numel(), _hip.dev_ptr(reward), _hip.cur_stream(dev)),
                   'acktr_gail_reward_f32')
        return reward

    def predict_reward(self, state, action, gamma, masks, update_rms=True):
        """gail.py:97-110"""
        self._check_dev()
        with torch.no_grad():
            self.eval()
            reward = self._reward(state, action)
            return self._normalise(reward.view(1, -1), masks.view(1, -1), gamma, update_rms).view(-1, 1)

    def predict_reward_batched(self, states, actions, gamma, masks, update_rms=True):
        """All steps of a rollout at once: states [T,N,Ds], actions [T,N,Da], masks [T,N,1] -> rewards [T,N,1], with
        exactly the per-step semantics of T consecutive `predict_reward(states[t], actions[t], gamma, masks[t])` calls
        (running return and RunningMeanStd updated step by step, step t scaled by the variance after step t)."""
        self._check_dev()
        with torch.no_grad():
            self.eval()
            T, N = states.shape[0], states.shape[1]
            reward = self._reward(states.reshape(T * N, -1), actions.reshape(T * N, -1)).view(T, N)
            return self._normalise(reward, masks.reshape(T, N), gamma, update_rms).view(T, N, 1)

    def _normalise(self, reward, masks, gamma, update_rms):
        L = _hip.lib()
        dev = reward.device
        S, N = reward.shape
        if self.returns is None:
            self.returns = reward[0].clone().view(-1, 1)
        if not update_rms:
            return reward / float(np.sqrt(self.ret_rms.var[0] + 1e-8))
        masks = masks.to(dev, f32).contiguous()
        ret_steps = torch.empty(S, N, dtype=f32, device=dev)
        mom = torch.empty(S, 2, dtype=torch.float64, device=dev)
        ret_io = self.returns.view(-1).contiguous()
        _hip.check(L.acktr_gail_returns_moments_f32(_hip.dev_ptr(reward), _hip.dev_ptr(masks), S, N, float(gamma),
                                                    _hip.dev_ptr(ret_io), _hip.dev_ptr(ret_steps),
                                                    _hip.dev_ptr(mom, torch.float64), _hip.cur_stream(dev)),
                   'acktr_gail_returns_moments_f32')
        self.returns = ret_io.view(-1, 1)
        # RunningMeanStd merge of the S per-step moments on the device (one lane, fp64, the numpy restatement's operation
        # order) -- round 1 pulled the moments to the host and merged them in a Python loop, half of the 14 ms of a
        # 2048-step rollout; the public `ret_rms` object is refreshed from the 24-byte state afterwards
        first = lambda a: float(np.asarray(a, np.float64).reshape(-1)[0])
        state = torch.tensor([first(self.ret_rms.mean), first(self.ret_rms.var), first(self.ret_rms.count)],
                             dtype=torch.float64).to(dev)
        sc = torch.empty(S, dtype=f32, device=dev)
        _hip.check(L.acktr_gail_rms_merge_f64(_hip.dev_ptr(mom, torch.float64), S, float(N),
                                              _hip.dev_ptr(state, torch.float64), _hip.dev_ptr(sc), _hip.cur_stream(dev)),
                   'acktr_gail_rms_merge_f64')
        st_h = state.cpu().numpy()
        # (shape (1,) like the reference's after `ret_rms.update(returns[N,1])`, gail.py:107-110 reads `var[0]`)
        self.ret_rms.mean, self.ret_rms.var, self.ret_rms.count = st_h[0:1].copy(), st_h[1:2].copy(), float(st_h[2])
        out = reward.clone()
        _hip.check(L.acktr_gail_scale_rows_f32(_hip.dev_ptr(out), _hip.dev_ptr(sc), S, N, _hip.cur_stream(dev)),
                   'acktr_gail_scale_rows_f32')
        return out


def _expert_batches(loader):
    """Batches of `loader` -- the same rows in the same order from the same RNG draws.  For the loader main.py builds
    (a plain single-process DataLoader over an ExpertDataset with the default collate, gail.py:57-66 / main.py:78-83)
    the index batches are taken from the loader's own iterator and the rows come from ONE index_select on the
    dataset's flat tensors, instead of `batch_size` __getitem__ calls and a default_collate per batch (400 us of the
    ~1.1 ms a discriminator step took at batch 128).  Anything else is iterated as it is."""
    from torch.utils.data import dataloader as _dl
    ds = getattr(loader, 'dataset', None)
    plain = (type(loader) is torch.utils.data.DataLoader and type(ds) is ExpertDataset and loader.num_workers == 0
             and loader.collate_fn is _dl.default_collate and not loader.pin_memory and loader.batch_size is not None
             and hasattr(_dl._BaseDataLoaderIter, '_next_index'))
    if not plain:
        for b in loader:
            yield b
        return
    it = iter(loader)                       # the real iterator: its constructor and sampler draw from the CPU generator
    while True:
        try:
            index = it._next_index()
        except StopIteration:
            return
        yield ds.rows(index)


class ExpertDataset(torch.utils.data.Dataset):
    """gail.py:113-166 -- the gail_experts on-disk format (gail_experts/convert_to_pytorch.py:28-48): a `torch.save`d
    dict {'states' [B,T,Do] f32, 'actions' [B,T,Da] f32, 'rewards' [B,T] f32, 'lengths' [B] i64}."""

    def __init__(self, file_name, num_trajectories=4, subsample_frequency=20):
        all_trajectories = torch.load(file_name, weights_only=True)    # plain tensors only: nothing is executed from the file
        for k in ('states', 'actions', 'rewards', 'lengths'):
            if k not in all_trajectories:
                raise KeyError('expert file %r lacks %r (gail_experts format)' % (file_name, k))
        perm = torch.randperm(all_trajectories['states'].size(0))
        idx = perm[:num_trajectories]
        self.trajectories = {}
        start_idx = torch.randint(0, subsample_frequency, size=(num_trajectories,)).long()
        for k, v in all_trajectories.items():
            data = v[idx]
            if k != 'lengths':
                self.trajectories[k] = torch.stack(
                    [data[i, start_idx[i]::subsample_frequency] for i in range(num_trajectories)])
            else:
                self.trajectories[k] = data // subsample_frequency
        self.i2traj_idx = {}
        self.i2i = {}
        self.length = self.trajectories['lengths'].sum().item()
        lengths = [int(x) for x in self.trajectories['lengths']]
        traj_idx, i = 0, 0
        self.get_idx = []
        for _ in range(self.length):
            while lengths[traj_idx] <= i:
                i -= lengths[traj_idx]
                traj_idx += 1
            self.get_idx.append((traj_idx, i))
            i += 1

        # flat [length, D] copies of the rows __getitem__ returns, for whole-batch fetches (`rows`)
        ti = torch.tensor([a for a, _ in self.get_idx], dtype=torch.int64)
        ii = torch.tensor([b for _, b in self.get_idx], dtype=torch.int64)
        self._flat = (self.trajectories['states'][ti, ii].contiguous(), self.trajectories['actions'][ti, ii].contiguous())

    def rows(self, index):
        """what default_collate([self[i] for i in index]) returns, as two index_selects"""
        ix = torch.as_tensor(index, dtype=torch.int64)
        return [self._flat[0].index_select(0, ix), self._flat[1].index_select(0, ix)]

    def __len__(self):
        return self.length

    def __getitem__(self, i):
        traj_idx, i = self.get_idx[i]
        return self.trajectories['states'][traj_idx][i], self.trajectories['actions'][traj_idx][i]
