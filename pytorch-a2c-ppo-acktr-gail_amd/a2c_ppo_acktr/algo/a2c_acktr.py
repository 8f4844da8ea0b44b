"""A2C / ACKTR -- host-side mirror of reference a2c_ppo_acktr/algo/a2c_acktr.py:8-80.

A2C: one full-batch HIP forward, the fused A2C loss kernel, HIP backward, clip-norm + RMSprop in two
launches.  ACKTR adds the sampled-Fisher backward whose layer inputs / grad-outputs feed the KFAC
factor kernels, then `KFACOptimizer.step` (algo/kfac.py).
"""
import os

import torch

from .. import _hip, engine, ops
from ..parallel import DataParallel, PiecewiseGraph


class _FlatRMSprop(object):
    """optim.RMSprop (alpha, eps; not centred, no momentum) over the flat buffer (torch/optim/rmsprop.py:316-339)."""

    def __init__(self, policy, lr, eps, alpha):
        self.policy = policy
        self.param_groups = [{'params': list(policy.parameters()), 'lr': lr, 'eps': eps, 'alpha': alpha}]
        self.square_avg = None

    def zero_grad(self, set_to_none=False):
        self.policy.flat_grads.zero_()

    def step(self, max_grad_norm=None):
        flat, gflat = self.policy.flat_params, self.policy.flat_grads
        if self.square_avg is None or self.square_avg.device != flat.device or self.square_avg.numel() != flat.numel():
            self.square_avg = torch.zeros_like(flat)
        g = self.param_groups[0]
        L = _hip.lib()
        ws = ops.workspace(flat.device, L.acktr_optim_workspace_bytes(), 'optim')
        rc = L.acktr_clip_rmsprop_f32(_hip.dev_ptr(flat), _hip.dev_ptr(gflat), _hip.dev_ptr(self.square_avg),
                                      flat.numel(), float(max_grad_norm) if max_grad_norm else 0.0, g['lr'],
                                      g['alpha'], g['eps'], None, _hip.dev_ptr(ws), _hip.cur_stream(flat.device))
        _hip.check(rc, 'acktr_clip_rmsprop_f32')


class A2C_ACKTR():
    def __init__(self,
                 actor_critic,
                 value_loss_coef,
                 entropy_coef,
                 lr=None,
                 eps=None,
                 alpha=None,
                 max_grad_norm=None,
                 acktr=False):

        self.actor_critic = actor_critic
        self.acktr = acktr

        self.value_loss_coef = value_loss_coef
        self.entropy_coef = entropy_coef

        self.max_grad_norm = max_grad_norm

        if acktr:
            from .kfac import KFACOptimizer
            self.optimizer = KFACOptimizer(actor_critic)
        else:
            self.optimizer = _FlatRMSprop(actor_critic, lr, eps=eps, alpha=alpha)
        self.dp = DataParallel()
        if acktr:
            self.optimizer.dp = self.dp
        # hipGraph replay of the whole A2C update (launch-bound at the reference's 80-sample batch): captured at the
        # second update() that sees the same storage, shapes and hyper-parameters; ACKTR_HIP_GRAPH=0 disables it.
        self.use_graph = os.environ.get('ACKTR_HIP_GRAPH', '1') != '0'
        self._graph, self._seen_key, self._loss_acc = None, None, None
        self._kgraphs, self._noise_dev = {}, None

    def _capture(self, rollouts, dev):
        """One replayable object for _update(): a hipGraph, or under data parallelism graph pieces around the collectives
        (parallel.PiecewiseGraph; the loss accumulator's all-reduce follows the last piece)."""
        if self.dp.active:
            pg = PiecewiseGraph(dev).capture(self.dp, lambda: self._update(rollouts, self._loss_acc))
            pg.items.append(lambda: self.dp.all_reduce_sum_(self._loss_acc))
            return pg
        g = torch.cuda.CUDAGraph()
        with ops.graph_capture(g):
            self._loss_acc.zero_()                  # (a node of the graph: no launch of its own per update)
            self._update(rollouts, self._loss_acc)
        return g

    def _update_acktr_graphed(self, rollouts):
        """ACKTR update with the launch-bound part replayed from a hipGraph: everything except the steps that
        eigendecompose (rocSOLVER's host-driven solver cannot be captured) and the first two (one-off initialisation
        flags).  The Fisher noise keeps coming from the CPU generator (F7): it is drawn before every replay into the
        buffer the captured loss kernel reads."""
        ac, opt = self.actor_critic, self.optimizer
        T, N = rollouts.rewards.size()[0:2]
        dev = rollouts.rewards.device
        key = tuple(getattr(rollouts, k).data_ptr() for k in ('obs', 'actions', 'returns')) + \
            (T, N, ac.flat_params.data_ptr(), self.value_loss_coef, self.entropy_coef, opt.lr, opt.Ts, opt.Tf,
             opt.damping, opt.kl_clip, opt.stat_decay, opt.momentum, opt.weight_decay, opt.fast_cnn)
        if self._seen_key != key:
            self._seen_key, self._kgraphs = key, {}
            self._loss_acc = torch.zeros(3, dtype=torch.float32, device=dev)
            self._noise_dev = torch.zeros(T * N, 1, dtype=torch.float32, device=dev)
            return self._update(rollouts, None)
        do_stats = opt.steps % opt.Ts == 0
        if opt.steps < 2 or opt.steps % opt.Tf == 0 or opt.weight_decay > 0:
            return self._update(rollouts, None)
        g = self._kgraphs.get(do_stats)
        if g is None:
            steps0 = opt.steps
            opt._ext_noise = self._noise_dev
            try:
                g = self._capture(rollouts, dev)
            finally:
                opt._ext_noise = None
                opt.steps = steps0                  # capture only records: the replay below is this update
            self._kgraphs[do_stats] = g
        if do_stats:
            self._noise_dev.copy_(torch.randn(T * N, 1))       # the draw accumulate_stats makes (a2c_acktr.py:59)
        if self.dp.active:                          # (single process: the graph zeroes it itself, _capture)
            self._loss_acc.zero_()
        g.replay()
        opt.steps += 1
        vl, al, en = self._loss_acc.tolist()
        return vl, al, en

    def update(self, rollouts):
        ac = self.actor_critic
        if not self.use_graph or ops.profiler is not None or (self.dp.active and ac.is_recurrent):
            return self._update(rollouts, None)
        if self.acktr:
            return self._update_acktr_graphed(rollouts)
        g = self.optimizer.param_groups[0]
        key = tuple(getattr(rollouts, k).data_ptr() for k in ('obs', 'actions', 'returns', 'masks',
                                                                'recurrent_hidden_states')) + \
            (tuple(rollouts.rewards.size()[0:2]), ac.flat_params.data_ptr(), self.value_loss_coef, self.entropy_coef,
             self.max_grad_norm, g['lr'], g['alpha'], g['eps'])
        if self._seen_key != key:                   # first sight of this configuration: run it launch by launch
            self._seen_key, self._graph = key, None
            self._loss_acc = torch.zeros(3, dtype=torch.float32, device=rollouts.rewards.device)
            return self._update(rollouts, None)
        if self._graph is None:
            self._graph = self._capture(rollouts, rollouts.rewards.device)
        if self.dp.active:
            self._loss_acc.zero_()
        self._graph.replay()
        vl, al, en = self._loss_acc.tolist()
        return vl, al, en

    def _update(self, rollouts, captured_loss_acc):
        ac = self.actor_critic
        T, N = rollouts.rewards.size()[0:2]
        rows = T * N
        dev = rollouts.rewards.device
        with torch.no_grad():
            capturing = captured_loss_acc is not None
            loss_acc = captured_loss_acc if capturing else torch.zeros(3, dtype=torch.float32, device=dev)
            obs_flat = rollouts.obs[:T].view(rows, *rollouts.obs.size()[2:])
            actions = rollouts.actions.view(rows, -1)
            returns = rollouts.returns[:T].view(rows)
            ac.bind_grads()
            inv_batch = 1.0 / rows
            idx = hxs = masks = None
            if ac.is_recurrent:                    # a2c_acktr.py:38-43: all N trajectories, states stored at step 0
                if self.acktr:
                    raise NotImplementedError('Recurrent policy is not implemented for ACKTR')   # arguments.py:157-159
                H = ac.recurrent_hidden_state_size
                hxs = rollouts.recurrent_hidden_states[0].view(-1, H)
                masks = rollouts.masks[:T].reshape(rows, 1)
                if self.dp.active:                 # shard by env: a trajectory cannot be split
                    lo, hi = self.dp.bounds(N)
                    mine = torch.arange(lo, hi, device=dev)
                    idx = rollouts.recurrent_rows(mine)
                    hxs = hxs[lo:hi]
                    masks = rollouts._gather(masks, idx, rows)
            elif self.dp.active:                   # shard the full batch over ranks (rows are independent)
                lo, hi = self.dp.bounds(rows)
                idx = torch.arange(lo, hi, device=dev)
            nloc = rows if idx is None else idx.numel()
            if not self.acktr and hxs is None and engine.mlp_fused_ok(ac, nloc):
                engine.mlp_fused_step(ac, obs_flat, idx, nloc, engine.ALGO_A2C, actions, loss_acc, inv_batch,
                                      returns=returns, value_coef=self.value_loss_coef,
                                      entropy_coef=self.entropy_coef)
            else:
                # A2C on the NatureCNN: head, loss and the head's backward in one launch (ACKTR keeps the layer-wise
                # path: its statistics pass reads the head activations)
                fuse_head = not self.acktr and hxs is None and engine.head_step_ok(ac, nloc)
                acts = engine.forward(ac, obs_flat, gather_idx=idx, rnn_hxs=hxs, masks=masks,
                                      head=not fuse_head)                                    # a2c_acktr.py:38-43
                if self.acktr and self.optimizer.steps % self.optimizer.Ts == 0:
                    self.optimizer.accumulate_stats(acts, actions, idx, inv_batch, self.dp)  # a2c_acktr.py:53-68
                (engine.head_step if fuse_head else engine.policy_loss)(
                    ac, acts, engine.ALGO_A2C, actions, loss_acc, inv_batch, returns=returns, gather_idx=idx,
                    value_coef=self.value_loss_coef, entropy_coef=self.entropy_coef)        # a2c_acktr.py:48-51,70-72
                engine.backward(ac, acts, ac.flat_grads)
            self.dp.all_reduce_sum_(ac.flat_grads)
            if self.acktr == False:
                self.optimizer.step(self.max_grad_norm)                                 # a2c_acktr.py:74-78
            else:
                self.optimizer.step()
            if capturing:
                return None
            self.dp.all_reduce_sum_(loss_acc)
            vl, al, en = loss_acc.tolist()
        return vl, al, en
