"""PPO -- host-side mirror of reference a2c_ppo_acktr/algo/ppo.py:7-96.

Same constructor and `update(rollouts) -> (value_loss, action_loss, dist_entropy)` contract.  One
minibatch step is: HIP forward (minibatch gather fused into the first layer) -> fused
categorical/gaussian + clipped-surrogate loss kernel that also emits the backward seeds -> HIP backward
into the flat gradient bucket -> [RCCL all-reduce] -> clip-norm + Adam in two launches.  Nothing syncs
with the host until the three averaged losses are read back once at the end of `update`.
"""
import ctypes
import os

import torch

from .. import _hip, engine, ops
from ..parallel import DataParallel, bounded_sync, quiesce_collectives

# ACKTR_DEFER_HEAD=0: the head step finalizes in its own launch (comparison switch)
DEFER_HEAD_FINALIZE = os.environ.get('ACKTR_DEFER_HEAD', '1') != '0'


class _FlatAdam(object):
    """optim.Adam over the policy's flat buffer (torch/optim/adam.py:529-547 single-tensor math).
    Exposes `param_groups` so update_linear_schedule (utils.py:46-50) works as in the reference.
    The step counter and {lr, betas, eps} live in device memory, so the same launch sequence serves eager
    calls and hipGraph replays."""

    def __init__(self, policy, lr, eps, betas=(0.9, 0.999)):
        self.policy = policy
        self.param_groups = [{'params': list(policy.parameters()), 'lr': lr, 'eps': eps, 'betas': betas}]
        self.step_count = 0
        self.exp_avg = None
        self.exp_avg_sq = None
        self._hyper_host = None

    def _state(self):
        flat = self.policy.flat_params
        if self.exp_avg is None or self.exp_avg.device != flat.device or self.exp_avg.numel() != flat.numel():
            self.exp_avg = torch.zeros_like(flat)
            self.exp_avg_sq = torch.zeros_like(flat)
            self.step_dev = torch.full((1,), self.step_count, dtype=torch.int32, device=flat.device)
            self.hyper_dev = torch.zeros(4, dtype=torch.float32, device=flat.device)
            self._hyper_host = None
        g = self.param_groups[0]
        hyper = (float(g['lr']), float(g['betas'][0]), float(g['betas'][1]), float(g['eps']))
        if hyper != self._hyper_host:                      # rare: construction and lr-schedule changes
            self.hyper_dev.copy_(torch.tensor(hyper, dtype=torch.float32))
            self._hyper_host = hyper

    def zero_grad(self, set_to_none=False):
        self.policy.flat_grads.zero_()

    def prepare(self):
        """Host-side refresh of the device hyper-parameters; call outside graph capture."""
        self._state()

    def step(self, max_grad_norm=None, capturing=False, pending=None, slab=None, image_map=None):
        """clip_grad_norm_(max_grad_norm) (if given) + Adam step on the flat buffers.  `pending`: a head step's deferred
        finalize (engine.head_step(defer_finalize=True)), `slab`: the last weight gradient's deferred slab sum
        (engine.backward(defer_last=...)); both are carried by the norm launch.  `image_map` (engine.refresh_conv_images):
        the new conv weights are also stored to their slots in the stationary kernels' weight images."""
        if not capturing:
            self._state()
        self.step_count += 1
        flat, gflat = self.policy.flat_params, self.policy.flat_grads
        L = _hip.lib()
        ws = ops.workspace(flat.device, L.acktr_optim_workspace_bytes(), 'optim')
        if pending is not None or slab is not None or (image_map is not None and max_grad_norm):
            rc = L.acktr_clip_adam_dev_pack_f32(_hip.dev_ptr(flat), _hip.dev_ptr(gflat), _hip.dev_ptr(self.exp_avg),
                                                _hip.dev_ptr(self.exp_avg_sq), flat.numel(), float(max_grad_norm),
                                                _hip.dev_ptr(self.hyper_dev), _hip.dev_ptr(self.step_dev, torch.int32),
                                                ctypes.byref(pending) if pending is not None else None,
                                                ctypes.byref(slab) if slab is not None else None,
                                                ctypes.byref(image_map) if image_map is not None else None,
                                                _hip.dev_ptr(ws), _hip.cur_stream(flat.device))
            _hip.check(rc, 'acktr_clip_adam_dev_pack_f32')
            return
        rc = L.acktr_clip_adam_dev_f32(_hip.dev_ptr(flat), _hip.dev_ptr(gflat), _hip.dev_ptr(self.exp_avg),
                                       _hip.dev_ptr(self.exp_avg_sq), flat.numel(),
                                       float(max_grad_norm) if max_grad_norm else 0.0, _hip.dev_ptr(self.hyper_dev),
                                       _hip.dev_ptr(self.step_dev, torch.int32), _hip.dev_ptr(ws),
                                       _hip.cur_stream(flat.device))
        _hip.check(rc, 'acktr_clip_adam_dev_f32')


class PPO():
    def __init__(self,
                 actor_critic,
                 clip_param,
                 ppo_epoch,
                 num_mini_batch,
                 value_loss_coef,
                 entropy_coef,
                 lr=None,
                 eps=None,
                 max_grad_norm=None,
                 use_clipped_value_loss=True):

        self.actor_critic = actor_critic

        self.clip_param = clip_param
        self.ppo_epoch = ppo_epoch
        self.num_mini_batch = num_mini_batch

        self.value_loss_coef = value_loss_coef
        self.entropy_coef = entropy_coef

        self.max_grad_norm = max_grad_norm
        self.use_clipped_value_loss = use_clipped_value_loss

        self.optimizer = _FlatAdam(actor_critic, lr=lr, eps=eps)
        self.dp = DataParallel()
        # hipGraph replay of the minibatch step (launch-bound otherwise): captured at the second update() that
        # sees the same storage / shapes; ACKTR_HIP_GRAPH=0 disables it.
        self.use_graph = os.environ.get('ACKTR_HIP_GRAPH', '1') != '0'
        self.dp_overlap = os.environ.get('ACKTR_DP_OVERLAP', '1') != '0'
        self._graph_key = None
        self._graphs = None
        self._seen_key = None
        self._full_checked = False
        self._image_map = None

    # one minibatch step = ppo.py:57-84
    def _fwd_bwd(self, b, phase=None, with_optimizer=False):
        """phase None: forward, loss and the whole backward.  Returns True when the optimiser step ran too
        (`with_optimizer` and the one-launch MLP step: clip + Adam are the tail of that launch).  Data-parallel NatureCNN runs split it in two so that the
        all-reduce of the late gradient bucket (conv3, FC, head: 97 % of the bytes) overlaps with the rest of the
        backward: phase 1 = forward, loss, backward down to conv3's weights; phase 2 = the remaining backward."""
        ac = self.actor_critic
        if phase == 2:
            engine.backward(ac, self._acts, ac.flat_grads, phase=2)
            return
        nrows = b['idx'].numel()
        if phase is None and b.get('hxs') is None and engine.mlp_fused_ok(ac, nrows):
            engine.mlp_fused_step(ac, b['obs'], b['idx'], nrows, engine.ALGO_PPO, b['actions'], b['loss_acc'],
                                  b['inv_batch'], old_logp=b['old_logp'], old_values=b['old_values'],
                                  returns=b['returns'], adv=b['adv'], clip=self.clip_param,
                                  value_coef=self.value_loss_coef, entropy_coef=self.entropy_coef,
                                  use_clipped_value_loss=self.use_clipped_value_loss,
                                  adam=self.optimizer if with_optimizer else None, max_grad_norm=self.max_grad_norm)
            return with_optimizer
        fuse_head = b.get('hxs') is None and engine.head_step_ok(ac, nrows)
        acts = engine.forward(ac, b['obs'], gather_idx=b['idx'], rnn_hxs=b.get('hxs'), masks=b.get('masks'),
                              head=not fuse_head, trust_images=self._image_map is not None)
        # single process, clipped: the head's slab sums ride in the optimiser's norm launch (one launch fewer per step)
        defer = bool(fuse_head and with_optimizer and self.max_grad_norm and DEFER_HEAD_FINALIZE)
        ret = (engine.head_step if fuse_head else engine.policy_loss)(
            ac, acts, engine.ALGO_PPO, b['actions'], b['loss_acc'], b['inv_batch'],
            old_logp=b['old_logp'], old_values=b['old_values'], returns=b['returns'], adv=b['adv'],
            gather_idx=b['idx'], clip=self.clip_param, value_coef=self.value_loss_coef,
            entropy_coef=self.entropy_coef, use_clipped_value_loss=self.use_clipped_value_loss,
            **(dict(defer_finalize=True) if defer else {}))
        self._pending = ret if defer else None
        self._pending_slab = _hip.SlabFin() if defer else None
        engine.backward(ac, acts, ac.flat_grads, phase=phase, defer_last=self._pending_slab)
        self._acts = acts if phase == 1 else None
        return False

    def _step(self, b, capturing=False):
        """one single-process minibatch step: forward, loss, backward, clip + Adam (ppo.py:57-84)"""
        self._pending = self._pending_slab = None
        if self._fwd_bwd(b, with_optimizer=True):
            self.optimizer.step_count += 1
        else:
            self.optimizer.step(self.max_grad_norm, capturing=capturing, pending=self._pending, slab=self._pending_slab,
                                image_map=self._image_map)
            self._pending = self._pending_slab = None

    def _dp_step(self, b, split, graphs, capturing=False):
        """One data-parallel minibatch step: (phase 1) -> start all-reduce of grads[split:] -> (phase 2) -> all-reduce
        of grads[:split] -> optimiser; without a split, one all-reduce of the whole bucket after the backward."""
        ac = self.actor_critic
        run = (lambda i, fn: graphs[i].replay()) if graphs is not None else (lambda i, fn: fn())
        if split is None:
            run(0, lambda: self._fwd_bwd(b))
            self.dp.all_reduce_sum_(ac.flat_grads)
        else:
            run(0, lambda: self._fwd_bwd(b, phase=1))
            late = self.dp.all_reduce_sum_async_(ac.flat_grads[split:])
            run(1, lambda: self._fwd_bwd(b, phase=2))
            self.dp.all_reduce_sum_(ac.flat_grads[:split])
            late.wait()
        if graphs is not None:
            graphs[2].replay()
            self.optimizer.step_count += 1
        else:
            self.optimizer.step(self.max_grad_norm, capturing=capturing, image_map=self._image_map)   # ppo.py:82-84

    def update(self, rollouts):
        ac = self.actor_critic
        if ac.is_recurrent:
            return self._update_recurrent(rollouts)
        T, N = rollouts.rewards.size()[0:2]
        rows = T * N
        dev = rollouts.rewards.device
        mbs = rows // self.num_mini_batch
        # batches the sampler really yields (BatchSampler(drop_last=True), storage.py:122-125): more than
        # num_mini_batch when rows is not a multiple of it (80 rows / 32 -> 40 batches of 2)
        nb = rows // max(mbs, 1)
        lo, hi = self.dp.bounds(mbs)
        # the conv weight images are re-ordered ONCE per update (whatever happened to the parameters since the last one);
        # inside the update the Adam steps keep them current, so the minibatch steps have no pack launch
        with torch.no_grad():
            self._image_map = engine.refresh_conv_images(ac, hi - lo) if self.max_grad_norm else None
        key = (self._image_map.base if self._image_map is not None else None, rollouts.obs.data_ptr(), rollouts.actions.data_ptr(), rollouts.returns.data_ptr(),
               rollouts.value_preds.data_ptr(), rollouts.action_log_probs.data_ptr(), T, N, mbs, lo, hi,
               ac.flat_params.data_ptr(), self.clip_param, self.value_loss_coef, self.entropy_coef,
               self.max_grad_norm, self.use_clipped_value_loss)
        with torch.no_grad():
            if self._seen_key != key:                       # (re)allocate the persistent per-update buffers
                self._adv = torch.empty(T, N, 1, dtype=torch.float32, device=dev)
                self._loss_acc = torch.zeros(3, dtype=torch.float32, device=dev)
                self._idx = torch.zeros(hi - lo, dtype=torch.int64, device=dev)
                self._graphs, self._graph_key = None, None
            rollouts.normalized_advantages(out=self._adv)                       # ppo.py:35-37
            self._loss_acc.zero_()
            b = dict(obs=rollouts.obs[:T].view(rows, *rollouts.obs.size()[2:]), actions=rollouts.actions.view(rows, -1),
                     old_values=rollouts.value_preds[:T].view(rows), returns=rollouts.returns[:T].view(rows),
                     old_logp=rollouts.action_log_probs.view(rows), adv=self._adv.view(rows), idx=self._idx,
                     loss_acc=self._loss_acc, inv_batch=1.0 / mbs)
            ac.bind_grads()
            self.optimizer.prepare()
            graph_ok = self.use_graph and ops.profiler is None and self._seen_key == key
            split = engine.late_bucket_offset(ac) if (self.dp.active and self.dp_overlap) else None
            if split is not None:
                # overlapped buckets run launch by launch: replaying the three graph pieces around the two
                # collectives measured 13.2 ms per update against 11.9 eager (and 11.8 for two graphs around ONE
                # un-overlapped all-reduce, which then pays the whole transfer on the critical path)
                graph_ok = graph_ok and os.environ.get('ACKTR_DP_GRAPH', '0') in ('1', 'full')
            # ACKTR_DP_GRAPH=full (RCCL only, opt-in): the collectives are captured too -- torch records an RCCL all-reduce
            # into the graph as a branch on RCCL's stream (tools/rccl_graph_probe.py) -- so a data-parallel epoch is ONE
            # graph like the single-process one, overlap included.  Rehearsed with one rank only; not the default until a
            # multi-GPU box has replayed it.
            full = (self.dp.active and os.environ.get('ACKTR_DP_GRAPH', '0') == 'full' and not self.dp._host_staged)
            # Single process: ONE graph per epoch; step k reads its rows from slice k of a device buffer that receives the
            # whole epoch's permutation in one copy -- a 512-byte host-to-device copy between
            # every two replays cost 4.4 us of device time plus an exposed graph-launch latency per step (10 % of the
            # 44 us MLP step).  Data-parallel: the all-reduce sits between two graphs that read `self._idx`.
            per_pos = graph_ok and (not self.dp.active or full)
            if graph_ok and self._graphs is None:           # capture (records only; executes nothing)
                if per_pos:
                    self._idx_all = torch.zeros(nb, hi - lo, dtype=torch.int64, device=dev)
                    g = torch.cuda.CUDAGraph()              # the whole epoch: nb steps back to back
                    if full:
                        quiesce_collectives(dev)
                    with ops.graph_capture(g):
                        for k in range(nb):
                            if self.dp.active:
                                self._dp_step(dict(b, idx=self._idx_all[k]), split, None, capturing=True)
                            else:
                                self._step(dict(b, idx=self._idx_all[k]), capturing=True)
                            self.optimizer.step_count -= 1
                    self._graphs = (g,)
                else:                                       # the all-reduce(s) sit between the graphs
                    g1, g2, g3 = torch.cuda.CUDAGraph(), None, torch.cuda.CUDAGraph()
                    with ops.graph_capture(g1):
                        self._fwd_bwd(b, phase=1 if split is not None else None)
                    if split is not None:
                        g2 = torch.cuda.CUDAGraph()
                        with ops.graph_capture(g2):
                            self._fwd_bwd(b, phase=2)
                    with ops.graph_capture(g3):
                        self.optimizer.step(self.max_grad_norm, capturing=True, image_map=self._image_map)
                        self.optimizer.step_count -= 1
                    self._graphs = (g1, g2, g3)
                self._graph_key = key
            self._seen_key = key
            for e in range(self.ppo_epoch):
                if per_pos:
                    # (one randperm per epoch either way: same RNG consumption); the batches are consecutive slices of
                    # one device permutation, so the epoch's index table is a strided view of it
                    perm, _, _ = rollouts.minibatch_permutation(self.num_mini_batch)    # ppo.py:48-51
                    self._idx_all.copy_(perm[:nb * mbs].view(nb, mbs)[:, lo:hi])
                    self._graphs[0].replay()
                    self.optimizer.step_count += nb
                    if full and not self._full_checked:
                        # first replay of an epoch graph that contains RCCL collectives: bounded wait, then trusted
                        bounded_sync(dev, float(os.environ.get('ACKTR_DP_GRAPH_TIMEOUT', '120')),
                                     'the first replay of the ACKTR_DP_GRAPH=full epoch graph (captured RCCL all-reduces)')
                        self._full_checked = True
                    continue
                for indices in rollouts.minibatch_index_batches(self.num_mini_batch):   # ppo.py:48-51
                    self._idx.copy_(indices[lo:hi])
                    if self.dp.active:
                        self._dp_step(b, split, self._graphs if graph_ok else None)
                    else:
                        self._step(b)                                                  # ppo.py:57-84
            self.dp.all_reduce_sum_(self._loss_acc)
            num_updates = self.ppo_epoch * self.num_mini_batch
            vl, al, en = (self._loss_acc / num_updates).tolist()                    # the only host sync
        return vl, al, en

    def _update_recurrent(self, rollouts):
        """ppo.py:34-96 with `recurrent_generator` (ppo.py:44-46): a minibatch is the whole trajectory of
        N // num_mini_batch envs.  Rows are addressed through one index tensor (t*N + env), so the fused gathers of
        the feed-forward path serve here too; data-parallel ranks split the ENVS of a minibatch."""
        ac = self.actor_critic
        self._image_map = None                      # (the recurrent step re-orders its conv weights every step)
        T, N = rollouts.rewards.size()[0:2]
        rows = T * N
        dev = rollouts.rewards.device
        H = ac.recurrent_hidden_state_size
        with torch.no_grad():
            adv = rollouts.normalized_advantages()                                   # ppo.py:35-37
            loss_acc = torch.zeros(3, dtype=torch.float32, device=dev)
            masks_flat = rollouts.masks[:T].reshape(rows, 1)
            b = dict(obs=rollouts.obs[:T].view(rows, *rollouts.obs.size()[2:]), actions=rollouts.actions.view(rows, -1),
                     old_values=rollouts.value_preds[:T].view(rows), returns=rollouts.returns[:T].view(rows),
                     old_logp=rollouts.action_log_probs.view(rows), adv=adv.view(rows), loss_acc=loss_acc)
            ac.bind_grads()
            self.optimizer.prepare()
            n_steps = 0
            # Single process: the minibatch step (forward over T time steps, loss, BPTT, clip + Adam: ~300 launches at
            # T = 128) is replayed from a hipGraph captured at the second update that sees the same storage and shapes;
            # its operands are persistent buffers (row index, initial states, masks) refilled before every replay.
            ne = N // self.num_mini_batch
            key = ('rec', rollouts.obs.data_ptr(), rollouts.actions.data_ptr(), rollouts.returns.data_ptr(),
                   rollouts.value_preds.data_ptr(), rollouts.action_log_probs.data_ptr(), T, N, ne,
                   ac.flat_params.data_ptr(), self.clip_param, self.value_loss_coef, self.entropy_coef, self.max_grad_norm,
                   self.use_clipped_value_loss)
            graph_ok = (self.use_graph and ops.profiler is None and not self.dp.active and ne >= 1
                        and self._seen_key == key)
            if self._seen_key != key:
                self._rgraph = None
                self._rbuf = dict(idx=torch.zeros(T * ne, dtype=torch.int64, device=dev),
                                  hxs=torch.zeros(ne, H, dtype=torch.float32, device=dev),
                                  masks=torch.zeros(T * ne, 1, dtype=torch.float32, device=dev),
                                  adv=torch.zeros(rows, dtype=torch.float32, device=dev),
                                  loss_acc=torch.zeros(3, dtype=torch.float32, device=dev))
            self._seen_key = key
            if graph_ok:
                rb = self._rbuf
                rb['adv'].copy_(adv.view(rows))
                rb['loss_acc'].zero_()
                gb = dict(b, adv=rb['adv'], loss_acc=rb['loss_acc'], idx=rb['idx'], hxs=rb['hxs'], masks=rb['masks'],
                          inv_batch=1.0 / (T * ne))
                if self._rgraph is None:                    # capture (records only; executes nothing)
                    g = torch.cuda.CUDAGraph()
                    with ops.graph_capture(g):
                        self._fwd_bwd(gb)
                        self.optimizer.step(self.max_grad_norm, capturing=True)
                        self.optimizer.step_count -= 1
                    self._rgraph = g
            for e in range(self.ppo_epoch):
                for envs in rollouts.recurrent_env_batches(self.num_mini_batch):
                    lo, hi = self.dp.bounds(envs.numel())
                    mine = envs[lo:hi].to(dev)
                    idx = rollouts.recurrent_rows(mine)
                    hxs = rollouts._gather(rollouts.recurrent_hidden_states[0].view(N, H), mine, N)
                    masks = rollouts._gather(masks_flat, idx, rows)
                    n_steps += 1
                    if graph_ok and envs.numel() == ne:
                        rb['idx'].copy_(idx)
                        rb['hxs'].copy_(hxs)
                        rb['masks'].copy_(masks)
                        self._rgraph.replay()
                        self.optimizer.step_count += 1
                        continue
                    b['idx'], b['hxs'], b['masks'] = idx, hxs, masks
                    b['inv_batch'] = 1.0 / (T * envs.numel())
                    self._fwd_bwd(b)
                    self.dp.all_reduce_sum_(ac.flat_grads)
                    self.optimizer.step(self.max_grad_norm)                            # ppo.py:82-84
            if graph_ok:
                loss_acc.add_(rb['loss_acc'])
            self.dp.all_reduce_sum_(loss_acc)
            vl, al, en = (loss_acc / n_steps).tolist()
        return vl, al, en
