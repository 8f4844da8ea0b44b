"""Policy networks -- host-side mirror of reference a2c_ppo_acktr/model.py:15-79,169-229.

`Policy`, `CNNBase`, `MLPBase` keep the reference's constructor signatures, attribute and
parameter names (`base.main.{0,2,4,7}`, `base.critic_linear`, `dist.linear` / `dist.fc_mean` /
`dist.logstd._bias`, `base.actor.{0,2}`, `base.critic.{0,2}`) and are initialised by the same torch
initialiser calls in the same order, so `torch.manual_seed(s)` gives the reference's weights.

The nn.Modules are parameter containers only: every forward/backward runs in the HIP kernels of
csrc/ through `engine.py`.  All parameters of a Policy are views into ONE flat fp32 buffer
(`policy.flat_params`, gradient twin `policy.flat_grads`), which is what lets clip-norm + optimiser
be two launches and the data-parallel all-reduce a single bucket.

Recurrent bases (`base_kwargs={'recurrent': True}`, model.py:89-95,111-166): `base.gru` is an nn.GRU used as a
parameter container like the rest; the sequence runs in engine.gru_forward / gru_backward.

Not built: the broken Bernoulli head (reference distributions.py:49-50, F11).
"""
import numpy as np
import os

import torch
import torch.nn as nn

from . import engine
from .distributions import Categorical, DiagGaussian, FixedCategorical, FixedNormal
from .utils import init


class Flatten(nn.Module):
    def forward(self, x):
        return x.view(x.size(0), -1)


def _align4(n):
    return (n + 3) // 4 * 4


_DEFAULT_SAMPLE_STREAM = 'device' if os.environ.get('ACKTR_ACT_RNG', 'reference') == 'device' else 'reference'


class Policy(nn.Module):
    def __init__(self, obs_shape, action_space, base=None, base_kwargs=None):
        super(Policy, self).__init__()
        if base_kwargs is None:
            base_kwargs = {}
        if base is None:
            if len(obs_shape) == 3:
                base = CNNBase
            elif len(obs_shape) == 1:
                base = MLPBase
            else:
                raise NotImplementedError

        self.base = base(obs_shape[0], **base_kwargs)

        if action_space.__class__.__name__ == "Discrete":
            num_outputs = action_space.n
            self.dist = Categorical(self.base.output_size, num_outputs)
        elif action_space.__class__.__name__ == "Box":
            num_outputs = action_space.shape[0]
            self.dist = DiagGaussian(self.base.output_size, num_outputs)
        elif action_space.__class__.__name__ == "MultiBinary":
            raise NotImplementedError('Bernoulli head: broken in the reference (distributions.py:49-50), not built')
        else:
            raise NotImplementedError
        self.obs_shape = tuple(obs_shape)
        if isinstance(self.base, CNNBase) and tuple(obs_shape[1:]) != (84, 84):
            raise NotImplementedError('CNNBase expects 84x84 frames (reference model.py:176-180)')
        if isinstance(self.dist, DiagGaussian) and num_outputs > 16:
            raise NotImplementedError('DiagGaussian head: at most 16 action dimensions (csrc/loss.hip)')
        self._flatten_parameters()

    # ------------------------------------------------------------------ flat parameter storage
    def _param_groups_for_flat(self):
        """Groups of parameters that must be contiguous; each group starts 16-byte aligned."""
        if isinstance(self.base, CNNBase):
            w = engine.cnn_layer_params(self)
            groups = [[p] for p in self.base.gru.parameters()] if self.base.is_recurrent else []
            for name in ('conv1', 'conv2', 'conv3', 'fc'):
                groups.append([w[name][0]])
                groups.append([w[name][1]])
            groups.append([w['critic'][0], w['actor'][0]])     # fused head weight [1+A, hidden]
            groups.append([w['critic'][1], w['actor'][1]])     # fused head bias   [1+A]
            if self.dist.kind == 1:                            # CNN + Box: the state-independent logstd
                groups.append([engine.logstd_param(self)])
            return groups
        return [[p] for p in self.parameters()]

    def _flatten_parameters(self):
        groups = self._param_groups_for_flat()
        listed = set(id(p) for g in groups for p in g)
        assert listed == set(id(p) for p in self.parameters()), 'flat layout must cover every parameter'
        first = next(self.parameters())
        offs, off = {}, 0
        for g in groups:
            off = _align4(off)
            for p in g:
                offs[id(p)] = (off, p.numel())
                off += p.numel()
        total = _align4(off)
        flat = torch.zeros(total, dtype=torch.float32, device=first.device)
        gflat = torch.zeros(total, dtype=torch.float32, device=first.device)
        for p in self.parameters():
            o, n = offs[id(p)]
            flat[o:o + n].copy_(p.data.reshape(-1))
            if p.grad is not None:
                gflat[o:o + n].copy_(p.grad.data.reshape(-1))
            p.data = flat[o:o + n].view(p.shape)
            p.grad = None
        self.__dict__['flat_params'] = flat
        self.__dict__['flat_grads'] = gflat
        self.__dict__['_offsets'] = offs

    # checkpoints: the reference pickles the whole module (main.py:165-176, enjoy.py:52-54).  The flat buffers and
    # the id()-keyed offset table are derived state: they are left out of the pickle and rebuilt from the parameters.
    _DERIVED = ('flat_params', 'flat_grads', '_offsets', '_pack_ws', '_lp_cache')

    def __getstate__(self):
        state = self.__dict__.copy()
        for k in self._DERIVED:
            state.pop(k, None)
        return state

    def __setstate__(self, state):
        super(Policy, self).__setstate__(state)
        self._flatten_parameters()

    def _apply(self, fn, *args, **kwargs):
        r = super(Policy, self)._apply(fn, *args, **kwargs)
        self._flatten_parameters()
        return r

    def rebind_flat(self):
        """Re-derive offsets after module surgery that re-wraps parameters (KFAC's split_bias)."""
        base_ptr, esz = self.flat_params.data_ptr(), 4
        offs = {}
        for p in self.parameters():
            d = p.data.data_ptr() - base_ptr
            assert 0 <= d < self.flat_params.numel() * esz and d % esz == 0, 'parameter left the flat buffer'
            offs[id(p)] = (d // esz, p.numel())
        self.__dict__['_offsets'] = offs

    def grad_view(self, p, flat=None):
        o, n = self._offsets[id(p)]
        return (self.flat_grads if flat is None else flat)[o:o + n].view(p.shape)

    def bind_grads(self):
        """Make every p.grad a view of flat_grads (idempotent; zero_grad(set_to_none=True) undoes it)."""
        for p in self.parameters():
            gv = self.grad_view(p)
            if p.grad is None or p.grad.data_ptr() != gv.data_ptr():
                p.grad = gv

    def adopt_grads(self):
        """Gradients that autograd left in tensors of their own (after `module.zero_grad()` set .grad to None, the next
        backward() allocates fresh ones) are copied into the flat gradient bucket and re-bound, so an optimiser of this
        package that is driven by user code (loss.backward(); optimizer.step()) steps on them."""
        for p in self.parameters():
            gv = self.grad_view(p)
            if p.grad is not None and p.grad.data_ptr() != gv.data_ptr():
                gv.copy_(p.grad)
                p.grad = gv

    # ------------------------------------------------------------------ reference API
    @property
    def is_recurrent(self):
        return self.base.is_recurrent

    @property
    def recurrent_hidden_state_size(self):
        """Size of rnn_hx."""
        return self.base.recurrent_hidden_state_size

    def forward(self, inputs, rnn_hxs, masks):
        raise NotImplementedError

    def _dist_from_head(self, acts):
        """torch-side view of the head's distribution (not used by act(); kept for user code that wants the object)."""
        if self.dist.kind == 0:
            return FixedCategorical(acts.dist_in())
        logstd = engine.logstd_param(self).detach().view(1, -1)
        return FixedNormal(acts.dist_in(), logstd.exp().expand_as(acts.dist_in()))

    def act(self, inputs, rnn_hxs, masks, deterministic=False):
        """reference model.py:54-66, rollout side: HIP forward without grad, then ONE head kernel (csrc/loss.hip
        dist_act_kernel) for mode / sample and the log-prob of the chosen action.
        Sampling stream (`self.sample_stream`, default from ACKTR_ACT_RNG, 'reference' unless set to 'device'):
        'reference' consumes the CPU default generator exactly as the reference's CPU path does -- Categorical:
        torch.multinomial(probs, 1)'s one-sample path draws q = empty_like(probs).exponential_(1) and returns
        argmax p / q; DiagGaussian: torch.normal(mean, std) draws a standard normal tensor of mean's shape -- so the same
        torch.manual_seed gives the reference's actions (tests/golden/act.npz `sample/*`).  'device' draws uniform [B] /
        normal [B, A] noise on the GPU generator instead: same distributions, no host draw or copy, another stream."""
        from . import _hip
        with torch.no_grad():
            acts = engine.forward(self, inputs, rnn_hxs=rnn_hxs, masks=masks, gate_bits=False)
            dev = inputs.device
            B, A, kind = inputs.shape[0], self.dist.num_outputs, self.dist.kind
            noise, mode = None, 1
            if not deterministic:
                if getattr(self, 'sample_stream', _DEFAULT_SAMPLE_STREAM) == 'device':
                    noise, mode = (torch.rand(B, device=dev) if kind == 0 else torch.randn(B, A, device=dev)), 0
                elif kind == 0:
                    noise, mode = torch.empty(B, A).exponential_(1).to(dev, non_blocking=True), 2
                else:
                    noise, mode = torch.empty(B, A).normal_().to(dev, non_blocking=True), 0
            action = torch.empty((B, 1), dtype=torch.int64, device=dev) if kind == 0 else \
                torch.empty((B, A), dtype=torch.float32, device=dev)
            logp = torch.empty(B, 1, dtype=torch.float32, device=dev)
            ls = engine.logstd_param(self).detach() if kind == 1 else None
            dp, dld = acts.dist_ptr_ld()
            rc = _hip.lib().acktr_dist_act_f32(kind, dp, dld, _hip.dev_ptr(ls), _hip.dev_ptr(noise),
                                               mode, B, A, _hip.dev_ptr(action, action.dtype),
                                               _hip.dev_ptr(logp), _hip.cur_stream(dev))
            _hip.check(rc, 'acktr_dist_act_f32')
            if self.is_recurrent:
                rnn_hxs = acts.hxs_out.clone()
            return acts.value().clone(), action, logp, rnn_hxs

    def get_value(self, inputs, rnn_hxs, masks):
        """reference model.py:68-70."""
        with torch.no_grad():
            return engine.forward(self, inputs, rnn_hxs=rnn_hxs, masks=masks, gate_bits=False).value().clone()

    def evaluate_actions(self, inputs, rnn_hxs, masks, action):
        """reference model.py:72-79: (value [B,1], action_log_probs [B,1], dist_entropy scalar, rnn_hxs).
        Differentiable w.r.t. every parameter through a torch.autograd.Function whose forward and
        backward are the HIP kernels."""
        if self.is_recurrent:
            value, logp, ent, hxs = engine.EvalActions.apply(self, inputs, action, rnn_hxs, masks, None, False,
                                                             *list(self.parameters()))
            return value, logp, ent, hxs
        # a KFACOptimizer built on this policy gathers its statistics here, where the reference's module hooks fire
        # (kfac.py:144-177): input statistics on every forward with grad enabled at steps % Ts == 0
        kfac = engine.kfac_of(self)
        stats_a = kfac is not None and torch.is_grad_enabled() and kfac.steps % kfac.Ts == 0
        value, logp, ent, _ = engine.EvalActions.apply(self, inputs, action, None, None, kfac, stats_a,
                                                       *list(self.parameters()))
        return value, logp, ent, rnn_hxs


class NNBase(nn.Module):
    def __init__(self, recurrent, recurrent_input_size, hidden_size):
        super(NNBase, self).__init__()
        self._hidden_size = hidden_size
        self._recurrent = recurrent
        if recurrent:                                   # reference model.py:89-95 (same initialiser calls, same order)
            self.gru = nn.GRU(recurrent_input_size, hidden_size)
            for name, param in self.gru.named_parameters():
                if 'bias' in name:
                    nn.init.constant_(param, 0)
                elif 'weight' in name:
                    nn.init.orthogonal_(param)

    @property
    def is_recurrent(self):
        return self._recurrent

    @property
    def recurrent_hidden_state_size(self):
        if self._recurrent:
            return self._hidden_size
        return 1

    @property
    def output_size(self):
        return self._hidden_size


class CNNBase(NNBase):
    def __init__(self, num_inputs, recurrent=False, hidden_size=512):
        super(CNNBase, self).__init__(recurrent, hidden_size, hidden_size)

        init_ = lambda m: init(m, nn.init.orthogonal_, lambda x: nn.init.
                               constant_(x, 0), nn.init.calculate_gain('relu'))

        self.main = nn.Sequential(
            init_(nn.Conv2d(num_inputs, 32, 8, stride=4)), nn.ReLU(),
            init_(nn.Conv2d(32, 64, 4, stride=2)), nn.ReLU(),
            init_(nn.Conv2d(64, 32, 3, stride=1)), nn.ReLU(), Flatten(),
            init_(nn.Linear(32 * 7 * 7, hidden_size)), nn.ReLU())

        init_ = lambda m: init(m, nn.init.orthogonal_, lambda x: nn.init.
                               constant_(x, 0))

        self.critic_linear = init_(nn.Linear(hidden_size, 1))

        self.train()


class MLPBase(NNBase):
    def __init__(self, num_inputs, recurrent=False, hidden_size=64):
        super(MLPBase, self).__init__(recurrent, num_inputs, hidden_size)

        if recurrent:
            num_inputs = hidden_size

        init_ = lambda m: init(m, nn.init.orthogonal_, lambda x: nn.init.
                               constant_(x, 0), np.sqrt(2))

        self.actor = nn.Sequential(
            init_(nn.Linear(num_inputs, hidden_size)), nn.Tanh(),
            init_(nn.Linear(hidden_size, hidden_size)), nn.Tanh())

        self.critic = nn.Sequential(
            init_(nn.Linear(num_inputs, hidden_size)), nn.Tanh(),
            init_(nn.Linear(hidden_size, hidden_size)), nn.Tanh())

        self.critic_linear = init_(nn.Linear(hidden_size, 1))

        self.train()
