"""RolloutStorage -- host-side mirror of reference a2c_ppo_acktr/storage.py:9-143.

Same constructor, public tensor attributes and methods as the reference; the
arithmetic (`compute_returns`, the minibatch gathers) runs in hand-written HIP
kernels through the C-ABI (include/acktr_hip.h).  There is no CPU path: the
storage must be moved to the GPU with `.to(device)` before `compute_returns`
or a generator is used.
"""
import ctypes

import numpy as np
import torch

from . import _hip


def _flat(t, rows):
    return t.view(rows, -1)


class RolloutStorage(object):
    def __init__(self, num_steps, num_processes, obs_shape, action_space,
                 recurrent_hidden_state_size, obs_dtype=torch.float32):
        """reference storage.py:10-33 (allocation only; fp32 observations, F6).
        obs_dtype=torch.uint8 (not in the reference): Atari frames are kept as bytes -- `insert` /
        `obs[0].copy_(obs)` convert the float frames the envs deliver (integer values 0..255, envs.py:176,188) exactly,
        the NatureCNN kernels read the bytes directly (a quarter of the observation traffic, 0.93 GB instead of 3.7 for
        128 x 256 frames), and every result is bit-identical to the float storage.  `obs` is then a uint8 tensor:
        Policy.act / get_value / evaluate_actions and the algorithms of this package take it as it is."""
        if obs_dtype not in (torch.float32, torch.uint8):
            raise ValueError('obs_dtype must be torch.float32 (the reference) or torch.uint8')
        if obs_dtype == torch.uint8 and len(obs_shape) != 3:
            raise ValueError('uint8 observations are for image observations (NatureCNN) only')
        self.obs = torch.zeros(num_steps + 1, num_processes, *obs_shape, dtype=obs_dtype)
        self.recurrent_hidden_states = torch.zeros(
            num_steps + 1, num_processes, recurrent_hidden_state_size)
        self.rewards = torch.zeros(num_steps, num_processes, 1)
        self.value_preds = torch.zeros(num_steps + 1, num_processes, 1)
        self.returns = torch.zeros(num_steps + 1, num_processes, 1)
        self.action_log_probs = torch.zeros(num_steps, num_processes, 1)
        if action_space.__class__.__name__ == 'Discrete':
            action_shape = 1
        else:
            action_shape = action_space.shape[0]
        self.actions = torch.zeros(num_steps, num_processes, action_shape)
        if action_space.__class__.__name__ == 'Discrete':
            self.actions = self.actions.long()
        self.masks = torch.ones(num_steps + 1, num_processes, 1)
        self.bad_masks = torch.ones(num_steps + 1, num_processes, 1)

        self.num_steps = num_steps
        self.step = 0
        self._adv_ws = None

    _TENSORS = ('obs', 'recurrent_hidden_states', 'rewards', 'value_preds', 'returns',
                'action_log_probs', 'actions', 'masks', 'bad_masks')

    def to(self, device):
        # reference storage.py:35-44
        for k in self._TENSORS:
            setattr(self, k, getattr(self, k).to(device))

    def insert(self, obs, recurrent_hidden_states, actions, action_log_probs,
               value_preds, rewards, masks, bad_masks):
        # reference storage.py:46-58
        self.obs[self.step + 1].copy_(obs)
        self.recurrent_hidden_states[self.step + 1].copy_(recurrent_hidden_states)
        self.actions[self.step].copy_(actions)
        self.action_log_probs[self.step].copy_(action_log_probs)
        self.value_preds[self.step].copy_(value_preds)
        self.rewards[self.step].copy_(rewards)
        self.masks[self.step + 1].copy_(masks)
        self.bad_masks[self.step + 1].copy_(bad_masks)
        self.step = (self.step + 1) % self.num_steps

    def after_update(self):
        # reference storage.py:60-64
        self.obs[0].copy_(self.obs[-1])
        self.recurrent_hidden_states[0].copy_(self.recurrent_hidden_states[-1])
        self.masks[0].copy_(self.masks[-1])
        self.bad_masks[0].copy_(self.bad_masks[-1])

    # ------------------------------------------------------------------ compute_returns
    def compute_returns(self, next_value, use_gae, gamma, gae_lambda,
                        use_proper_time_limits=True):
        """reference storage.py:66-105 -- one launch of the lane-per-env reverse scan."""
        T, N = self.rewards.size(0), self.rewards.size(1)
        L = _hip.lib()
        f32 = torch.float32
        nv = next_value
        if not torch.is_tensor(nv):
            raise TypeError('next_value must be a tensor')
        if nv.device != self.rewards.device or nv.dtype != f32 or not nv.is_contiguous():
            nv = nv.to(device=self.rewards.device, dtype=f32).contiguous()
        if nv.numel() != N:
            raise RuntimeError('next_value must have %d elements, got %d' % (N, nv.numel()))
        # python scalars -> fp32 once; gamma*gae_lambda multiplied in double first (F8)
        g32 = float(np.float32(gamma))
        gl32 = float(np.float32(float(gamma) * float(gae_lambda)))
        rc = L.acktr_compute_returns_f32(
            _hip.dev_ptr(self.rewards, f32, 'rewards'), _hip.dev_ptr(self.value_preds, f32, 'value_preds'),
            _hip.dev_ptr(self.returns, f32, 'returns'), _hip.dev_ptr(self.masks, f32, 'masks'),
            _hip.dev_ptr(self.bad_masks, f32, 'bad_masks'), _hip.dev_ptr(nv, f32, 'next_value'),
            T, N, g32, gl32, 1 if use_gae else 0, 1 if use_proper_time_limits else 0,
            _hip.cur_stream(self.rewards.device))
        _hip.check(rc, 'acktr_compute_returns_f32')

    # ------------------------------------------------------------------ advantages (ppo.py:35-37)
    def normalized_advantages(self, out=None):
        """(R[:-1]-V[:-1] - mean)/(std_unbiased + 1e-5) as a [T,N,1] device tensor (into `out` if given)."""
        T, N = self.rewards.size(0), self.rewards.size(1)
        L = _hip.lib()
        dev = self.rewards.device
        if self._adv_ws is None or self._adv_ws.device != dev:
            self._adv_ws = torch.empty(L.acktr_adv_normalize_workspace_bytes(), dtype=torch.uint8, device=dev)
        adv = out if out is not None else torch.empty(T, N, 1, dtype=torch.float32, device=dev)
        assert adv.shape == (T, N, 1) and adv.is_contiguous() and adv.device == dev
        rc = L.acktr_adv_normalize_f32(
            _hip.dev_ptr(self.returns, torch.float32, 'returns'),
            _hip.dev_ptr(self.value_preds, torch.float32, 'value_preds'),
            _hip.dev_ptr(adv), T * N, _hip.dev_ptr(self._adv_ws), _hip.cur_stream(dev))
        _hip.check(rc, 'acktr_adv_normalize_f32')
        return adv

    # ------------------------------------------------------------------ samplers
    @staticmethod
    def _gather(src2d, idx, src_rows):
        """dst[i,:] = src2d[idx[i],:] through the HIP gather kernel."""
        rows = idx.numel()
        out = torch.empty((rows,) + tuple(src2d.shape[1:]), dtype=src2d.dtype, device=src2d.device)
        row_bytes = src2d[0].numel() * src2d.element_size() if src2d.size(0) > 0 else 0
        if rows == 0 or row_bytes == 0:
            return out
        rc = _hip.lib().acktr_gather_rows(_hip.dev_ptr(src2d, None, 'gather source'),
                                          _hip.dev_ptr(idx, torch.int64, 'indices'), _hip.dev_ptr(out),
                                          rows, row_bytes, src_rows, _hip.cur_stream(src2d.device))
        _hip.check(rc, 'acktr_gather_rows')
        return out

    def minibatch_index_batches(self, num_mini_batch=None, mini_batch_size=None):
        """Index batches exactly as BatchSampler(SubsetRandomSampler(range(T*N)), mbs, drop_last=True)
        draws them (reference storage.py:111-125): ONE torch.randperm on the CPU default generator
        per call (F7), trailing partial batch dropped.  Returns an iterable of int64 device tensors (slices of one
        device copy of the permutation, cut on demand: a GAIL-style caller with mini_batch_size=128 over 2 M rows
        consumes one or two of the 16 384 batches)."""
        num_steps, num_processes = self.rewards.size()[0:2]
        batch_size = num_processes * num_steps
        if mini_batch_size is None:
            assert batch_size >= num_mini_batch, (
                "PPO requires the number of processes ({}) "
                "* number of steps ({}) = {} "
                "to be greater than or equal to the number of PPO mini batches ({})."
                "".format(num_processes, num_steps, num_processes * num_steps,
                          num_mini_batch))
            mini_batch_size = batch_size // num_mini_batch
        perm_dev, mini_batch_size, nb = self.minibatch_permutation(num_mini_batch, mini_batch_size, _checked=True)
        return (perm_dev[i * mini_batch_size:(i + 1) * mini_batch_size] for i in range(nb))

    def minibatch_permutation(self, num_mini_batch=None, mini_batch_size=None, _checked=False):
        """The permutation behind minibatch_index_batches as ONE int64 device tensor plus (mini_batch_size, number of
        batches): batch i is perm[i*mbs:(i+1)*mbs], so an epoch's index table is perm[:nb*mbs].view(nb, mbs)."""
        num_steps, num_processes = self.rewards.size()[0:2]
        batch_size = num_processes * num_steps
        if mini_batch_size is None:
            if not _checked:
                assert batch_size >= num_mini_batch
            mini_batch_size = batch_size // num_mini_batch
        perm = torch.randperm(batch_size)
        perm_dev = perm.to(self.rewards.device, non_blocking=True)
        return perm_dev, mini_batch_size, batch_size // mini_batch_size

    def feed_forward_generator(self, advantages, num_mini_batch=None, mini_batch_size=None):
        """reference storage.py:107-143: yields the same 8-tuples, gathered on the GPU."""
        T, N = self.rewards.size()[0:2]
        rows = T * N
        for indices in self.minibatch_index_batches(num_mini_batch, mini_batch_size):
            indices = indices.contiguous()
            g = lambda t: self._gather(t[:T].view(rows, *t.size()[2:]), indices, rows)
            obs_batch = g(self.obs)
            recurrent_hidden_states_batch = g(self.recurrent_hidden_states)
            actions_batch = g(self.actions)
            value_preds_batch = g(self.value_preds)
            return_batch = g(self.returns)
            masks_batch = g(self.masks)
            old_action_log_probs_batch = g(self.action_log_probs)
            if advantages is None:
                adv_targ = None
            else:
                adv_targ = self._gather(advantages.contiguous().view(rows, 1), indices, rows)
            yield obs_batch, recurrent_hidden_states_batch, actions_batch, \
                value_preds_batch, return_batch, masks_batch, old_action_log_probs_batch, adv_targ

    def recurrent_env_batches(self, num_mini_batch):
        """The env groups recurrent_generator walks (reference storage.py:146-153): ONE torch.randperm over the
        processes on the CPU default generator per call, consecutive groups of N // num_mini_batch envs.
        Returns a list of int64 CPU tensors."""
        num_processes = self.rewards.size(1)
        assert num_processes >= num_mini_batch, (
            "PPO requires the number of processes ({}) "
            "to be greater than or equal to the number of "
            "PPO mini batches ({}).".format(num_processes, num_mini_batch))
        num_envs_per_batch = num_processes // num_mini_batch
        perm = torch.randperm(num_processes)
        if num_processes % num_envs_per_batch != 0:
            # the reference's loop then reads perm[] out of range in its last batch (storage.py:153,164)
            raise IndexError('recurrent_generator: %d processes do not split into batches of %d'
                             % (num_processes, num_envs_per_batch))
        return [perm[s:s + num_envs_per_batch] for s in range(0, num_processes, num_envs_per_batch)]

    def recurrent_rows(self, envs):
        """Flat storage rows t*N + env of whole trajectories, ordered row t*Nb + j (storage.py:176-199), as an int64
        device tensor."""
        T, N = self.rewards.size()[0:2]
        dev = self.rewards.device
        envs = envs.to(dev)
        return (torch.arange(T, device=dev).view(T, 1) * N + envs.view(1, -1)).reshape(-1)

    def recurrent_generator(self, advantages, num_mini_batch):
        """reference storage.py:145-202: whole trajectories of N // num_mini_batch envs per batch, flattened to
        [T*Nb, ...] (row t*Nb + j); hidden states are the ones stored at step 0.  Gathered on the GPU."""
        T, N = self.rewards.size()[0:2]
        rows = T * N
        for envs in self.recurrent_env_batches(num_mini_batch):
            indices = self.recurrent_rows(envs)
            g = lambda t: self._gather(t[:T].view(rows, *t.size()[2:]), indices, rows)
            obs_batch = g(self.obs)
            recurrent_hidden_states_batch = self._gather(self.recurrent_hidden_states[0], envs.to(indices.device), N)
            actions_batch = g(self.actions)
            value_preds_batch = g(self.value_preds)
            return_batch = g(self.returns)
            masks_batch = g(self.masks)
            old_action_log_probs_batch = g(self.action_log_probs)
            adv_targ = self._gather(advantages.contiguous().view(rows, 1), indices, rows)
            yield obs_batch, recurrent_hidden_states_batch, actions_batch, \
                value_preds_batch, return_batch, masks_batch, old_action_log_probs_batch, adv_targ
