"""PPO -- host-side mirror of reference a2c_ppo_acktr/algo/ppo.py:7-96.

Same constructor and `update(rollouts) -> (value_loss, action_loss, dist_entropy)` contract.  One
minibatch step is: HIP forward (minibatch gather fused into the first layer) -> fused
categorical/gaussian + clipped-surrogate loss kernel that also emits the backward seeds -> HIP backward
into the flat gradient bucket -> [RCCL all-reduce] -> clip-norm + Adam in two launches.  Nothing syncs
with the host until the three averaged losses are read back once at the end of `update`.
"""
import torch

from .. import _hip, engine, ops
from ..parallel import DataParallel


class _FlatAdam(object):
    """optim.Adam over the policy's flat buffer (torch/optim/adam.py:529-547 single-tensor math).
    Exposes `param_groups` so update_linear_schedule (utils.py:46-50) works as in the reference."""

    def __init__(self, policy, lr, eps, betas=(0.9, 0.999)):
        self.policy = policy
        self.param_groups = [{'params': list(policy.parameters()), 'lr': lr, 'eps': eps, 'betas': betas}]
        self.step_count = 0
        self.exp_avg = None
        self.exp_avg_sq = None

    def _state(self):
        flat = self.policy.flat_params
        if self.exp_avg is None or self.exp_avg.device != flat.device or self.exp_avg.numel() != flat.numel():
            self.exp_avg = torch.zeros_like(flat)
            self.exp_avg_sq = torch.zeros_like(flat)

    def zero_grad(self, set_to_none=False):
        self.policy.flat_grads.zero_()

    def step(self, max_grad_norm=None):
        """clip_grad_norm_(max_grad_norm) (if given) + Adam step on the flat buffers."""
        self._state()
        g = self.param_groups[0]
        self.step_count += 1
        b1, b2 = g['betas']
        bc1 = 1 - b1 ** self.step_count
        bc2 = 1 - b2 ** self.step_count
        flat, gflat = self.policy.flat_params, self.policy.flat_grads
        L = _hip.lib()
        ws = ops.workspace(flat.device, L.acktr_optim_workspace_bytes(), 'optim')
        rc = L.acktr_clip_adam_f32(_hip.dev_ptr(flat), _hip.dev_ptr(gflat), _hip.dev_ptr(self.exp_avg),
                                   _hip.dev_ptr(self.exp_avg_sq), flat.numel(),
                                   float(max_grad_norm) if max_grad_norm else 0.0, g['lr'] / bc1, b1, b2, g['eps'],
                                   bc2 ** 0.5, None, _hip.dev_ptr(ws), _hip.cur_stream(flat.device))
        _hip.check(rc, 'acktr_clip_adam_f32')


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

    def update(self, rollouts):
        ac = self.actor_critic
        if ac.is_recurrent:
            raise NotImplementedError('recurrent policies are outside the MI355X build')
        T, N = rollouts.rewards.size()[0:2]
        rows = T * N
        dev = rollouts.rewards.device
        with torch.no_grad():
            advantages = rollouts.normalized_advantages()                      # ppo.py:35-37
            loss_acc = torch.zeros(3, dtype=torch.float32, device=dev)
            obs_flat = rollouts.obs[:T].view(rows, *rollouts.obs.size()[2:])
            actions = rollouts.actions.view(rows, -1)
            old_values = rollouts.value_preds[:T].view(rows)
            returns = rollouts.returns[:T].view(rows)
            old_logp = rollouts.action_log_probs.view(rows)
            adv = advantages.view(rows)
            ac.bind_grads()
            for e in range(self.ppo_epoch):
                for indices in rollouts.minibatch_index_batches(self.num_mini_batch):   # ppo.py:48-51
                    inv_batch = 1.0 / indices.numel()
                    idx = self.dp.shard(indices).contiguous()
                    acts = engine.forward(ac, obs_flat, gather_idx=idx)             # ppo.py:57-59
                    engine.policy_loss(ac, acts, engine.ALGO_PPO, actions, loss_acc, inv_batch,
                                       old_logp=old_logp, old_values=old_values, returns=returns, adv=adv,
                                       gather_idx=idx, clip=self.clip_param, value_coef=self.value_loss_coef,
                                       entropy_coef=self.entropy_coef,
                                       use_clipped_value_loss=self.use_clipped_value_loss)   # ppo.py:61-81
                    engine.backward(ac, acts, ac.flat_grads)
                    self.dp.all_reduce_sum_(ac.flat_grads)
                    self.optimizer.step(self.max_grad_norm)                        # ppo.py:82-84
            self.dp.all_reduce_sum_(loss_acc)
            num_updates = self.ppo_epoch * self.num_mini_batch
            vl, al, en = (loss_acc / num_updates).tolist()                          # the only host sync
        return vl, al, en
