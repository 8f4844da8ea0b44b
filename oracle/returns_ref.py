"""Oracle: return / GAE reverse scan (numpy fp32).

TEST INFRASTRUCTURE -- see oracle/__init__.py.

Restates ``RolloutStorage.compute_returns`` (reference
a2c_ppo_acktr/storage.py:66-105) with the exact fp32 expression trees the
reference's torch ops produce: python scalars are rounded to fp32 once
(``gamma`` and the *double* product ``gamma * gae_lambda``), every binary op is
one fp32 rounding, nothing is contracted into an FMA.  Bit-exact against the
reference in all four (use_gae, use_proper_time_limits) modes (pinned by
tests/golden/returns_*.npz).
"""
import numpy as np

f32 = np.float32


def compute_returns(rewards, value_preds, returns, masks, bad_masks,
                    next_value, use_gae, gamma, gae_lambda,
                    use_proper_time_limits=True):
    """In-place on numpy fp32 arrays shaped like the storage tensors.

    rewards [T,N,1]; value_preds, returns, masks, bad_masks [T+1,N,1];
    next_value [N,1].  Side effects follow storage.py:74,85,92,102:
    GAE modes write value_preds[-1]; non-GAE modes write returns[-1].
    """
    T = rewards.shape[0]
    g = f32(gamma)                      # scalar -> fp32 once
    gl = f32(gamma * gae_lambda)        # product in double, then fp32
    one = f32(1.0)
    if use_proper_time_limits:
        if use_gae:                     # storage.py:73-83
            value_preds[-1] = next_value
            gae = np.zeros_like(rewards[0])
            for t in reversed(range(T)):
                delta = rewards[t] + (g * value_preds[t + 1]) * masks[t + 1] \
                    - value_preds[t]
                gae = delta + (gl * masks[t + 1]) * gae
                gae = gae * bad_masks[t + 1]
                returns[t] = gae + value_preds[t]
        else:                           # storage.py:85-89
            returns[-1] = next_value
            for t in reversed(range(T)):
                returns[t] = ((returns[t + 1] * g) * masks[t + 1]
                              + rewards[t]) * bad_masks[t + 1] \
                    + (one - bad_masks[t + 1]) * value_preds[t]
    else:
        if use_gae:                     # storage.py:91-100
            value_preds[-1] = next_value
            gae = np.zeros_like(rewards[0])
            for t in reversed(range(T)):
                delta = rewards[t] + (g * value_preds[t + 1]) * masks[t + 1] \
                    - value_preds[t]
                gae = delta + (gl * masks[t + 1]) * gae
                returns[t] = gae + value_preds[t]
        else:                           # storage.py:102-105
            returns[-1] = next_value
            for t in reversed(range(T)):
                returns[t] = (returns[t + 1] * g) * masks[t + 1] + rewards[t]
