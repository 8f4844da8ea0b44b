"""Closed-form, machine-independent synthetic inputs for parity tests.

Every value is an exact function of (flat index, salt) through a 32-bit integer
hash evaluated in numpy uint64 arithmetic, so the build container (where the
goldens are made from the reference) and the GPU box (where the HIP path is
checked) regenerate bit-identical inputs without shipping large tensors and
without depending on LAPACK / RNG implementation details.
"""
from collections import OrderedDict

import numpy as np

_M32 = np.uint64(0xFFFFFFFF)


def hash_u32(n, salt):
    """murmur3 fmix32 of (index * golden-ratio + salt * c); returns uint64 array < 2^32."""
    i = np.arange(n, dtype=np.uint64)
    h = (i * np.uint64(0x9E3779B1) + np.uint64(salt) * np.uint64(0x85EBCA77) + np.uint64(0x165667B1)) & _M32
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x85EBCA6B)) & _M32
    h ^= h >> np.uint64(13)
    h = (h * np.uint64(0xC2B2AE35)) & _M32
    h ^= h >> np.uint64(16)
    return h


def uniform(shape, salt, lo=-1.0, hi=1.0):
    n = int(np.prod(shape)) if len(shape) else 1
    u = hash_u32(n, salt).astype(np.float64) / 4294967296.0
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def normalish(shape, salt):
    """Irwin-Hall(4) scaled to unit variance; exact in float64."""
    n = int(np.prod(shape)) if len(shape) else 1
    s = np.zeros(n, dtype=np.float64)
    for j in range(4):
        s += hash_u32(n, salt * 4 + j + 1000003).astype(np.float64) / 4294967296.0
    return ((s - 2.0) * np.sqrt(3.0)).astype(np.float32).reshape(shape)


def bernoulli(shape, salt, p_one):
    n = int(np.prod(shape)) if len(shape) else 1
    u = hash_u32(n, salt).astype(np.float64) / 4294967296.0
    return (u < p_one).astype(np.float32).reshape(shape)


def randint(shape, salt, hi):
    n = int(np.prod(shape)) if len(shape) else 1
    return (hash_u32(n, salt) % np.uint64(hi)).astype(np.int64).reshape(shape)


def make_rollouts(T, N, obs_shape, action_kind, A, salt, p_mask=0.97, p_bad=0.98, hidden=None):
    """Synthetic rollout of the shapes in SURVEY.md 8(d).

    action_kind 'discrete' -> int64 actions [T,N,1] in [0,A); 'box' -> fp32 [T,N,A].
    Atari-shaped obs (3-D) are integer valued 0..255 stored as fp32 (F6).
    """
    ro = OrderedDict()
    if len(obs_shape) == 3:
        ro['obs'] = randint((T + 1, N) + tuple(obs_shape), salt + 1, 256).astype(np.float32)
    else:
        ro['obs'] = normalish((T + 1, N) + tuple(obs_shape), salt + 1)
    ro['rewards'] = normalish((T, N, 1), salt + 2)
    ro['value_preds'] = normalish((T + 1, N, 1), salt + 3)
    ro['returns'] = np.zeros((T + 1, N, 1), np.float32)
    ro['action_log_probs'] = (0.1 * normalish((T, N, 1), salt + 4) - 2.0).astype(np.float32)
    if action_kind == 'discrete':
        ro['actions'] = randint((T, N, 1), salt + 5, A)
    else:
        ro['actions'] = normalish((T, N, A), salt + 5)
    ro['masks'] = bernoulli((T + 1, N, 1), salt + 6, p_mask)
    ro['bad_masks'] = bernoulli((T + 1, N, 1), salt + 7, p_bad)
    ro['next_value'] = normalish((N, 1), salt + 8)
    if hidden is not None:                       # recurrent policies: stored GRU states in (-1, 1) like real ones
        ro['recurrent_hidden_states'] = uniform((T + 1, N, hidden), salt + 9, -0.9, 0.9)
    return ro


def make_params(shapes, salt, head_gain=0.5):
    """Deterministic weights: U(-1,1)*sqrt(3/fan_in)*gain, small non-zero biases."""
    P = OrderedDict()
    for j, (name, shp) in enumerate(shapes.items()):
        s = salt * 131 + j
        if name.endswith('logstd._bias'):
            P[name] = uniform(shp, s, -0.5, 0.5)
        elif name.endswith('bias') or '.bias_' in name:          # incl. nn.GRU's bias_ih_l0 / bias_hh_l0
            P[name] = uniform(shp, s, -0.1, 0.1)
        else:
            fan_in = int(np.prod(shp[1:]))
            gain = np.sqrt(2.0)
            if name.startswith('dist.') or 'critic_linear' in name:
                gain = head_gain
            P[name] = (uniform(shp, s) * np.float32(np.sqrt(3.0 / fan_in) * gain)).astype(np.float32)
    return P
