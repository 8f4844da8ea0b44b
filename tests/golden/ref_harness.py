"""Import the reference implementation in the BUILD CONTAINER ONLY.

Used solely by tests/golden/make_golden.py to produce the committed golden
vectors.  /root/reference does not exist on the GPU box; nothing in tests/,
bench.py or smoke() imports this module.

The reference's hot path needs gym / stable_baselines3 / h5py only at import
time (a2c_ppo_acktr/utils.py:7 -> envs.py:3-17, algo/gail.py:1,9), so empty stub
modules exposing the imported names are registered first; ``torch.symeig``
(removed from torch, kfac.py:208-211) is shimmed with ``torch.linalg.eigh``.
No reference file is modified or copied.
"""
import sys
import types

import torch

REF = '/root/reference'


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def install():
    if 'a2c_ppo_acktr' in sys.modules and getattr(sys.modules['a2c_ppo_acktr'], '__file__', '').startswith(REF):
        return

    class _Dummy(object):
        def __init__(self, *a, **k):
            pass

    # sb3 is not installed and its source is not in the container: the published algorithm restated in
    # oracle/gail_ref.py stands in (predict_reward's normalisation is therefore "parity unpinned")
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from oracle.gail_ref import RunningMeanStd

    gym = _stub('gym', Wrapper=_Dummy, ObservationWrapper=_Dummy, Env=_Dummy)
    spaces = _stub('gym.spaces', Box=_Dummy)
    gym.spaces = spaces
    _stub('gym.spaces.box', Box=_Dummy)
    _stub('gym.wrappers')
    _stub('gym.wrappers.clip_action', ClipAction=_Dummy)
    _stub('stable_baselines3')
    _stub('stable_baselines3.common')
    _stub('stable_baselines3.common.atari_wrappers', ClipRewardEnv=_Dummy, EpisodicLifeEnv=_Dummy,
          FireResetEnv=_Dummy, MaxAndSkipEnv=_Dummy, NoopResetEnv=_Dummy, WarpFrame=_Dummy)
    _stub('stable_baselines3.common.monitor', Monitor=_Dummy)
    _stub('stable_baselines3.common.vec_env', DummyVecEnv=_Dummy, SubprocVecEnv=_Dummy, VecEnvWrapper=_Dummy)
    _stub('stable_baselines3.common.vec_env.vec_normalize', VecNormalize=_Dummy)
    _stub('stable_baselines3.common.running_mean_std', RunningMeanStd=RunningMeanStd)
    _stub('h5py')
    if not hasattr(torch, 'symeig') or True:
        torch.symeig = lambda A, eigenvectors=False, upper=True: torch.linalg.eigh(A, UPLO='U')
    for k in [k for k in sys.modules if k == 'a2c_ppo_acktr' or k.startswith('a2c_ppo_acktr.')]:
        del sys.modules[k]
    sys.path.insert(0, REF)


class Discrete(object):
    """Duck-typed action space: the reference dispatches on the class NAME (model.py:30-38)."""
    def __init__(self, n):
        self.n = n


class Box(object):
    def __init__(self, dim):
        self.shape = (dim,)
