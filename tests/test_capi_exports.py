"""CPU: the C-ABI library loads and exports every symbol include/*.h declares (no compute calls)."""
import ctypes
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, 'pytorch-a2c-ppo-acktr-gail_amd', 'lib', 'libacktr_hip.so')


def declared_symbols():
    names = []
    for h in glob.glob(os.path.join(ROOT, 'include', '*.h')):
        txt = open(h).read()
        txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
        names += re.findall(r'\b(acktr_[a-z0-9_]+)\s*\(', txt)
    return sorted(set(names))


def test_library_exports_every_declared_symbol():
    assert os.path.exists(LIB), 'build the library first: python pytorch-a2c-ppo-acktr-gail_amd/csrc/build.py'
    lib = ctypes.CDLL(LIB)
    names = declared_symbols()
    assert len(names) >= 7
    for n in names:
        assert hasattr(lib, n), 'missing export ' + n


def test_python_binding_matches_header():
    from a2c_ppo_acktr import _hip
    import a2c_ppo_acktr.storage  # noqa: F401  (registers nothing extra, import must work on CPU)
    try:
        import a2c_ppo_acktr.model  # noqa: F401
        import a2c_ppo_acktr.algo   # noqa: F401
    except ImportError:
        pass
    L = _hip.lib()
    assert L.acktr_abi_version() == _hip.ABI_VERSION
    assert L.acktr_target_arch() == b'gfx950'
    assert set(declared_symbols()) == set(_hip._PROTOS.keys())


def test_no_cpu_fallback():
    """The product path refuses CPU tensors instead of silently computing elsewhere."""
    import pytest
    import torch
    from a2c_ppo_acktr.storage import RolloutStorage

    class Box(object):
        shape = (2,)
    rs = RolloutStorage(4, 3, (5,), Box(), 1)
    with pytest.raises(RuntimeError):
        rs.compute_returns(torch.zeros(3, 1), True, 0.99, 0.95, True)
