"""GPU: the allocation-free entry point acktr_adv_normalize_noalloc_f32 (include/acktr_hip.h) against the existing
acktr_adv_normalize_f32, straight through the C-ABI.  Same kernels, same fixed-order reduction -> bit-identical.
Two calls in a row share the library's per-GPU scratch array; the second uses a smaller input.  Collected last
(file name), so a failure here cannot hide another test under `pytest -x`."""
import gc

import pytest
import torch

from a2c_ppo_acktr import _hip

pytestmark = pytest.mark.gpu


def _both(n, seed):
    L = _hip.lib()
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(seed)
    ret = torch.randn(n, generator=g).to(dev)
    val = torch.randn(n, generator=g).to(dev)
    stream = _hip.cur_stream(dev)
    ws = torch.empty(L.acktr_adv_normalize_workspace_bytes(), dtype=torch.uint8, device=dev)
    want = torch.full((n,), float('nan'), device=dev)
    got = torch.full((n,), float('nan'), device=dev)
    _hip.check(L.acktr_adv_normalize_f32(_hip.dev_ptr(ret), _hip.dev_ptr(val), _hip.dev_ptr(want), n,
                                         _hip.dev_ptr(ws), stream), 'acktr_adv_normalize_f32')
    torch.cuda.synchronize()
    # hipMemGetInfo sees every allocation on the device, the library's own hipMalloc / hipFree included (torch's
    # memory_allocated() only counts torch's caching allocator)
    # (Python's cyclic collector is kept out of the measured region: a collection that happens to run there frees dead
    #  tensors of EARLIER tests and moves both counters)
    gc.collect()
    gc_was_on = gc.isenabled()
    gc.disable()
    try:
        free_before, _ = torch.cuda.mem_get_info(dev)
        before = torch.cuda.memory_allocated(dev)
        _hip.check(L.acktr_adv_normalize_noalloc_f32(_hip.dev_ptr(ret), _hip.dev_ptr(val), _hip.dev_ptr(got), n, stream),
                   'acktr_adv_normalize_noalloc_f32')
        torch.cuda.synchronize()
        assert torch.cuda.memory_allocated(dev) == before
        assert torch.cuda.mem_get_info(dev)[0] == free_before, 'device memory was allocated or freed inside the call'
    finally:
        if gc_was_on:
            gc.enable()
    return want, got


def test_noalloc_adv_normalize_matches_workspace_form_twice():
    # first call: enough elements for every one of the 1024 partial slots to be used (n > 1024 * 2048)
    want, got = _both(128 * 32768 + 13, 1)
    assert torch.isfinite(got).all()
    assert torch.equal(want.view(torch.int32), got.view(torch.int32))
    # second call, smaller input: only 3 partial slots are live, the rest still holds the first call's values
    want, got = _both(128 * 32 + 5, 2)
    assert torch.isfinite(got).all()
    assert torch.equal(want.view(torch.int32), got.view(torch.int32))
    # and the smallest legal one
    want, got = _both(2, 3)
    assert torch.equal(want.view(torch.int32), got.view(torch.int32))


def test_noalloc_under_graph_capture():
    """The use case the header names: the entry point recorded into a hipGraph (after acktr_adv_normalize_noalloc_init
    outside the capture) and replayed twice on changing inputs, against the workspace form."""
    L = _hip.lib()
    dev = torch.device('cuda:0')
    assert L.acktr_adv_normalize_noalloc_init() == 0
    n = 128 * 300 + 7
    g0 = torch.Generator().manual_seed(5)
    ret, val = torch.randn(n, generator=g0).to(dev), torch.randn(n, generator=g0).to(dev)
    got = torch.full((n,), float('nan'), device=dev)
    ws = torch.empty(L.acktr_adv_normalize_workspace_bytes(), dtype=torch.uint8, device=dev)
    graph = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(graph):
        _hip.check(L.acktr_adv_normalize_noalloc_f32(_hip.dev_ptr(ret), _hip.dev_ptr(val), _hip.dev_ptr(got), n,
                                                     _hip.cur_stream(dev)), 'acktr_adv_normalize_noalloc_f32 (capture)')
    for k in range(2):
        ret.copy_(torch.randn(n, generator=g0))
        val.copy_(torch.randn(n, generator=g0))
        got.fill_(float('nan'))
        free_before, _ = torch.cuda.mem_get_info(dev)
        graph.replay()
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info(dev)[0] == free_before
        want = torch.empty(n, device=dev)
        _hip.check(L.acktr_adv_normalize_f32(_hip.dev_ptr(ret), _hip.dev_ptr(val), _hip.dev_ptr(want), n, _hip.dev_ptr(ws),
                                             _hip.cur_stream(dev)), 'acktr_adv_normalize_f32')
        torch.cuda.synchronize()
        assert torch.equal(want.view(torch.int32), got.view(torch.int32)), k


def test_noalloc_rejects_oversize_and_bad_arguments():
    L = _hip.lib()
    dev = torch.device('cuda:0')
    x = torch.zeros(8, device=dev)
    mx = L.acktr_adv_normalize_noalloc_max_n()
    assert mx == 1 << 28
    rc = L.acktr_adv_normalize_noalloc_f32(_hip.dev_ptr(x), _hip.dev_ptr(x), _hip.dev_ptr(x), mx + 1,
                                           _hip.cur_stream(dev))
    assert rc == -1 and b'exceeds the maximum' in L.acktr_last_error()
    rc = L.acktr_adv_normalize_noalloc_f32(_hip.dev_ptr(x), _hip.dev_ptr(x), _hip.dev_ptr(x), 1, _hip.cur_stream(dev))
    assert rc == -1
    rc = L.acktr_adv_normalize_noalloc_f32(None, _hip.dev_ptr(x), _hip.dev_ptr(x), 8, _hip.cur_stream(dev))
    assert rc == -1
