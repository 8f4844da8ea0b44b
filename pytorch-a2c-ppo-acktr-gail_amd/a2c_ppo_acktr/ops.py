"""Tensor-level wrappers over the C-ABI operators (include/acktr_hip.h).

Thin plumbing only: shape checks on the host (a wrong shape must never reach a
kernel), device pointers + the current stream handed to the library.  Every
function requires fp32 contiguous GPU tensors and raises otherwise.
"""
import contextlib
import ctypes
import gc
import torch

from . import _hip

f32 = torch.float32
_ws_cache = {}
WS_MIN_BYTES = 1 << 20
_ws_retired = []          # outgrown buffers, kept alive for captured graphs that still point into them

# Optional per-launch timing with HIP events on the launch stream (used by bench.py for the roofline
# figures).  `profiler` is None (off) or a dict name -> list of (start_event, end_event, work) tuples.
profiler = None


class _Timed(object):
    __slots__ = ('name', 'work', 'ev')

    def __init__(self, name, work):
        self.name, self.work = name, work

    def __enter__(self):
        if profiler is not None:
            s = torch.cuda.Event(enable_timing=True)
            s.record()
            self.ev = s
        return self

    def __exit__(self, *a):
        if profiler is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            profiler.setdefault(self.name, []).append((self.ev, e, self.work))
        return False


def profile_summary():
    """name -> dict(calls, avg_ms, total_ms, flops_per_call) after a device sync."""
    out = {}
    if profiler is None:
        return out
    torch.cuda.synchronize()
    for name, recs in profiler.items():
        ms = [s.elapsed_time(e) for s, e, _ in recs]
        # an event bracket also counts GPU idle time when the host is late with the launch inside it (one 35 ms host
        # stall once turned a 78 us kernel into a 2.2 ms "average"): brackets above 3x the median are left out
        med = sorted(ms)[len(ms) // 2]
        kept = [m for m in ms if m <= 3.0 * med] or ms
        out[name] = dict(calls=len(ms), avg_ms=sum(kept) / len(kept), total_ms=sum(kept) / len(kept) * len(ms),
                         flops_per_call=sum(w for _, _, w in recs) / len(recs), outliers_dropped=len(ms) - len(kept))
    return out


def workspace(device, nbytes, tag='ws'):
    """Per-(device, tag) scratch buffer, grown on demand and reused across calls.

    A buffer that is outgrown is RETIRED, never freed: hipGraphs captured earlier (PPO / A2C / ACKTR / GAIL steps)
    hold raw pointers into it and keep replaying while another agent with larger shapes makes this cache grow.
    Growth is at least 2x, so the retired buffers of a tag add up to less than its live one."""
    key = (str(device), tag)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        if buf is not None:
            _ws_retired.append(buf)
        want = max(int(nbytes), WS_MIN_BYTES, 2 * buf.numel() if buf is not None else 0)
        buf = torch.empty(want, dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    # exactly the bytes asked for: several entry points cut their reduction into as many split-K slabs as the workspace
    # they are GIVEN holds, so handing over the whole (grown) buffer made results depend on what had run before in the
    # process -- by one rounding, which a ReLU gate at zero can turn into another trajectory (tests/test_u8_gpu.py)
    return buf[:nbytes] if nbytes < buf.numel() else buf


@contextlib.contextmanager
def graph_capture(g, **kw):
    """`with torch.cuda.graph(g)` with Python's cyclic garbage collector OFF for the duration.  torch collects once on
    entry, but a collection that the allocation counters trigger DURING the capture can still finalise an unreachable
    torch.cuda.CUDAGraph (a dropped agent's, kept alive until then by a reference cycle): its destructor calls into HIP,
    which answers "operation not permitted when stream is capturing", and torch aborts the process from the destructor
    (seen once in the GPU suite, inside the GAIL step's capture)."""
    was_enabled = gc.isenabled()
    gc.disable()                # (torch.cuda.graph collects explicitly on entry; a second full collection here cost the
                                #  bench's ACKTR loop, whose capture falls into its timed updates, 110 ms)
    try:
        with torch.cuda.graph(g, **kw):
            yield g
    finally:
        if was_enabled:
            gc.enable()


def _p(t, name, dtype=f32):
    return _hip.dev_ptr(t, dtype, name)


def _conv_out(H, K, S):
    return (H - K) // S + 1


def _pack_layout(layers, B):
    """[(key, byte offset, bytes, job tuple)] of the weight-image workspaces inside ONE buffer (256-byte aligned slots),
    and the buffer size.  Layers the stationary kernels do not cover are left out."""
    L = _hip.lib()
    out, off = [], 0
    for key, w, stride, hw, dgrad in layers:
        OC, C, KH, KW = w.shape
        if dgrad:
            nb = L.acktr_conv2d_dgrad_workspace_bytes(B, C, hw, hw, OC, KH, KW, stride)
        else:
            nb = L.acktr_conv2d_fwd_workspace_bytes(C, hw, hw, OC, KH, KW, stride)
        if not nb or (w.data_ptr() & 15):
            continue
        out.append((key, off, int(nb), (C, hw, OC, KH, KW, stride, 1 if dgrad else 0)))
        off += (int(nb) + 255) // 256 * 256
    return out, off


def _pack_jobs(layout, weights, buf):
    jobs = [_hip.ConvPackJob(weights[key].data_ptr(), buf.data_ptr() + off, nb, C, hw, hw, OC, KH, KW, stride, dg)
            for key, off, nb, (C, hw, OC, KH, KW, stride, dg) in layout]
    return (_hip.ConvPackJob * len(jobs))(*jobs)


def _run_pack(layout, weights, buf, stream_dev):
    arr = _pack_jobs(layout, weights, buf)
    return _hip.lib().acktr_conv2d_pack_weights_f32(arr, len(arr), _hip.cur_stream(stream_dev))


def pack_conv_weights(layers, B, cache, trust=False):
    """Re-ordered weights of several conv layers in ONE launch.  layers: list of (key, w, stride, in_hw, dgrad); the
    result maps key -> workspace tensor to hand to conv2d_fwd / conv2d_dgrad as `packed`.  Layers the stationary
    kernels do not cover are left out (those entry points then prepare their weights themselves).  `cache` (a dict
    owned by the policy) keeps the workspaces: they must survive from a forward to its backward, so they cannot be
    the process-wide scratch buffers.  All workspaces are slots of ONE buffer, so that an optimiser can keep the images
    in step with the parameters through one offset table (conv_image_map).
    trust=True: the images are known to be current (the caller re-packed at the start of its update and its optimiser
    steps write the images themselves): no launch."""
    if not layers:
        return {}
    dev = layers[0][1].device
    wptr = layers[0][1].data_ptr()
    if cache.get('_jobs') is not None and cache.get('_B', 0) >= B and cache.get('_wptr') == wptr and cache['_buf'].device == dev:
        # the common case, without the host work of laying the buffer out again (it showed as GPU idle time in front of
        # every update's first graph launch): same parameters, a batch the buffer already covers
        if not trust:
            if _hip.lib().acktr_conv2d_pack_weights_f32(cache['_jobs'], len(cache['_jobs']), _hip.cur_stream(dev)) != 0:
                return {}
        return cache['_views']
    # sized for the largest batch seen so far: a smaller batch (Policy.act between two updates) uses the front of the same
    # slots, so captured graphs that hold pointers into the buffer stay valid; an outgrown buffer is retired, not freed
    B = max(int(B), cache.get('_B', 0) if cache.get('_buf') is not None and cache['_buf'].device == dev else 0)
    layout, total = _pack_layout(layers, B)
    if not layout:
        return {}
    sig = tuple((k, o, n) for k, o, n, _ in layout)
    buf = cache.get('_buf')
    fresh = buf is None or buf.device != dev or cache.get('_sig') != sig
    if fresh:
        if buf is not None:
            _ws_retired.append(buf)
        cache['_B'] = B
        # zeroed: behind the weight image a dgrad workspace holds the zero-padded dY planes, whose borders nobody
        # writes when the producer of dY fills the interiors itself (padded_dy_view / conv2d_dgrad(dy_prepadded))
        buf = torch.zeros(total, dtype=torch.uint8, device=dev)
        cache['_buf'], cache['_sig'], cache['_map'] = buf, sig, None
        cache['_views'] = {k: buf[o:o + n] for k, o, n, _ in layout}
    weights = {key: w for key, w, _, _, _ in layers}
    if fresh or not trust:
        if _run_pack(layout, weights, buf, dev) != 0:      # a layer outside the stationary kernels' shapes
            return {}
    cache['_jobs'] = _pack_jobs(layout, weights, buf)
    cache['_wptr'] = wptr
    return cache['_views']


def conv_image_map(layers, B, cache, offset_of):
    """_hip.PackMap for an optimiser that stores conv weights to their weight-image slots as it updates them
    (acktr_clip_adam_dev_pack_f32), or None.  offset_of(w) -> float offset of parameter w in the flat buffer.  The
    tables come from the pack kernel itself: it is run once on weights that hold their own element number, and every
    image slot then says which parameter element it wants."""
    if cache.get('_map') is not None and cache.get('_jobs') is not None and cache.get('_B', 0) >= B:
        return cache['_map'][0]
    if not pack_conv_weights(layers, B, cache, trust=True):
        return None
    if cache.get('_map') is not None:
        return cache['_map'][0]
    dev = cache['_buf'].device
    layout, total = _pack_layout(layers, cache['_B'])
    by_key = {key: w for key, w, _, _, _ in layers}
    lo = min(offset_of(w) for w in by_key.values())
    hi = max(offset_of(w) + w.numel() for w in by_key.values())
    idx_w = {key: (torch.arange(1, w.numel() + 1, dtype=torch.float32, device=dev).view(w.shape).contiguous())
             for key, w in by_key.items()}
    if max(w.numel() for w in by_key.values()) >= (1 << 24):
        return None
    tmp = torch.zeros(total, dtype=torch.uint8, device=dev)
    if _run_pack(layout, idx_w, tmp, dev) != 0:
        return None
    img = tmp.view(torch.float32)
    t = [torch.full((hi - lo,), -1, dtype=torch.int32, device=dev) for _ in range(2)]
    for key, off, nb, job in layout:
        region = img[off // 4:(off + nb) // 4]
        slots = region.nonzero().view(-1)
        elems = region[slots].long() - 1 + (offset_of(by_key[key]) - lo)
        which = t[1] if job[-1] else t[0]
        if bool((which[elems] >= 0).any()):
            return None                        # two images of one kind want the same parameter: not a layout this map covers
        which[elems] = (slots + off // 4).to(torch.int32)
    pm = _hip.PackMap(lo, hi, t[0].data_ptr(), t[1].data_ptr(), cache['_buf'].data_ptr())
    cache['_map'] = (pm, t)                    # (the tables stay alive with the cache)
    return pm


def gate_mask_like(B, OC, OH, OW, device):
    """Buffer for the ReLU gate of a [B,OC,OH,OW] conv output as bits (acktr_conv2d_gate_mask_bytes)."""
    nb = _hip.lib().acktr_conv2d_gate_mask_bytes(B, OC, OH, OW)
    return torch.empty(nb // 4, dtype=torch.int32, device=device)


def conv2d_fwd(x, w, bias, stride, in_scale=1.0, relu=True, gather_idx=None, out=None, packed=None, gate_mask=None):
    """y = relu?(in_scale*conv2d(x[gather_idx], w) + bias); x [Bsrc,C,H,W], w [OC,C,KH,KW].
    gate_mask (int32 words, gate_mask_like): also receives (y > 0) as bits for conv2d_dgrad(gate_bits=) of the next
    layer -- the 8x8 stride-4 layer on its stationary path only."""
    OC, C, KH, KW = w.shape
    H, W = x.shape[-2], x.shape[-1]
    assert x.shape[-3] == C, 'channel mismatch'
    B = x.shape[0] if gather_idx is None else gather_idx.numel()
    OH, OW = _conv_out(H, KH, stride), _conv_out(W, KW, stride)
    if out is None:
        out = torch.empty(B, OC, OH, OW, dtype=f32, device=x.device)
    assert out.shape == (B, OC, OH, OW)
    L = _hip.lib()
    nb = L.acktr_conv2d_fwd_workspace_bytes(C, H, W, OC, KH, KW, stride)
    ws = packed if packed is not None else (workspace(x.device, nb, tag='convfwd') if nb else None)
    if x.dtype == torch.uint8:             # frames kept as bytes (RolloutStorage(obs_dtype=torch.uint8)): conv1 only
        if (C, H, W, OC, KH, KW, stride) != (4, 84, 84, 32, 8, 8, 4) or ws is None:
            raise RuntimeError('uint8 inputs are supported for the first NatureCNN layer only (4x84x84, 32 filters 8x8 s4)')
        with _Timed('conv%ds%d_fwd' % (KH, stride), 2.0 * B * OH * OW * OC * C * KH * KW):
            args = (_p(x, 'x', torch.uint8), _p(gather_idx, 'gather_idx', torch.int64), _p(w, 'w'),
                    _p(bias, 'bias'), _p(out, 'y'), B, float(in_scale), 1 if relu else 0,
                    _p(ws, 'workspace', torch.uint8), ws.numel(), 1 if packed is not None else 0)
            if gate_mask is not None:
                assert gate_mask.numel() * 4 >= L.acktr_conv2d_gate_mask_bytes(B, OC, OH, OW)
                rc = L.acktr_conv1_fwd_u8_gate(*args, _p(gate_mask, 'gate_mask', torch.int32), _hip.cur_stream(x.device))
            else:
                rc = L.acktr_conv1_fwd_u8(*args, _hip.cur_stream(x.device))
        _hip.check(rc, 'acktr_conv1_fwd_u8')
        return out
    with _Timed('conv%ds%d_fwd' % (KH, stride), 2.0 * B * OH * OW * OC * C * KH * KW):
        args = (_p(x, 'x'), _p(gather_idx, 'gather_idx', torch.int64), _p(w, 'w'), _p(bias, 'bias'), _p(out, 'y'),
                B, C, H, W, OC, KH, KW, stride, float(in_scale), 1 if relu else 0,
                _p(ws, 'workspace', torch.uint8), ws.numel() if ws is not None else 0, 1 if packed is not None else 0)
        if gate_mask is not None:
            assert gate_mask.numel() * 4 >= L.acktr_conv2d_gate_mask_bytes(B, OC, OH, OW)
            rc = L.acktr_conv2d_fwd_gate_f32(*args, _p(gate_mask, 'gate_mask', torch.int32), _hip.cur_stream(x.device))
        else:
            rc = L.acktr_conv2d_fwd_f32(*args, _hip.cur_stream(x.device))
    _hip.check(rc, 'acktr_conv2d_fwd_f32')
    return out


def conv2d_wgrad(x, dy, w_shape, stride, dw, db=None, in_scale=1.0, gather_idx=None, pending=None):
    """pending (a _hip.SlabFin): the final slab sum may be left to the optimiser's norm launch
    (acktr_conv2d_wgrad_deferred_f32); pending.dst is None afterwards when the call completed on its own.  The slabs
    live in the shared workspace: nothing else may use it before that optimiser call."""
    OC, C, KH, KW = w_shape
    H, W = x.shape[-2], x.shape[-1]
    B = dy.shape[0]
    assert dy.shape == (B, OC, _conv_out(H, KH, stride), _conv_out(W, KW, stride))
    assert dw.numel() == OC * C * KH * KW and (db is None or db.numel() == OC)
    if gather_idx is None:
        assert x.shape[0] == B
    else:
        assert gather_idx.numel() == B
    L = _hip.lib()
    if x.dtype == torch.uint8:
        if (C, H, W, OC, KH, KW, stride) != (4, 84, 84, 32, 8, 8, 4):
            raise RuntimeError('uint8 inputs are supported for the first NatureCNN layer only (4x84x84, 32 filters 8x8 s4)')
        ws = workspace(x.device, L.acktr_conv1_wgrad_u8_workspace_bytes(B))
        with _Timed('conv%ds%d_wgrad' % (KH, stride), 2.0 * dy[0].numel() * B * C * KH * KW):
            rc = L.acktr_conv1_wgrad_u8(_p(x, 'x', torch.uint8), _p(gather_idx, 'gather_idx', torch.int64), _p(dy, 'dy'),
                                        _p(dw, 'dw'), _p(db, 'db'), B, float(in_scale), _p(ws, 'workspace', torch.uint8),
                                        ws.numel(), ctypes.byref(pending) if pending is not None else None,
                                        _hip.cur_stream(x.device))
        _hip.check(rc, 'acktr_conv1_wgrad_u8')
        return
    nbytes = L.acktr_conv2d_wgrad_workspace_bytes(B, C, H, W, OC, KH, KW, stride)
    ws = workspace(x.device, nbytes)
    with _Timed('conv%ds%d_wgrad' % (KH, stride), 2.0 * dy[0].numel() * B * C * KH * KW):
        if pending is not None:
            rc = L.acktr_conv2d_wgrad_deferred_f32(
                _p(x, 'x'), _p(gather_idx, 'gather_idx', torch.int64), _p(dy, 'dy'), _p(dw, 'dw'), _p(db, 'db'),
                B, C, H, W, OC, KH, KW, stride, float(in_scale), _p(ws, 'workspace', torch.uint8), ws.numel(),
                ctypes.byref(pending), _hip.cur_stream(x.device))
        else:
            rc = L.acktr_conv2d_wgrad_f32(
                _p(x, 'x'), _p(gather_idx, 'gather_idx', torch.int64), _p(dy, 'dy'), _p(dw, 'dw'), _p(db, 'db'),
                B, C, H, W, OC, KH, KW, stride, float(in_scale), _p(ws, 'workspace', torch.uint8), ws.numel(),
                _hip.cur_stream(x.device))
    _hip.check(rc, 'acktr_conv2d_wgrad_f32')


def padded_dy_view(packed, B, w, stride, in_hw):
    """The zero-padded dY region ([B, OC, 11, 11] fp32) inside a stationary-dgrad workspace from pack_conv_weights, or
    None.  A producer of that layer's dY (linear_dgrad / conv2d_dgrad with padded_out=) writes the interiors there and
    conv2d_dgrad(..., dy_prepadded=True) then skips its padding pass; the borders stay zero from the allocation."""
    if packed is None:
        return None
    OC, C, KH, KW = w.shape
    off = _hip.lib().acktr_conv2d_dgrad_padded_dy_offset_bytes(C, in_hw, in_hw, OC, KH, KW, stride)
    n = B * OC * 121
    if off < 0 or off % 4 or packed.numel() < off + 4 * n:
        return None
    return packed[off:off + 4 * n].view(f32).view(B, OC, 11, 11)


def conv2d_dgrad(dy, w, x_act, stride, out=None, packed=None, dy_prepadded=False, padded_out=None, carry=None,
                 gate_bits=None):
    """dx = conv_transpose(dy, w) * (x_act > 0); x_act [B,C,H,W] is the (post-ReLU) input of the layer.
    dy_prepadded: `packed` already holds this dY zero-padded (see padded_dy_view); padded_out [B,C,H+2p,W+2p]: dx is
    also written there, interior only.  carry (a _hip.SlabFin left pending by conv2d_wgrad(pending=...)): that weight
    gradient's slab sum rides in this launch (acktr_conv2d_dgrad_carry_f32).
    gate_bits (int32 words written by conv2d_fwd(gate_mask=) of the layer that produced x_act): read instead of x_act."""
    OC, C, KH, KW = w.shape
    B, _, OH, OW = dy.shape
    H, W = (OH - 1) * stride + KH, (OW - 1) * stride + KW
    if out is None:
        out = torch.empty(B, C, H, W, dtype=f32, device=dy.device)
    assert out.shape == (B, C, H, W) and (x_act is None or x_act.shape == out.shape)
    pad = 0
    if padded_out is not None:
        pad = (padded_out.shape[-1] - W) // 2
        assert padded_out.shape == (B, C, H + 2 * pad, W + 2 * pad) and pad > 0
    assert not dy_prepadded or packed is not None
    L = _hip.lib()
    ws = packed if packed is not None else \
        workspace(dy.device, L.acktr_conv2d_dgrad_workspace_bytes(B, C, H, W, OC, KH, KW, stride), tag='dgrad')
    with _Timed('conv%ds%d_dgrad' % (KH, stride), 2.0 * dy.numel() * C * KH * KW):
        gate = _p(x_act, 'x_act')
        if gate_bits is not None:
            assert gate_bits.numel() * 4 >= L.acktr_conv2d_gate_mask_bytes(B, C, H, W)
            gate = _p(gate_bits, 'gate_bits', torch.int32)
        rc = L.acktr_conv2d_dgrad_carry_f32(_p(dy, 'dy'), _p(w, 'w'), gate, _p(out, 'dx'),
                                            B, C, H, W, OC, KH, KW, stride, _p(ws, 'workspace', torch.uint8), ws.numel(),
                                            (1 if packed is not None else 0) | (2 if dy_prepadded else 0)
                                            | (4 if gate_bits is not None else 0),
                                            _p(padded_out, 'dx_padded'), pad,
                                            ctypes.byref(carry) if (carry is not None and carry.dst) else None,
                                            _hip.cur_stream(dy.device))
    _hip.check(rc, 'acktr_conv2d_dgrad_carry_f32')
    return out


ACT = {None: 0, 'none': 0, 'relu': 1, 'tanh': 2}


def linear_fwd(x, w, bias, act=None, out=None):
    B, IN = x.shape
    OUT = w.shape[0]
    assert w.shape[1] == IN
    if out is None:
        out = torch.empty(B, OUT, dtype=f32, device=x.device)
    L = _hip.lib()
    nb = L.acktr_linear_fwd_workspace_bytes(B, IN, OUT)
    ws = workspace(x.device, nb) if nb else None
    with _Timed('linear%dx%d_fwd' % (IN, OUT), 2.0 * B * IN * OUT):
        rc = L.acktr_linear_fwd_f32(_p(x, 'x'), _p(w, 'w'), _p(bias, 'bias'), _p(out, 'y'), B, IN, OUT, ACT[act],
                                    _p(ws, 'workspace', torch.uint8), ws.numel() if ws is not None else 0,
                                    _hip.cur_stream(x.device))
    _hip.check(rc, 'acktr_linear_fwd_f32')
    return out


def linear_fwd_slabs(x, w):
    """x w^T as split-K slabs (no bias, no activation, no reduction): returns (workspace, nslabs, slab stride in floats).
    The slabs live in a scratch buffer of their own tag: valid until the next call on this device."""
    B, IN = x.shape
    OUT = w.shape[0]
    assert w.shape[1] == IN
    L = _hip.lib()
    ws = workspace(x.device, L.acktr_linear_fwd_slabs_workspace_bytes(B, IN, OUT), tag='fcslabs')
    ns = ctypes.c_int(0)
    with _Timed('linear%dx%d_fwd' % (IN, OUT), 2.0 * B * IN * OUT):
        rc = L.acktr_linear_fwd_slabs_f32(_p(x, 'x'), _p(w, 'w'), B, IN, OUT, _p(ws, 'workspace', torch.uint8), ws.numel(),
                                          ctypes.byref(ns), _hip.cur_stream(x.device))
    _hip.check(rc, 'acktr_linear_fwd_slabs_f32')
    return ws, ns.value, B * OUT


def linear_dgrad(dy, w, gate_src=None, gate=None, out=None, accumulate=False, padded_out=None, planes=None):
    """padded_out [B, C, OH+2p, OW+2p] with planes=(C, OH, OW): dx is also written there as zero-padded planes
    (interior only) -- the dY layout of the conv dgrad that follows an FC layer."""
    B, OUT = dy.shape
    IN = w.shape[1]
    assert w.shape[0] == OUT
    if out is None:
        assert not accumulate
        out = torch.empty(B, IN, dtype=f32, device=dy.device)
    C = OH = OW = pad = 0
    if padded_out is not None:
        C, OH, OW = planes
        pad = (padded_out.shape[-1] - OW) // 2
        assert C * OH * OW == IN and pad > 0 and padded_out.shape == (B, C, OH + 2 * pad, OW + 2 * pad)
    with _Timed('linear%dx%d_dgrad' % (IN, OUT), 2.0 * B * IN * OUT):
        rc = _hip.lib().acktr_linear_dgrad_ex_f32(_p(dy, 'dy'), _p(w, 'w'), _p(gate_src, 'gate_src'), ACT[gate],
                                                  _p(out, 'dx'), B, IN, OUT, 1 if accumulate else 0,
                                                  _p(padded_out, 'dx_padded'), C, OH, OW, pad,
                                                  _hip.cur_stream(dy.device))
    _hip.check(rc, 'acktr_linear_dgrad_ex_f32')
    return out


def linear_wgrad(dy, x, dw, db=None):
    B, OUT = dy.shape
    IN = x.shape[1]
    assert x.shape[0] == B and dw.numel() == OUT * IN and (db is None or db.numel() == OUT)
    L = _hip.lib()
    ws = workspace(dy.device, L.acktr_linear_wgrad_workspace_bytes(B, IN, OUT))
    with _Timed('linear%dx%d_wgrad' % (IN, OUT), 2.0 * B * IN * OUT):
        rc = L.acktr_linear_wgrad_f32(_p(dy, 'dy'), _p(x, 'x'), _p(dw, 'dw'), _p(db, 'db'), B, IN, OUT,
                                      _p(ws, 'workspace', torch.uint8), ws.numel(), _hip.cur_stream(dy.device))
    _hip.check(rc, 'acktr_linear_wgrad_f32')


def gemm(a, b, out=None, trans_a=False, trans_b=False, alpha=1.0, accumulate=False):
    """out (=|+=) alpha * op(a) @ op(b) for 2-D contiguous fp32 GPU tensors (op = transpose if trans_*)."""
    M, K = (a.shape[1], a.shape[0]) if trans_a else a.shape
    K2, N = (b.shape[1], b.shape[0]) if trans_b else b.shape
    assert K == K2, 'inner dimensions differ'
    if out is None:
        assert not accumulate
        out = torch.empty(M, N, dtype=f32, device=a.device)
    assert out.shape == (M, N) and out.is_contiguous()
    # A(m,k): a[m,k] -> (row stride lda, k stride 1); transposed: a[k,m] -> (row stride 1, k stride lda)
    a_sr, a_sk = (1, a.shape[1]) if trans_a else (a.shape[1], 1)
    # B(n,k) = op(b)[k,n]: b[k,n] -> (row stride 1, k stride ldb); transposed: b[n,k] -> (row stride ldb, k stride 1)
    b_sr, b_sk = (b.shape[1], 1) if trans_b else (1, b.shape[1])
    rc = _hip.lib().acktr_gemm_f32(_p(a, 'a'), a_sr, a_sk, _p(b, 'b'), b_sr, b_sk, _p(out, 'c'), N, M, N, K,
                                   float(alpha), 1 if accumulate else 0, _hip.cur_stream(a.device))
    _hip.check(rc, 'acktr_gemm_f32')
    return out
