"""Forward / backward of the two policy architectures, composed from the C-ABI operators.

CNN  (reference model.py:189-195, distributions.py:71-73):
    obs/255 -> conv8s4+ReLU -> conv4s2+ReLU -> conv3s1+ReLU -> flatten -> FC+ReLU -> fused head
    The critic (hidden->1) and the action head (hidden->A) are ONE [1+A, hidden] GEMM because their
    parameters are adjacent in the flat buffer; `head[:,0]` is the value, `head[:,1:]` the logits.
MLP  (reference model.py:220-229, distributions.py:86-95):
    separate tanh towers for critic and actor, value head on the critic tower, mean head on the actor.

Recurrent bases (model.py:111-166): a GRU between the trunk and the heads (CNN: on the FC features; MLP: on the raw
inputs, the towers then read the GRU output).  `gru_forward` / `gru_backward` below run it over [T*N] rows (row
t*N+n) as one input GEMM for all steps + per step one hidden GEMM and one gate kernel (csrc/gru.hip).

`gather_idx` (int64 device tensor) makes every rollout-side read go through the minibatch index, so
storage.py:127's gathered copies are never made on the NatureCNN path (fused into conv1's loader).
"""
import ctypes
import os

import torch

from . import _hip, ops

f32 = torch.float32


def _wb(mod):
    """(weight, bias) parameters of a layer, whether or not KFAC's SplitBias wraps it (kfac.py:74-84)."""
    if hasattr(mod, 'add_bias') and hasattr(mod, 'module'):
        return mod.module.weight, mod.add_bias._bias
    return mod.weight, mod.bias


def is_cnn(policy):
    return hasattr(policy.base, 'main')


def dist_linear(policy):
    """The head's nn.Linear: `dist.linear` (Categorical: logits) or `dist.fc_mean` (DiagGaussian: mean) -- the reference
    pairs either head with either base (model.py:30-38)."""
    return policy.dist.linear if policy.dist.kind == 0 else policy.dist.fc_mean


def _per_layout(policy, what, build):
    """`build(policy)` once per parameter layout.  Walking the module tree (nn.Module.__getattr__, Sequential indexing) for
    the same twelve parameters eight times per minibatch step was a quarter of the host time of an eager (data-parallel)
    PPO step.  The layout changes exactly when Policy rebuilds its offset table (to(), unpickling, KFAC's split_bias ->
    rebind_flat): the cache is keyed on that table's identity."""
    d = policy.__dict__
    offs = d.get('_offsets')
    c = d.get('_lp_cache')
    if c is None or c[0] is not offs:
        c = d['_lp_cache'] = (offs, {})
    r = c[1].get(what)
    if r is None:
        r = c[1][what] = build(policy)
    return r


def _cnn_layer_params(policy):
    m = policy.base.main
    return {'conv1': _wb(m[0]), 'conv2': _wb(m[2]), 'conv3': _wb(m[4]), 'fc': _wb(m[7]),
            'critic': _wb(policy.base.critic_linear), 'actor': _wb(dist_linear(policy))}


def cnn_layer_params(policy):
    """{'conv1': (weight, bias), ...} of a CNNBase policy (read-only: shared between calls)."""
    return _per_layout(policy, 'cnn', _cnn_layer_params)


def _mlp_layer_params(policy):
    b = policy.base
    return {'a0': _wb(b.actor[0]), 'a2': _wb(b.actor[2]), 'c0': _wb(b.critic[0]), 'c2': _wb(b.critic[2]),
            'critic': _wb(b.critic_linear), 'mean': _wb(dist_linear(policy))}


def mlp_layer_params(policy):
    return _per_layout(policy, 'mlp', _mlp_layer_params)


def logstd_param(policy):
    return policy.dist.logstd._bias


class Acts(object):
    """Activations of one forward pass (kept for the backward and for the KFAC statistics)."""

    def __init__(self, cnn):
        self.cnn = cnn
        self._feat = None
        self._feat_fn = None

    # FC features: materialised on demand when the forward left them as split-K slabs for the fused head step
    # (forward(head=False), FOLD_FC_SLABS) -- tests and diagnostics read them, the update path never does
    @property
    def feat(self):
        if self._feat is None and self._feat_fn is not None:
            self._feat = self._feat_fn()
        return self._feat

    @feat.setter
    def feat(self, v):
        self._feat = v

    def value(self):
        return self.head[:, 0:1] if self.cnn else self.v

    def dist_in(self):
        return self.head[:, 1:] if self.cnn else self.mean

    # raw pointers / leading dimensions for the loss kernels
    def value_ptr_ld(self):
        if self.cnn:
            return self.head.data_ptr(), self.head.shape[1]
        return self.v.data_ptr(), 1

    def dist_ptr_ld(self):
        if self.cnn:
            return self.head.data_ptr() + 4, self.head.shape[1]
        return self.mean.data_ptr(), self.mean.shape[1]


def _check_input(policy, inputs, gather_idx):
    if not inputs.is_cuda:
        raise RuntimeError('policy inputs must be on the MI355X device; this package has no CPU path')
    if next(policy.parameters()).device != inputs.device:
        raise RuntimeError('policy parameters and inputs are on different devices')
    if inputs.dtype != f32 and not (inputs.dtype == torch.uint8 and is_cnn(policy)):
        raise RuntimeError('policy inputs must be float32 (reference storage.py:12, F6); NatureCNN policies also take uint8 '
                           'frames (RolloutStorage(obs_dtype=torch.uint8))')
    if tuple(inputs.shape[1:]) != policy.obs_shape:
        raise RuntimeError('expected inputs of shape [B, %s], got %s' % (policy.obs_shape, tuple(inputs.shape)))


def head_views(policy, flat):
    """[1+A, hidden] weight and [1+A] bias views of the fused CNN head inside a flat buffer."""
    def where(policy):
        L = cnn_layer_params(policy)
        wc, bc = L['critic']
        wa, ba = L['actor']
        return policy._offsets[id(wc)][0], policy._offsets[id(bc)][0], wc.shape[1], wa.shape[0]
    ow, ob, hidden, A = _per_layout(policy, 'head', where)
    return (flat[ow:ow + (1 + A) * hidden].view(1 + A, hidden), flat[ob:ob + 1 + A])


def gru_params(policy):
    g = policy.base.gru
    return g.weight_ih_l0, g.weight_hh_l0, g.bias_ih_l0, g.bias_hh_l0


def gru_forward(policy, a, x_seq, hxs, masks):
    """model.py:111-166.  x_seq [T*N, I] (row t*N+n), hxs [N, H], masks [T*N(,1)].  The carried state is multiplied by
    masks[t] at every step -- the same arithmetic as the reference's chunking, whose skipped multiplications are by
    exactly 1.0.  Leaves on `a`: gru_out [T*N, H], hxs_out [N, H] (view of the last step) and what BPTT needs
    (gru_rzn = r,z,n per row; gru_gh = W_hh products; gru_hm = masked previous states; gru_x, gru_masks)."""
    if hxs is None or masks is None:
        raise RuntimeError('a recurrent policy needs rnn_hxs and masks')
    dev = x_seq.device
    w_ih, w_hh, b_ih, b_hh = gru_params(policy)
    H = w_hh.shape[1]
    hxs = hxs.to(dev, f32).reshape(-1, H).contiguous()
    N = hxs.shape[0]
    B = x_seq.shape[0]
    T = B // N
    if T * N != B or T < 1:
        raise RuntimeError('recurrent forward: %d input rows are not a multiple of the %d hidden-state rows' % (B, N))
    masks = masks.to(dev, f32).reshape(-1).contiguous()
    if masks.numel() != B:
        raise RuntimeError('recurrent forward: expected %d mask entries, got %d' % (B, masks.numel()))
    L = _hip.lib()
    stream = _hip.cur_stream(dev)
    gi = ops.linear_fwd(x_seq, w_ih, b_ih, None)                       # all steps at once: [T*N, 3H]
    gh = torch.empty(B, 3 * H, dtype=f32, device=dev)
    hm = torch.empty(B, H, dtype=f32, device=dev)
    out = torch.empty(B, H, dtype=f32, device=dev)
    _hip.check(L.acktr_gru_mask_rows_f32(_hip.dev_ptr(hxs), _hip.dev_ptr(masks), _hip.dev_ptr(hm), N, H, stream),
               'acktr_gru_mask_rows_f32')
    # few rows per step (a minibatch of whole trajectories): product + gates in ONE launch per step; raw pointers, the
    # loop body is one foreign call (as two launches behind torch slicing a step cost ~70 us of host time)
    fused = GRU_STEP_FUSED and L.acktr_gru_step_supported(N, H) == 1
    p_gi, p_gh, p_hm, p_out, p_m = gi.data_ptr(), gh.data_ptr(), hm.data_ptr(), out.data_ptr(), masks.data_ptr()
    p_w, p_b = _hip.dev_ptr(w_hh), _hip.dev_ptr(b_hh)
    vp = ctypes.c_void_p
    for t in range(T):
        nxt = t + 1 < T
        if fused:
            rc = L.acktr_gru_step_fwd_f32(vp(p_gi + 12 * H * N * t), vp(p_gh + 12 * H * N * t), vp(p_hm + 4 * H * N * t), p_w, p_b,
                                          vp(p_m + 4 * N * (t + 1)) if nxt else None, vp(p_out + 4 * H * N * t),
                                          vp(p_hm + 4 * H * N * (t + 1)) if nxt else None, N, H, stream)
            if rc:
                _hip.check(rc, 'acktr_gru_step_fwd_f32')
            continue
        s = slice(t * N, (t + 1) * N)
        ops.linear_fwd(hm[s], w_hh, b_hh, None, out=gh[s])
        rc = L.acktr_gru_gates_fwd_f32(_hip.dev_ptr(gi[s]), _hip.dev_ptr(gh[s]), _hip.dev_ptr(hm[s]),
                                       _hip.dev_ptr(masks[(t + 1) * N:(t + 2) * N]) if nxt else None,
                                       _hip.dev_ptr(out[s]),
                                       _hip.dev_ptr(hm[(t + 1) * N:(t + 2) * N]) if nxt else None, N, H, stream)
        _hip.check(rc, 'acktr_gru_gates_fwd_f32')
    a.gru_x, a.gru_masks, a.gru_rzn, a.gru_gh, a.gru_hm = x_seq, masks, gi, gh, hm
    a.gru_T, a.gru_N = T, N
    a.gru_out, a.hxs_out = out, out[(T - 1) * N:]
    return out


def gru_backward(policy, a, dout, gflat, wgrad=True, dx_gate_src=None, dx_gate=None, need_dx=True):
    """BPTT through gru_forward.  dout [T*N, H] = d loss / d gru_out.  Overwrites the saved gates with d gi / d gh,
    writes the four GRU parameter gradients into `gflat` and returns d loss / d x_seq (gated by `dx_gate` on
    `dx_gate_src` when the GRU input was an activation output), or None."""
    w_ih, w_hh, b_ih, b_hh = gru_params(policy)
    T, N = a.gru_T, a.gru_N
    H = w_hh.shape[1]
    dev = dout.device
    L = _hip.lib()
    stream = _hip.cur_stream(dev)
    dout = dout.contiguous()
    carry = torch.empty(N, H, dtype=f32, device=dev)
    rzn, gh, hm, masks = a.gru_rzn, a.gru_gh, a.gru_hm, a.gru_masks
    fused = GRU_STEP_FUSED and L.acktr_gru_step_supported(N, H) == 1
    if fused:
        w_t = torch.empty(H, 3 * H, dtype=f32, device=dev)           # rows of W_hh^T: what a hidden unit's carry sums over
        _hip.check(L.acktr_transpose_f32(_hip.dev_ptr(w_hh), 3 * H, H, _hip.dev_ptr(w_t), stream), 'acktr_transpose_f32')
        p_do, p_rzn, p_gh, p_hm, p_m = dout.data_ptr(), rzn.data_ptr(), gh.data_ptr(), hm.data_ptr(), masks.data_ptr()
        p_wt, p_c = _hip.dev_ptr(w_t), _hip.dev_ptr(carry)
        vp = ctypes.c_void_p
    for t in range(T - 1, -1, -1):
        last = t == T - 1
        if fused:
            rc = L.acktr_gru_step_bwd_f32(vp(p_do + 4 * H * N * t), None if last else vp(p_m + 4 * N * (t + 1)),
                                          vp(p_rzn + 12 * H * N * t), vp(p_gh + 12 * H * N * t), vp(p_hm + 4 * H * N * t),
                                          None if last else vp(p_gh + 12 * H * N * (t + 1)), p_wt, p_c, 1 if last else 0, N, H,
                                          stream)
            if rc:
                _hip.check(rc, 'acktr_gru_step_bwd_f32')
            continue
        s = slice(t * N, (t + 1) * N)
        rc = L.acktr_gru_gates_bwd_f32(_hip.dev_ptr(dout[s]),
                                       None if last else _hip.dev_ptr(masks[(t + 1) * N:(t + 2) * N]),
                                       _hip.dev_ptr(rzn[s]), _hip.dev_ptr(gh[s]), _hip.dev_ptr(hm[s]),
                                       _hip.dev_ptr(carry), 1 if last else 0, N, H, stream)
        _hip.check(rc, 'acktr_gru_gates_bwd_f32')
        if t > 0:                                                      # d hm[t] += d gh[t] W_hh  (d hxs itself is unused)
            ops.linear_dgrad(gh[s], w_hh, out=carry, accumulate=True)
    if wgrad:
        gv = lambda p: policy.grad_view(p, gflat)
        ops.linear_wgrad(rzn, a.gru_x, gv(w_ih), gv(b_ih))
        ops.linear_wgrad(gh, hm, gv(w_hh), gv(b_hh))
    if need_dx:
        return ops.linear_dgrad(rzn, w_ih, gate_src=dx_gate_src, gate=dx_gate)
    return None


def _conv_image_layers(L):
    return [('f1', L['conv1'][0], 4, 84, False), ('f2', L['conv2'][0], 2, 20, False), ('f3', L['conv3'][0], 1, 9, False),
            ('d3', L['conv3'][0], 1, 9, True), ('d2', L['conv2'][0], 2, 20, True)]


def refresh_conv_images(policy, B):
    """Re-order the conv weights into the stationary kernels' images NOW (one launch) and return the _hip.PackMap with
    which an optimiser keeps them current (acktr_clip_adam_dev_pack_f32), or None when the policy has no such images.
    An update loop calls this once per update() and then runs forward(..., trust_images=True): between two updates
    anybody may have changed the parameters, inside one only the optimiser does."""
    if not is_cnn(policy) or not TRUST_IMAGES:
        return None
    L = cnn_layer_params(policy)
    layers = _conv_image_layers(L)
    cache = policy.__dict__.setdefault('_pack_ws', {})
    if not ops.pack_conv_weights(layers, B, cache):
        return None
    return ops.conv_image_map(layers, B, cache, lambda w: policy._offsets[id(w)][0])


def forward(policy, inputs, gather_idx=None, rnn_hxs=None, masks=None, head=True, trust_images=False, gate_bits=True):
    """head=False stops at the FC features (`head_step` then runs head, loss and their backward in one launch).
    trust_images: see refresh_conv_images.  gate_bits=False: a forward nobody differentiates (Policy.act / get_value)
    skips the ReLU gate bits that conv2's input gradient would read."""
    inputs = inputs.contiguous()
    _check_input(policy, inputs, gather_idx)
    rec = policy.is_recurrent
    if is_cnn(policy):
        L = cnn_layer_params(policy)
        a = Acts(True)
        a.obs, a.idx = inputs, gather_idx
        B = inputs.shape[0] if gather_idx is None else gather_idx.numel()
        # all five re-ordered weight images of the step (three forward, two dgrad) in one launch; the parameters do
        # not change between this forward and its backward
        pk = ops.pack_conv_weights(_conv_image_layers(L), B, policy.__dict__.setdefault('_pack_ws', {}),
                                   trust=trust_images and TRUST_IMAGES)
        a.packs = pk
        # conv2's input gradient needs only the sign of y1: the conv1 forward leaves it as bits (1.6 MB instead of a second
        # pass over 52 MB at 1024 frames) when both layers run their stationary-weight kernels
        a.g1 = ops.gate_mask_like(B, 32, 20, 20, inputs.device) \
            if (GATE_BITS and gate_bits and pk.get('f1') is not None and pk.get('d2') is not None) else None
        a.y1 = ops.conv2d_fwd(inputs, L['conv1'][0], L['conv1'][1].view(-1), 4, in_scale=1.0 / 255.0, relu=True,
                              gather_idx=gather_idx, packed=pk.get('f1'), gate_mask=a.g1)
        a.y2 = ops.conv2d_fwd(a.y1, L['conv2'][0], L['conv2'][1].view(-1), 2, packed=pk.get('f2'))
        a.y3 = ops.conv2d_fwd(a.y2, L['conv3'][0], L['conv3'][1].view(-1), 1, packed=pk.get('f3'))
        a.flat3 = a.y3.view(a.y3.shape[0], -1)
        if not head and not rec and FOLD_FC_SLABS:
            # the fused head step is the features' only consumer on this path: it takes the FC product's split-K slabs
            # and forms relu(sum + bias) itself (csrc/head_fused.hip) -- no slab-reduction launch, no feature matrix
            a.head_in = None
            a.feat_slabs = ops.linear_fwd_slabs(a.flat3, L['fc'][0]) + (L['fc'][1].view(-1),)
            a._feat_fn = lambda: ops.linear_fwd(a.flat3, L['fc'][0], L['fc'][1].view(-1), 'relu')
            return a
        a.feat = ops.linear_fwd(a.flat3, L['fc'][0], L['fc'][1].view(-1), 'relu')
        a.head_in = gru_forward(policy, a, a.feat, rnn_hxs, masks) if rec else a.feat
        if head:
            wh, bh = head_views(policy, policy.flat_params)
            a.head = ops.linear_fwd(a.head_in, wh, bh, None)
        return a
    L = mlp_layer_params(policy)
    a = Acts(False)
    if gather_idx is not None:
        from .storage import RolloutStorage
        inputs = RolloutStorage._gather(inputs, gather_idx, inputs.shape[0])
    a.obs = inputs
    if rec:
        inputs = gru_forward(policy, a, inputs, rnn_hxs, masks)
    a.tower_in = inputs
    a.hc1 = ops.linear_fwd(inputs, L['c0'][0], L['c0'][1].view(-1), 'tanh')
    a.hc2 = ops.linear_fwd(a.hc1, L['c2'][0], L['c2'][1].view(-1), 'tanh')
    a.ha1 = ops.linear_fwd(inputs, L['a0'][0], L['a0'][1].view(-1), 'tanh')
    a.ha2 = ops.linear_fwd(a.ha1, L['a2'][0], L['a2'][1].view(-1), 'tanh')
    a.v = ops.linear_fwd(a.hc2, L['critic'][0], L['critic'][1].view(-1), None)
    a.mean = ops.linear_fwd(a.ha2, L['mean'][0], L['mean'][1].view(-1), None)
    return a


def new_seed_buffers(policy, acts):
    """d(loss)/d(head outputs) buffers the loss kernels write and backward() consumes."""
    if acts.cnn:
        acts.dhead = torch.empty_like(acts.head)
    else:
        acts.dv = torch.empty_like(acts.v)
        acts.dmean = torch.empty_like(acts.mean)


def seed_ptrs(acts):
    """(dvalues ptr, ld, ddist ptr, ld)"""
    if acts.cnn:
        ld = acts.dhead.shape[1]
        return acts.dhead.data_ptr(), ld, acts.dhead.data_ptr() + 4, ld
    return acts.dv.data_ptr(), 1, acts.dmean.data_ptr(), acts.dmean.shape[1]


def late_bucket_offset(policy):
    """Flat-buffer offset that splits the NatureCNN gradient into [conv1, conv2 | conv3, fc, head]: the right part
    (97 % of the bytes) is complete after `backward(..., phase=1)`, so a data-parallel caller can start its
    all-reduce while phase 2 (conv3 dgrad, conv2, conv1) still runs.  None when the policy has no such split."""
    if not is_cnn(policy) or policy.is_recurrent:
        return None
    L = cnn_layer_params(policy)
    off = policy._offsets[id(L['conv3'][0])][0]
    for name in ('conv1', 'conv2'):
        for p in L[name]:
            if policy._offsets[id(p)][0] >= off:
                return None
    return off


def backward(policy, acts, gflat, keep=False, wgrad=True, phase=None, defer_last=None):
    """Parameter gradients of one batch into the flat gradient buffer `gflat` (overwrites, never
    accumulates).  phase (NatureCNN only): None = everything; 1 = head, FC and conv3 gradients (everything at or
    behind late_bucket_offset()); 2 = the rest, continuing from what phase 1 left on `acts`.  With keep=True the pre-activation gradients of every layer stay on `acts`
    (dy1, dy2, dy3, dfeat / dhc*, dha*) -- the `grad_output`s KFAC's backward hooks would see.
    wgrad=False runs the data-gradient chain only (the sampled-Fisher pass of ACKTR needs no
    parameter gradients, a2c_acktr.py:66-70).
    defer_last (a _hip.SlabFin, NatureCNN only): the LAST weight gradient (conv1) may leave its slab sum to the optimiser's
    norm launch, see ops.conv2d_wgrad.
    (Measured and dropped: issuing the weight-gradient launches on a second stream -- a parallel branch of the
    captured graph -- to fill the dispatch bubbles of the data-gradient chain.  11.63 ms per update became 12.51
    with every wgrad on the side stream and 11.70-11.75 with only the head / FC ones: the big kernels each fill
    the chip and evict each other's L2 working sets.)"""
    gv = lambda p: policy.grad_view(p, gflat)
    if acts.cnn:
        L = cnn_layer_params(policy)
        wh, _ = head_views(policy, policy.flat_params)
        w3, w2, w1 = L['conv3'][0], L['conv2'][0], L['conv1'][0]
        if phase != 2:
            fused = getattr(acts, 'dfeat_fused', None)          # head_step already produced dWh, dbh and dfeat
            if wgrad and fused is None:
                gwh, gbh = head_views(policy, gflat)
                ops.linear_wgrad(acts.dhead, acts.head_in, gwh, gbh)
            if fused is not None:
                dfeat = fused
            elif policy.is_recurrent:
                dgru = ops.linear_dgrad(acts.dhead, wh)
                dfeat = gru_backward(policy, acts, dgru, gflat, wgrad=wgrad, dx_gate_src=acts.feat, dx_gate='relu')
            else:
                dfeat = ops.linear_dgrad(acts.dhead, wh, gate_src=acts.feat, gate='relu')
            if wgrad:
                ops.linear_wgrad(dfeat, acts.flat3, gv(L['fc'][0]), gv(L['fc'][1]).view(-1))
            # the zero-padded copies of dy3 / dy2 that the stationary dgrads read are written by the PRODUCERS of dy3 /
            # dy2 (FC dgrad epilogue, conv3 dgrad epilogue) straight into those layers' workspaces: two padding
            # passes (8.5 + 12.8 us of a 655 us step at B = 1024) disappear
            B3 = acts.y3.shape[0]
            acts.pd3 = ops.padded_dy_view(acts.packs.get('d3'), B3, w3, 1, 9) if FOLD_PAD else None
            acts.pd2 = ops.padded_dy_view(acts.packs.get('d2'), B3, w2, 2, 20) if FOLD_PAD else None
            dy3 = ops.linear_dgrad(dfeat, L['fc'][0], gate_src=acts.flat3, gate='relu', padded_out=acts.pd3,
                                   planes=tuple(acts.y3.shape[1:])).view_as(acts.y3)
            # the slab sums of the conv3 / conv2 weight gradients ride in the NEXT launch (that layer's input gradient) as
            # extra workgroups -- not across a phase boundary, where a data-parallel caller all-reduces in between
            carry3 = _hip.SlabFin() if (wgrad and phase is None and CARRY_SLABS) else None
            if wgrad:
                ops.conv2d_wgrad(acts.y2, dy3, tuple(w3.shape), 1, gv(w3), gv(L['conv3'][1]).view(-1), pending=carry3)
            if phase == 1:
                acts.dfeat, acts.dy3 = dfeat, dy3
                return
        else:
            dfeat, dy3 = acts.dfeat, acts.dy3
            carry3 = None
        pd3 = acts.pd3
        # (conv3 dgrad off the stationary path: nobody fills pd2.  pd3 is None with a 'd3' image when that layer's dgrad
        #  reads dY unpadded: the overlap-add kernel, csrc/conv3_dgrad_oa.hpp)
        pd2 = acts.pd2 if acts.packs.get('d3') is not None else None
        dy2 = ops.conv2d_dgrad(dy3, w3, acts.y2, 1, packed=acts.packs.get('d3'), dy_prepadded=pd3 is not None,
                               padded_out=pd2, carry=carry3)
        carry2 = _hip.SlabFin() if (wgrad and CARRY_SLABS) else None
        if wgrad:
            ops.conv2d_wgrad(acts.y1, dy2, tuple(w2.shape), 2, gv(w2), gv(L['conv2'][1]).view(-1), pending=carry2)
        dy1 = ops.conv2d_dgrad(dy2, w2, acts.y1, 2, packed=acts.packs.get('d2'), dy_prepadded=pd2 is not None,
                               carry=carry2, gate_bits=getattr(acts, 'g1', None))
        if wgrad:
            ops.conv2d_wgrad(acts.obs, dy1, tuple(w1.shape), 4, gv(w1), gv(L['conv1'][1]).view(-1),
                             in_scale=1.0 / 255.0, gather_idx=acts.idx, pending=defer_last)
        if keep:
            acts.dfeat, acts.dy3, acts.dy2, acts.dy1 = dfeat, dy3, dy2, dy1
        return
    L = mlp_layer_params(policy)
    # critic tower
    dhc2 = ops.linear_dgrad(acts.dv, L['critic'][0], gate_src=acts.hc2, gate='tanh')
    dhc1 = ops.linear_dgrad(dhc2, L['c2'][0], gate_src=acts.hc1, gate='tanh')
    # actor tower
    dha2 = ops.linear_dgrad(acts.dmean, L['mean'][0], gate_src=acts.ha2, gate='tanh')
    dha1 = ops.linear_dgrad(dha2, L['a2'][0], gate_src=acts.ha1, gate='tanh')
    if wgrad:
        ops.linear_wgrad(acts.dv, acts.hc2, gv(L['critic'][0]), gv(L['critic'][1]).view(-1))
        ops.linear_wgrad(dhc2, acts.hc1, gv(L['c2'][0]), gv(L['c2'][1]).view(-1))
        ops.linear_wgrad(dhc1, acts.tower_in, gv(L['c0'][0]), gv(L['c0'][1]).view(-1))
        ops.linear_wgrad(acts.dmean, acts.ha2, gv(L['mean'][0]), gv(L['mean'][1]).view(-1))
        ops.linear_wgrad(dha2, acts.ha1, gv(L['a2'][0]), gv(L['a2'][1]).view(-1))
        ops.linear_wgrad(dha1, acts.tower_in, gv(L['a0'][0]), gv(L['a0'][1]).view(-1))
    if policy.is_recurrent:                     # both towers read the GRU output: d gru_out = dhc1 Wc0 + dha1 Wa0
        dgru = ops.linear_dgrad(dhc1, L['c0'][0])
        ops.linear_dgrad(dha1, L['a0'][0], out=dgru, accumulate=True)
        gru_backward(policy, acts, dgru, gflat, wgrad=wgrad, need_dx=False)
    if keep:
        acts.dhc2, acts.dhc1, acts.dha2, acts.dha1 = dhc2, dhc1, dha2, dha1


# ----------------------------------------------------------------------------------------------
# fused loss + backward seeds (csrc/loss.hip)
# ----------------------------------------------------------------------------------------------
ALGO_PPO, ALGO_A2C, ALGO_FISHER = 0, 1, 2


def policy_loss(policy, acts, algo, actions, loss_acc, inv_batch, old_logp=None, old_values=None, returns=None,
                adv=None, gather_idx=None, clip=0.0, value_coef=0.0, entropy_coef=0.0, use_clipped_value_loss=True,
                gflat=None, logp_out=None, dlogstd_rows=None):
    """Launch the fused loss kernel; fills the seed buffers on `acts` (and d logstd into `gflat`)."""
    dev = acts.head.device if acts.cnn else acts.v.device
    L = _hip.lib()
    B = acts.head.shape[0] if acts.cnn else acts.v.shape[0]
    A = policy.dist.num_outputs
    new_seed_buffers(policy, acts)
    args = _hip.LossArgs()
    args.algo, args.dist, args.B, args.A = algo, policy.dist.kind, B, A
    vp, vld = acts.value_ptr_ld()
    dp, dld = acts.dist_ptr_ld()
    args.values, args.ld_values, args.dist_in, args.ld_dist = vp, vld, dp, dld
    p = lambda t, dt=f32, name='loss operand': (None if t is None else _hip.dev_ptr(t, dt, name).value)
    if policy.dist.kind == 1:
        ls = logstd_param(policy)
        args.logstd = p(ls.detach())
        args.dlogstd = p(policy.grad_view(ls, policy.flat_grads if gflat is None else gflat))
        args.actions = p(actions, f32, 'actions')
    else:
        args.actions = p(actions, torch.int64, 'actions')
    args.old_logp, args.old_values, args.returns, args.adv = p(old_logp), p(old_values), p(returns), p(adv)
    args.gather_idx = p(gather_idx, torch.int64, 'gather_idx')
    args.clip, args.value_coef, args.entropy_coef, args.inv_batch = clip, value_coef, entropy_coef, inv_batch
    args.use_clipped_value_loss = 1 if use_clipped_value_loss else 0
    dvp, dvld, ddp, ddld = seed_ptrs(acts)
    args.dvalues, args.ld_dvalues, args.ddist, args.ld_ddist = dvp, dvld, ddp, ddld
    args.logp_out = p(logp_out)
    args.dlogstd_rows = p(dlogstd_rows)
    args.loss_acc = p(loss_acc)
    ws = ops.workspace(dev, L.acktr_policy_loss_workspace_bytes(B, A), 'loss')
    args.workspace, args.workspace_bytes = ws.data_ptr(), ws.numel()
    rc = L.acktr_policy_loss_f32(ctypes.byref(args), _hip.cur_stream(dev))
    _hip.check(rc, 'acktr_policy_loss_f32')


# ----------------------------------------------------------------------------------------------
# fused head + loss + head backward of the NatureCNN / Categorical path (csrc/head_fused.hip)
# ----------------------------------------------------------------------------------------------
FUSED_HEAD = os.environ.get('ACKTR_FUSED_HEAD', '1') != '0'
FOLD_FC_SLABS = os.environ.get('ACKTR_FOLD_FC_SLABS', '1') != '0'      # comparison switch
CARRY_SLABS = os.environ.get('ACKTR_CARRY_SLABS', '1') != '0'          # comparison switch
GATE_BITS = os.environ.get('ACKTR_GATE_BITS', '1') != '0'              # comparison switch (conv1's ReLU gate as bits)
TRUST_IMAGES = os.environ.get('ACKTR_ADAM_PACKS', '1') != '0'          # comparison switch (optimiser keeps the weight images)


def head_step_ok(policy, B):
    """Non-recurrent CNNBase(hidden 512) + Categorical with at most 7 actions; anything else runs layer by layer."""
    if not FUSED_HEAD or not is_cnn(policy) or policy.is_recurrent:
        return False
    hidden = cnn_layer_params(policy)['critic'][0].shape[1]
    return bool(_hip.lib().acktr_head_step_supported(int(B), int(hidden), int(policy.dist.num_outputs),
                                                     int(policy.dist.kind)))


def head_step(policy, acts, algo, actions, loss_acc, inv_batch, old_logp=None, old_values=None, returns=None, adv=None,
              gather_idx=None, clip=0.0, value_coef=0.0, entropy_coef=0.0, use_clipped_value_loss=True, gflat=None,
              logp_out=None, defer_finalize=False):
    """After forward(..., head=False): fills acts.head / acts.dhead / acts.dfeat_fused, writes the head's weight and
    bias gradients into `gflat` and adds this batch's (value_loss, action_loss, entropy) to loss_acc.
    defer_finalize: the slab sums behind the head gradients and the loss are left pending and RETURNED (_hip.HeadFin)
    for the optimiser's norm launch to carry (`_FlatAdam.step(pending=...)`): until then gflat's head ranges and
    loss_acc are stale -- only for a caller that runs that optimiser step next on the same stream."""
    slabs = getattr(acts, 'feat_slabs', None)        # (workspace, nslabs, slab stride in floats, FC bias): forward(head=False)
    feat = acts.head_in if slabs is None else slabs[0]
    dev = feat.device
    if slabs is None:
        B, H = feat.shape
    else:
        B, H = acts.flat3.shape[0], slabs[3].numel()
    A = policy.dist.num_outputs
    L = _hip.lib()
    gflat = policy.flat_grads if gflat is None else gflat
    wh, bh = head_views(policy, policy.flat_params)
    gwh, gbh = head_views(policy, gflat)
    acts.head = torch.empty(B, 1 + A, dtype=f32, device=dev)
    acts.dhead = torch.empty(B, 1 + A, dtype=f32, device=dev)
    dfeat = torch.empty(B, H, dtype=f32, device=dev)
    args = _hip.LossArgs()
    args.algo, args.dist, args.B, args.A = algo, 0, B, A
    p = lambda t, dt=f32, name='head step operand': (None if t is None else _hip.dev_ptr(t, dt, name).value)
    args.values, args.ld_values = p(acts.head), 1 + A
    args.dvalues, args.ld_dvalues = p(acts.dhead), 1 + A
    args.actions = p(actions, torch.int64, 'actions')
    args.old_logp, args.old_values, args.returns, args.adv = p(old_logp), p(old_values), p(returns), p(adv)
    args.gather_idx = p(gather_idx, torch.int64, 'gather_idx')
    args.clip, args.value_coef, args.entropy_coef, args.inv_batch = clip, value_coef, entropy_coef, inv_batch
    args.use_clipped_value_loss = 1 if use_clipped_value_loss else 0
    args.logp_out = p(logp_out)
    args.loss_acc = p(loss_acc)
    ws = ops.workspace(dev, L.acktr_head_step_workspace_bytes(B, H, A), 'head')
    args.workspace, args.workspace_bytes = ws.data_ptr(), ws.numel()
    pending = _hip.HeadFin() if defer_finalize else None
    with ops._Timed('head%dx%d_step' % (H, 1 + A), 6.0 * B * H * (1 + A)):
        if slabs is not None:
            rc = L.acktr_head_step_slabs_f32(ctypes.byref(args), _hip.dev_ptr(slabs[0], torch.uint8), int(slabs[1]), int(slabs[2]),
                                             p(slabs[3]), p(wh.detach()), p(bh.detach()), p(dfeat), p(gwh), p(gbh), H,
                                             ctypes.byref(pending) if pending is not None else None, _hip.cur_stream(dev))
        elif pending is not None:
            rc = L.acktr_head_step_deferred_f32(ctypes.byref(args), p(feat), p(wh.detach()), p(bh.detach()), p(dfeat),
                                                p(gwh), p(gbh), H, ctypes.byref(pending), _hip.cur_stream(dev))
        else:
            rc = L.acktr_head_step_f32(ctypes.byref(args), p(feat), p(wh.detach()), p(bh.detach()), p(dfeat), p(gwh),
                                       p(gbh), H, _hip.cur_stream(dev))
    _hip.check(rc, 'acktr_head_step_f32')
    acts.dfeat_fused = dfeat
    return pending


# ----------------------------------------------------------------------------------------------
# one-launch MLP minibatch step (csrc/mlp_fused.hip)
# ----------------------------------------------------------------------------------------------
FUSED_MLP = os.environ.get('ACKTR_FUSED_MLP', '1') != '0'
FOLD_PAD = os.environ.get('ACKTR_FOLD_PAD', '1') != '0'
GRU_STEP_FUSED = os.environ.get('ACKTR_GRU_STEP_FUSED', '1') != '0'      # tuning / comparison switch


def mlp_fused_ok(policy, B):
    """The fused single-workgroup step covers the reference's MuJoCo shape: non-recurrent MLPBase(hidden 64) with a
    DiagGaussian (or Categorical) head, up to 64 rows.  Anything else runs layer by layer."""
    if not FUSED_MLP or is_cnn(policy) or policy.is_recurrent or ops.profiler is not None or policy.dist.kind not in (0, 1):
        return False
    b = policy.base
    if hasattr(b.actor[0], 'add_bias'):            # KFAC's split_bias surgery: ACKTR needs the per-layer activations
        return False
    H, IN = b.actor[0].weight.shape
    return bool(_hip.lib().acktr_mlp_fused_step_supported(int(B), int(IN), int(H), int(policy.dist.num_outputs)))


def mlp_fused_step(policy, obs, gather_idx, B, algo, actions, loss_acc, inv_batch, old_logp=None, old_values=None,
                   returns=None, adv=None, clip=0.0, value_coef=0.0, entropy_coef=0.0, use_clipped_value_loss=True,
                   gflat=None, adam=None, max_grad_norm=None):
    """forward + loss + backward of one minibatch in ONE launch; gradients overwrite `gflat` (default flat_grads).
    `adam` (a ppo._FlatAdam whose device state is prepared): clip_grad_norm_(max_grad_norm) + the Adam step run in the
    same launch (the device step counter is incremented by the kernel; the caller keeps `step_count` in step)."""
    L = mlp_layer_params(policy)
    gflat = policy.flat_grads if gflat is None else gflat
    dev = obs.device
    if not obs.is_cuda or obs.dtype != f32 or not obs.is_contiguous():
        raise RuntimeError('policy inputs must be contiguous float32 tensors on the MI355X device')
    a = _hip.MlpStepArgs()
    p = lambda t, dt=f32, name='mlp step operand': (None if t is None else _hip.dev_ptr(t, dt, name).value)
    cat = policy.dist.kind == 0
    ls = None if cat else logstd_param(policy)
    for k, (w, b) in (('c0', L['c0']), ('c2', L['c2']), ('a0', L['a0']), ('a2', L['a2'])):
        setattr(a, 'w' + k, p(w.detach())); setattr(a, 'b' + k, p(b.detach()))
        setattr(a, 'gw' + k, p(policy.grad_view(w, gflat))); setattr(a, 'gb' + k, p(policy.grad_view(b, gflat)))
    a.wv, a.bv = p(L['critic'][0].detach()), p(L['critic'][1].detach())
    a.gwv, a.gbv = p(policy.grad_view(L['critic'][0], gflat)), p(policy.grad_view(L['critic'][1], gflat))
    a.wm, a.bm = p(L['mean'][0].detach()), p(L['mean'][1].detach())
    a.gwm, a.gbm = p(policy.grad_view(L['mean'][0], gflat)), p(policy.grad_view(L['mean'][1], gflat))
    if not cat:
        a.logstd, a.glogstd = p(ls.detach()), p(policy.grad_view(ls, gflat))
    a.obs = p(obs)
    a.gather_idx = p(gather_idx, torch.int64, 'gather_idx')
    H, IN = L['a0'][0].shape
    a.B, a.IN, a.H, a.A, a.algo = int(B), int(IN), int(H), int(policy.dist.num_outputs), int(algo)
    if cat:
        a.categorical = 1
        a.actions_i64 = p(actions, torch.int64, 'actions')
    else:
        a.actions = p(actions, f32, 'actions')
    a.old_logp, a.old_values, a.returns, a.adv = p(old_logp), p(old_values), p(returns), p(adv)
    a.clip, a.value_coef, a.entropy_coef, a.inv_batch = clip, value_coef, entropy_coef, inv_batch
    a.use_clipped_value_loss = 1 if use_clipped_value_loss else 0
    a.loss_acc = p(loss_acc)
    if adam is not None:
        if gflat is not policy.flat_grads:
            raise RuntimeError('the fused optimiser tail works on the policy\'s own flat gradient buffer')
        a.opt_params, a.opt_grads = p(policy.flat_params), p(gflat)
        a.opt_exp_avg, a.opt_exp_avg_sq = p(adam.exp_avg), p(adam.exp_avg_sq)
        a.opt_n = policy.flat_params.numel()
        a.opt_max_norm = float(max_grad_norm) if max_grad_norm else 0.0
        a.opt_hyper_dev = p(adam.hyper_dev)
        a.opt_step_dev = p(adam.step_dev, torch.int32)
    rc = _hip.lib().acktr_mlp_fused_step_f32(ctypes.byref(a), _hip.cur_stream(dev))
    _hip.check(rc, 'acktr_mlp_fused_step_f32')


# ----------------------------------------------------------------------------------------------
# autograd bridge for Policy.evaluate_actions (model.py:72-79)
# ----------------------------------------------------------------------------------------------
_KFAC_OF = None        # policy -> weakref of the KFACOptimizer built on it (its hooks' stand-in, algo/kfac.py)


def register_kfac(policy, opt):
    global _KFAC_OF
    import weakref
    if _KFAC_OF is None:
        _KFAC_OF = weakref.WeakKeyDictionary()
    _KFAC_OF[policy] = weakref.ref(opt)


def kfac_of(policy):
    r = _KFAC_OF.get(policy) if _KFAC_OF is not None else None
    return r() if r is not None else None


class EvalActions(torch.autograd.Function):
    """`kfac`: the KFACOptimizer whose module hooks (kfac.py:144-188) this node stands in for, or None.  `stats_a`: the
    forward-pre hooks' condition (grad enabled and steps % Ts == 0, kfac.py:145) as evaluated by the caller -- grad mode
    is off inside an autograd.Function's forward."""

    @staticmethod
    def forward(ctx, policy, inputs, action, rnn_hxs, masks, kfac, stats_a, *params):
        acts = forward(policy, inputs, rnn_hxs=rnn_hxs, masks=masks)
        if kfac is not None and stats_a:
            kfac.on_forward(acts)
        B = inputs.shape[0]
        A = policy.dist.num_outputs
        dev = inputs.device
        L = _hip.lib()
        logp = torch.empty(B, 1, dtype=f32, device=dev)
        ent = torch.empty(B, dtype=f32, device=dev)
        kind = policy.dist.kind
        act = action.contiguous()
        if kind == 0:
            act = act.long().view(-1)
        else:
            act = act.to(f32)
        ls = logstd_param(policy).detach() if kind == 1 else None
        dp, dld = acts.dist_ptr_ld()
        ent_mean = torch.empty((), dtype=f32, device=dev)
        rc = L.acktr_dist_fwd_f32(kind, dp, dld, _hip.dev_ptr(ls), _hip.dev_ptr(act), B, A, _hip.dev_ptr(logp),
                                  _hip.dev_ptr(ent), _hip.dev_ptr(ent_mean), _hip.cur_stream(dev))
        _hip.check(rc, 'acktr_dist_fwd_f32')
        ctx.policy, ctx.acts, ctx.act, ctx.kfac = policy, acts, act, kfac
        ctx.nparams = len(params)
        hxs_out = acts.hxs_out.clone() if policy.is_recurrent else torch.zeros(1, device=dev)
        ctx.mark_non_differentiable(hxs_out)
        return acts.value().clone(), logp, ent_mean, hxs_out

    @staticmethod
    def backward(ctx, dvalue, dlogp, dent, _dhxs):
        policy, acts, act = ctx.policy, ctx.acts, ctx.act
        B = act.shape[0]
        A = policy.dist.num_outputs
        kind = policy.dist.kind
        dev = act.device
        L = _hip.lib()
        new_seed_buffers(policy, acts)
        gflat = torch.zeros_like(policy.flat_params)
        dvp, dvld, ddp, ddld = seed_ptrs(acts)
        if acts.cnn:
            acts.dhead[:, 0:1].copy_(dvalue)
        else:
            acts.dv.copy_(dvalue)
        dl = dlogp.contiguous().view(-1)
        de = (dent / B).expand(B).contiguous()
        ls = logstd_param(policy) if kind == 1 else None
        ws = ops.workspace(dev, L.acktr_policy_loss_workspace_bytes(B, A), 'loss')
        dp, dld = acts.dist_ptr_ld()
        # KFAC's backward hooks fire while `acc_stats` is set, at steps % Ts == 0 (kfac.py:161-164)
        kfac = ctx.kfac
        stats_g = kfac is not None and kfac.acc_stats and kfac.steps % kfac.Ts == 0
        dls_rows = torch.empty(B, A, dtype=f32, device=dev) if (stats_g and kind == 1) else None
        rc = L.acktr_dist_bwd_f32(kind, dp, dld, _hip.dev_ptr(ls.detach() if ls is not None else None),
                                  _hip.dev_ptr(act), _hip.dev_ptr(dl), _hip.dev_ptr(de), B, A, ddp, ddld,
                                  _hip.dev_ptr(policy.grad_view(ls, gflat) if ls is not None else None),
                                  _hip.dev_ptr(dls_rows), _hip.dev_ptr(ws), ws.numel(), _hip.cur_stream(dev))
        _hip.check(rc, 'acktr_dist_bwd_f32')
        backward(policy, acts, gflat, keep=stats_g)
        if stats_g:
            kfac.on_backward(acts, dls_rows)
        grads = tuple(policy.grad_view(p, gflat) for p in policy.parameters())
        assert len(grads) == ctx.nparams
        return (None, None, None, None, None, None, None) + grads
