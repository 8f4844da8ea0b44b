"""GPU parity of the operator-level C-ABI entry points against plain torch fp32 CPU ops
(the ops the reference calls at model.py:176-195; floating point -> tolerance 2e-5 relative L2,
well inside the 1e-4 loss/gradient budget of BASELINE.json)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
TOL = 2e-5


def rel(a, b):
    a = a.detach().cpu().double().reshape(-1)
    b = b.detach().cpu().double().reshape(-1)
    return float((a - b).norm() / max(float(b.norm()), 1e-30))


GEOMS = [(84, 8, 4), (20, 4, 2), (9, 3, 1)]


@pytest.mark.parametrize('H,K,S', GEOMS)
@pytest.mark.parametrize('B,C,OC', [(1, 4, 32), (5, 32, 64), (33, 64, 32), (7, 3, 40)])
def test_conv2d_fwd_wgrad_dgrad(H, K, S, B, C, OC):
    from a2c_ppo_acktr import ops
    g = torch.Generator().manual_seed(H * 1000 + B)
    x = torch.rand(B, C, H, H, generator=g) * (255.0 if H == 84 else 1.0)
    x = torch.relu(x - (0.3 if H != 84 else 0.0))          # post-ReLU-like inputs with exact zeros
    w = torch.randn(OC, C, K, K, generator=g) / np.sqrt(C * K * K)
    b = torch.randn(OC, generator=g) * 0.1
    scale = 1.0 / 255.0 if H == 84 else 1.0
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    y_ref = F.relu(F.conv2d(xr * scale, wr, br, stride=S))
    dy = torch.randn(y_ref.shape, generator=g)
    dy_pre = dy * (y_ref > 0).float()
    y_ref.backward(dy)
    xd, wd, bd = x.cuda(), w.cuda(), b.cuda()
    y = ops.conv2d_fwd(xd, wd, bd, S, in_scale=scale, relu=True)
    assert rel(y, y_ref) < TOL
    dw = torch.empty_like(wd)
    db = torch.empty_like(bd)
    ops.conv2d_wgrad(xd, dy_pre.cuda(), tuple(w.shape), S, dw, db, in_scale=scale)
    assert rel(dw, wr.grad) < TOL
    assert rel(db, br.grad) < TOL
    # dgrad (without the producing layer's relu gate, then with it)
    dx = ops.conv2d_dgrad(dy_pre.cuda(), wd, None, S)
    assert rel(dx * scale, xr.grad) < TOL
    dxg = ops.conv2d_dgrad(dy_pre.cuda(), wd, xd, S)
    assert rel(dxg * scale, xr.grad * (x > 0).float()) < TOL


@pytest.mark.parametrize('B', [1, 2, 3, 7, 80, 513, 640, 1024, 1031])
def test_conv3_dgrad_overlap_add(B):
    """NatureCNN's conv3 input gradient (64 <- 32 channels, 3x3 s1, 9x9 <- 7x7) runs in overlap-add form
    (csrc/conv3_dgrad_oa.hpp): image ranges of 1-5 images per wave, tiles that straddle images, the ring of two LDS image
    slots, the gate, the zero-padded second copy for the conv2 dgrad, and the same result twice (fixed summation order)."""
    from a2c_ppo_acktr import ops
    g = torch.Generator().manual_seed(900 + B)
    dy = torch.randn(B, 32, 7, 7, generator=g)
    w = torch.randn(32, 64, 3, 3, generator=g) / 24.0
    xact = torch.relu(torch.randn(B, 64, 9, 9, generator=g))
    ref = F.conv_transpose2d(dy.double(), w.double()).float()
    dyd, wd, xd = dy.cuda(), w.cuda(), xact.cuda()
    dx = ops.conv2d_dgrad(dyd, wd, None, 1)
    assert rel(dx, ref) < TOL
    pad = torch.full((B, 64, 11, 11), 7.0, device='cuda')
    dxg = ops.conv2d_dgrad(dyd, wd, xd, 1, padded_out=pad)
    refg = ref * (xact > 0).float()
    assert rel(dxg, refg) < TOL
    assert torch.equal(pad[:, :, 1:10, 1:10], dxg), 'padded copy = the same values, interior only'
    assert bool((pad[:, :, 0, :] == 7.0).all()) and bool((pad[:, :, :, 10] == 7.0).all()), 'borders are not written'
    assert torch.equal(dxg == 0, (xd <= 0) | (dxg == 0))
    dxg2 = ops.conv2d_dgrad(dyd, wd, xd, 1)
    assert torch.equal(dxg, dxg2), 'run-to-run deterministic'


@pytest.mark.parametrize('B,u8,use_idx', [(1, False, False), (3, True, False), (37, False, True), (80, True, True),
                                          (1024, False, True), (1031, True, False)])
def test_conv1_gate_bits_conv2_dgrad(B, u8, use_idx):
    """The ReLU gate of the first layer as bits (acktr_conv2d_fwd_gate_f32 / acktr_conv1_fwd_u8_gate) read by the second
    layer's input gradient (flags bit 2): same y, mask == (y > 0) bit for bit, dx BIT-IDENTICAL to the float gate
    (the reference keeps the activation itself for ReLU's backward, model.py:177)."""
    from a2c_ppo_acktr import ops
    g = torch.Generator().manual_seed(4000 + B)
    src = B + 3 if use_idx else B
    x = torch.randint(0, 256, (src, 4, 84, 84), generator=g, dtype=torch.int32)
    x = (x.to(torch.uint8) if u8 else x.float()).cuda()
    idx = torch.randperm(src, generator=g)[:B].cuda() if use_idx else None
    w1 = (torch.randn(32, 4, 8, 8, generator=g) * 0.05).cuda()
    b1 = (torch.randn(32, generator=g) * 0.1).cuda()
    w2 = (torch.randn(64, 32, 4, 4, generator=g) / np.sqrt(512.0)).cuda()
    L1 = [('f1', w1, 4, 84, False), ('d2', w2, 2, 20, True)]
    pk = ops.pack_conv_weights(L1, B, {})
    assert pk and pk.get('f1') is not None and pk.get('d2') is not None
    y0 = ops.conv2d_fwd(x, w1, b1, 4, in_scale=1.0 / 255.0, relu=True, gather_idx=idx, packed=pk['f1'])
    gm = ops.gate_mask_like(B, 32, 20, 20, x.device)
    gm.fill_(-1)                                            # stale contents must not survive
    y1 = ops.conv2d_fwd(x, w1, b1, 4, in_scale=1.0 / 255.0, relu=True, gather_idx=idx, packed=pk['f1'], gate_mask=gm)
    assert torch.equal(y0, y1)
    # mask layout: word [r / 32][oc], bit r % 32, r = b*400 + pixel
    R = B * 400
    nt = (R + 31) // 32
    bits = (y1 > 0).permute(0, 2, 3, 1).reshape(R, 32)      # [r][oc]
    bits = torch.cat([bits, torch.zeros(nt * 32 - R, 32, dtype=torch.bool, device=bits.device)]).view(nt, 32, 32)
    weights = (1 << torch.arange(32, dtype=torch.int64, device=bits.device)).view(1, 32, 1)
    words = (bits.long() * weights).sum(1)                  # [tile][oc]
    words = torch.where(words >= 2 ** 31, words - 2 ** 32, words).to(torch.int32)
    assert torch.equal(gm.view(nt, 32), words)
    dy = torch.randn(B, 64, 9, 9, generator=g).cuda()
    dx_f = ops.conv2d_dgrad(dy, w2, y1, 2, packed=pk['d2'])
    dx_b = ops.conv2d_dgrad(dy, w2, None, 2, packed=pk['d2'], gate_bits=gm)
    assert torch.equal(dx_f.view(torch.int32), dx_b.view(torch.int32))
    assert float(dx_f.abs().sum()) > 0
    # shapes off those two paths reject the mask instead of ignoring it
    with pytest.raises(RuntimeError, match='gate'):
        ops.conv2d_fwd(y1, w2, None, 2, gate_mask=gm)
    w3 = torch.randn(64, 64, 3, 3, generator=g).cuda()
    with pytest.raises(RuntimeError, match='gate'):
        ops.conv2d_dgrad(torch.randn(B, 64, 7, 7, device='cuda'), w3, None, 1, gate_bits=gm)


@pytest.mark.parametrize('B', [300, 1011, 1024])
def test_conv2_fwd_k_split_tail(B):
    """conv2's forward (model.py:177) where workgroups end with 4m + 1 tail items and cut the last one into four K parts
    (SwGeom::KSPLIT, conv_sw.hpp): B = 1024 -> five tail items on 64 workgroups, B = 300 -> five on 16 and six on the rest,
    B = 1011 -> none.  Against torch's CPU conv2d, and bit-identical run to run (the parts are summed in a fixed order)."""
    from a2c_ppo_acktr import ops
    g = torch.Generator().manual_seed(7000 + B)
    x = torch.relu(torch.rand(B, 32, 20, 20, generator=g) - 0.3)
    w = torch.randn(64, 32, 4, 4, generator=g) / np.sqrt(512.0)
    b = torch.randn(64, generator=g) * 0.1
    ref = F.relu(F.conv2d(x, w, b, stride=2))
    xd, wd, bd = x.cuda(), w.cuda(), b.cuda()
    y0 = ops.conv2d_fwd(xd, wd, bd, 2)
    assert rel(y0, ref) < TOL
    y1 = ops.conv2d_fwd(xd, wd, bd, 2)
    assert torch.equal(y0.view(torch.int32), y1.view(torch.int32))


def test_conv1_gather_fused():
    from a2c_ppo_acktr import ops
    g = torch.Generator().manual_seed(11)
    store = torch.randint(0, 256, (40, 4, 84, 84), generator=g).float()
    idx = torch.randperm(40, generator=g)[:13]
    w = torch.randn(32, 4, 8, 8, generator=g) / 16
    b = torch.randn(32, generator=g) * 0.1
    y_ref = F.relu(F.conv2d(store[idx] / 255.0, w, b, stride=4))
    y = ops.conv2d_fwd(store.cuda(), w.cuda(), b.cuda(), 4, in_scale=1 / 255.0, relu=True, gather_idx=idx.cuda())
    assert rel(y, y_ref) < TOL
    dy = torch.randn(y_ref.shape, generator=g)
    wr = w.clone().requires_grad_(True)
    F.conv2d(store[idx] / 255.0, wr, b, stride=4).backward(dy)
    dw = torch.empty(32, 4, 8, 8, device='cuda')
    ops.conv2d_wgrad(store.cuda(), dy.cuda(), (32, 4, 8, 8), 4, dw, None, in_scale=1 / 255.0, gather_idx=idx.cuda())
    assert rel(dw, wr.grad) < TOL


def test_conv_wgrad_large_batch_split_k_deterministic():
    from a2c_ppo_acktr import ops
    g = torch.Generator().manual_seed(5)
    B = 300
    x = torch.rand(B, 32, 20, 20, generator=g)
    dy = torch.randn(B, 64, 9, 9, generator=g)
    wr = torch.zeros(64, 32, 4, 4, requires_grad=True)
    F.conv2d(x, wr, None, stride=2).backward(dy)
    dw1 = torch.empty(64, 32, 4, 4, device='cuda')
    dw2 = torch.empty_like(dw1)
    ops.conv2d_wgrad(x.cuda(), dy.cuda(), (64, 32, 4, 4), 2, dw1)
    ops.conv2d_wgrad(x.cuda(), dy.cuda(), (64, 32, 4, 4), 2, dw2)
    assert rel(dw1, wr.grad) < TOL
    assert torch.equal(dw1, dw2), 'split-K reduction must be run-to-run deterministic'


@pytest.mark.parametrize('B,IN,OUT', [(1, 11, 64), (64, 64, 64), (80, 1568, 512), (256, 512, 7), (1000, 1568, 512),
                                      (64, 64, 3), (130, 17, 64), (33, 512, 1),
                                      # the stationary-accumulator weight gradient (128..512 tiles of 64x64, B >= 256):
                                      (1024, 1568, 512), (301, 1499, 501), (257, 700, 770),
                                      # the in-workgroup split-K forward (128..512 tiles of 64x32, IN % 4 == 0, IN >= 512):
                                      (1000, 1564, 500), (513, 516, 300),
                                      # skinny heads at configs[4]-class batch sizes (512 row slices, 8 row walkers per quad):
                                      (20000, 64, 6), (16390, 64, 1)])
@pytest.mark.parametrize('act', [None, 'relu', 'tanh'])
def test_linear_fwd_bwd(B, IN, OUT, act):
    from a2c_ppo_acktr import ops
    g = torch.Generator().manual_seed(B + IN)
    x0 = torch.randn(B, IN, generator=g)
    xin = torch.relu(x0) if act == 'relu' else torch.tanh(x0)      # the activation that feeds the layer
    w = torch.randn(OUT, IN, generator=g) / np.sqrt(IN)
    b = torch.randn(OUT, generator=g) * 0.1
    xr, wr, br = xin.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    y_ref = F.linear(xr, wr, br)
    if act == 'relu':
        y_ref = F.relu(y_ref)
    elif act == 'tanh':
        y_ref = torch.tanh(y_ref)
    y = ops.linear_fwd(xin.cuda(), w.cuda(), b.cuda(), act)
    assert rel(y, y_ref) < TOL
    dy = torch.randn(B, OUT, generator=g)
    y_ref.backward(dy)
    if act == 'relu':
        dpre = dy * (y_ref > 0).float()
    elif act == 'tanh':
        dpre = dy * (1 - y_ref.detach() ** 2)
    else:
        dpre = dy
    dw = torch.empty(OUT, IN, device='cuda')
    db = torch.empty(OUT, device='cuda')
    ops.linear_wgrad(dpre.cuda(), xin.cuda(), dw, db)
    assert rel(dw, wr.grad) < TOL
    assert rel(db, br.grad) < TOL
    dx = ops.linear_dgrad(dpre.cuda(), w.cuda())
    assert rel(dx, xr.grad) < TOL
    gate = act if act else None
    if gate:
        dxg = ops.linear_dgrad(dpre.cuda(), w.cuda(), gate_src=xin.cuda(), gate=gate)
        ref = xr.grad * ((xin > 0).float() if gate == 'relu' else (1 - xin ** 2))
        assert rel(dxg, ref) < TOL


@pytest.mark.parametrize('ta,tb', [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize('M,N,K', [(512, 1568, 512), (6, 512, 6), (33, 65, 97), (1568, 1568, 512)])
def test_gemm_all_layouts(ta, tb, M, N, K):
    """A = I-style check with ASYMMETRIC operands: any row/col swap in the MFMA C layout is caught."""
    from a2c_ppo_acktr import ops
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn((K, M) if ta else (M, K), generator=g)
    b = torch.randn((N, K) if tb else (K, N), generator=g)
    ref = (a.t() if ta else a).double() @ (b.t() if tb else b).double()
    c = ops.gemm(a.cuda(), b.cuda(), trans_a=bool(ta), trans_b=bool(tb))
    assert rel(c, ref) < TOL
    c2 = ops.gemm(a.cuda(), b.cuda(), out=c.clone(), trans_a=bool(ta), trans_b=bool(tb), alpha=0.5, accumulate=True)
    assert rel(c2, 1.5 * ref) < TOL


@pytest.mark.parametrize('B', [257, 600, 1024])
def test_conv3_wgrad_stationary_accumulator(B):
    """conv_wgrad_sa.hpp (3x3 stride 1, 64 -> 32 channels): several images per workgroup, ragged last workgroup, the
    tail-safe last image; against torch fp32 on the CPU."""
    from a2c_ppo_acktr import ops
    g = torch.Generator().manual_seed(B)
    x = torch.relu(torch.rand(B, 64, 9, 9, generator=g) - 0.3)
    dy = torch.randn(B, 32, 7, 7, generator=g)
    w = torch.zeros(32, 64, 3, 3, requires_grad=True)
    bias = torch.zeros(32, requires_grad=True)
    torch.nn.functional.conv2d(x, w, bias).backward(dy)
    flat = torch.full((32 * 576 + 32,), 3.0, device='cuda')          # dw and db adjacent, like the flat gradient buffer
    dw, db = flat[:32 * 576].view(32, 64, 3, 3), flat[32 * 576:]
    ops.conv2d_wgrad(x.cuda(), dy.cuda(), (32, 64, 3, 3), 1, dw, db)
    assert rel(dw, w.grad) < 2e-5 and rel(db, bias.grad) < 2e-5
    dw2 = torch.empty(32, 64, 3, 3, device='cuda')                   # separate buffers, no bias
    ops.conv2d_wgrad(x.cuda(), dy.cuda(), (32, 64, 3, 3), 1, dw2, None)
    assert torch.equal(dw2, dw)


@pytest.mark.parametrize('B,gather', [(256, False), (270, True), (300, True), (1024, False)])
def test_conv1_wgrad_stationary_accumulator(B, gather):
    """conv_wgrad_sa.hpp, 8x8 stride-4 layer on 84x84x4 frames (B >= 256): ragged reduction ranges, gathered minibatch rows
    out of a larger store, the /255 input scale, bias sums; against torch fp32 on the CPU, and run-to-run deterministic."""
    from a2c_ppo_acktr import ops
    g = torch.Generator().manual_seed(B)
    nstore = B + 17 if gather else B
    store = torch.randint(0, 256, (nstore, 4, 84, 84), generator=g).float()
    idx = torch.randperm(nstore, generator=g)[:B] if gather else None
    x = store[idx] if gather else store
    dy = torch.randn(B, 32, 20, 20, generator=g) * (torch.rand(B, 32, 20, 20, generator=g) < 0.6).float()
    w = torch.zeros(32, 4, 8, 8, requires_grad=True)
    bias = torch.zeros(32, requires_grad=True)
    F.conv2d(x / 255.0, w, bias, stride=4).backward(dy)
    flat = torch.full((32 * 256 + 32,), 3.0, device='cuda')          # dw and db adjacent, like the flat gradient buffer
    dw, db = flat[:32 * 256].view(32, 4, 8, 8), flat[32 * 256:]
    gi = idx.cuda() if gather else None
    ops.conv2d_wgrad(store.cuda(), dy.cuda(), (32, 4, 8, 8), 4, dw, db, in_scale=1 / 255.0, gather_idx=gi)
    assert rel(dw, w.grad) < 2e-5 and rel(db, bias.grad) < 2e-5
    dw2 = torch.empty(32, 4, 8, 8, device='cuda')                    # separate buffers, no bias
    ops.conv2d_wgrad(store.cuda(), dy.cuda(), (32, 4, 8, 8), 4, dw2, None, in_scale=1 / 255.0, gather_idx=gi)
    assert torch.equal(dw2, dw)


@pytest.mark.parametrize('B', [256, 271, 300, 1024])
def test_conv2_wgrad_stationary_accumulator(B):
    """conv_wgrad_sa.hpp, 4x4 stride-2 layer on 20x20x32 activations (B >= 256): ragged image ranges, the single-position
    eleventh block of every image, bias sums, the last image's tail; against torch fp32 on the CPU."""
    from a2c_ppo_acktr import ops
    g = torch.Generator().manual_seed(B + 1)
    x = torch.relu(torch.rand(B, 32, 20, 20, generator=g) - 0.3)
    dy = torch.randn(B, 64, 9, 9, generator=g) * (torch.rand(B, 64, 9, 9, generator=g) < 0.6).float()
    w = torch.zeros(64, 32, 4, 4, requires_grad=True)
    bias = torch.zeros(64, requires_grad=True)
    F.conv2d(x, w, bias, stride=2).backward(dy)
    flat = torch.full((64 * 512 + 64,), 3.0, device='cuda')          # dw and db adjacent, like the flat gradient buffer
    dw, db = flat[:64 * 512].view(64, 32, 4, 4), flat[64 * 512:]
    ops.conv2d_wgrad(x.cuda(), dy.cuda(), (64, 32, 4, 4), 2, dw, db)
    assert rel(dw, w.grad) < 2e-5 and rel(db, bias.grad) < 2e-5
    dw2 = torch.empty(64, 32, 4, 4, device='cuda')                   # separate buffers, no bias
    ops.conv2d_wgrad(x.cuda(), dy.cuda(), (64, 32, 4, 4), 2, dw2, None)
    assert torch.equal(dw2, dw)
