"""KFACOptimizer -- host-side mirror of reference a2c_ppo_acktr/algo/kfac.py:74-242.

Same constructor, `split_bias` model surgery (parameter names become `*.module.weight` /
`*.add_bias._bias`), public attributes (`steps, Ts, Tf, lr, acc_stats, m_aa, m_gg, Q_a, Q_g, d_a, d_g,
modules`) and `step()` contract.  The reference gathers its statistics through forward-pre / backward
hooks on torch modules; here the HIP engine already holds every layer's input and pre-activation
gradient, so `A2C_ACKTR.update` calls `accumulate_stats()` where the hooks would have fired.

Device data layout: all A factors live in one flat fp32 bucket, all G factors in another (sample and
running-average twins) -- one RCCL all-reduce each under data parallelism; eigenvectors are stored as
ROWS (`qt = Q^T`), eigenvalues ascending.
"""
import ctypes
import os

import torch
import torch.nn as nn
import torch.optim as optim

from .. import _hip, engine, ops
from ..parallel import DataParallel, assign_eig_owners
from ..utils import AddBias

f32 = torch.float32


class SplitBias(nn.Module):
    """reference kfac.py:74-84: bias-less module followed by an AddBias holding the former bias."""

    def __init__(self, module):
        super(SplitBias, self).__init__()
        self.module = module
        self.add_bias = AddBias(module.bias.data)
        self.module.bias = None


_JacobiDesc = _hip.JacobiJob          # {a, d, qt, n}: device descriptor of the LDS kernel, host job of the blocked one


class _Mod(object):
    """Per known-module bookkeeping: which engine layer it belongs to and its factor slots."""
    __slots__ = ('module', 'param', 'layer', 'role', 'kind', 'na', 'ng', 'oa', 'og', 'da_off', 'dg_off')


class KFACOptimizer(optim.Optimizer):
    def __init__(self,
                 model,
                 lr=0.25,
                 momentum=0.9,
                 stat_decay=0.99,
                 kl_clip=0.001,
                 damping=1e-2,
                 weight_decay=0,
                 fast_cnn=False,
                 Ts=1,
                 Tf=10):
        defaults = dict()

        def split_bias(module):
            for mname, child in module.named_children():
                if hasattr(child, 'bias') and child.bias is not None:
                    module._modules[mname] = SplitBias(child)
                else:
                    split_bias(child)

        if getattr(model, 'is_recurrent', False):
            raise NotImplementedError('Recurrent policy is not implemented for ACKTR')       # arguments.py:157-159
        split_bias(model)
        model.rebind_flat()

        super(KFACOptimizer, self).__init__(model.parameters(), defaults)


        self.known_modules = {'Linear', 'Conv2d', 'AddBias'}

        self.modules = []
        self.grad_outputs = {}

        self.model = model
        self._prepare_model()

        self.steps = 0

        self.m_aa, self.m_gg = {}, {}
        self.Q_a, self.Q_g = {}, {}
        self.d_a, self.d_g = {}, {}

        self.momentum = momentum
        self.stat_decay = stat_decay

        self.lr = lr
        self.kl_clip = kl_clip
        self.damping = damping
        self.weight_decay = weight_decay

        self.fast_cnn = fast_cnn

        self.Ts = Ts
        self.Tf = Tf
        self.acc_stats = False

        # inner SGD(lr*(1-momentum), momentum) of kfac.py:139-142, on the flat buffers
        self.sgd_lr = self.lr * (1 - self.momentum)
        self._mom_buf = None
        self._ready_device = None
        self.dp = DataParallel()          # replaced by A2C_ACKTR with its own context
        self._ext_noise = None
        self._side_stream = None
        self._bcast_buf = None
        self._n_stats = {'a': 0, 'g': 0}       # running-average updates so far (step() refuses to run on none)
        engine.register_kfac(model, self)

    # ------------------------------------------------------------------ module discovery (kfac.py:179-188)
    def _prepare_model(self):
        pol = self.model
        layers = engine.cnn_layer_params(pol) if engine.is_cnn(pol) else engine.mlp_layer_params(pol)
        owner = {}
        for lname, (w, b) in layers.items():
            owner[id(w)] = (lname, 'w')
            owner[id(b)] = (lname, 'b')
        if pol.dist.kind == 1:
            owner[id(engine.logstd_param(pol))] = ('logstd', 'b')
        self._mods = []
        for module in pol.modules():
            classname = module.__class__.__name__
            if classname in self.known_modules:
                assert not ((classname in ['Linear', 'Conv2d']) and module.bias is not None), \
                    "You must have a bias as a separate layer"
                self.modules.append(module)
                ps = list(module.parameters())
                assert len(ps) == 1, "Can handle only one parameter at the moment"
                m = _Mod()
                m.module, m.param = module, ps[0]
                m.layer, m.role = owner[id(ps[0])]
                m.kind = classname
                p = ps[0]
                m.ng = p.shape[0]
                m.na = 1 if classname == 'AddBias' else p[0].numel()
                self._mods.append(m)

    def _setup(self, dev):
        """Allocate factor buckets, eigen storage and scratch on `dev` (first use)."""
        if self._ready_device == dev:
            return
        oa = og = 0
        da = dg = 0
        for m in self._mods:
            m.oa, m.og = oa, og
            m.da_off, m.dg_off = da, dg
            al4 = lambda n: (n + 3) // 4 * 4                      # keep every slot 16-byte aligned (float4 loads)
            oa += al4(m.na * m.na)
            og += al4(m.ng * m.ng)
            da += al4(m.na)
            dg += al4(m.ng)
        z = lambda n: torch.zeros(n, dtype=f32, device=dev)
        self._bias_a_done = set()       # (the AddBias slots of the NEW sample bucket have not been filled)
        self.aa_sample, self.gg_sample = z(oa), z(og)
        self.aa_run, self.gg_run = z(oa), z(og)
        self.qt_a, self.qt_g = z(oa), z(og)
        self.q_a, self.q_g = z(oa), z(og)               # the eigenvector matrices themselves (transposes of the qt slots)
        self.d_a_flat, self.d_g_flat = z(da), z(dg)
        for m in self._mods:
            mod = m.module
            self.m_aa[mod] = self.aa_run[m.oa:m.oa + m.na * m.na].view(m.na, m.na)
            self.m_gg[mod] = self.gg_run[m.og:m.og + m.ng * m.ng].view(m.ng, m.ng)
            self.Q_a[mod] = self.qt_a[m.oa:m.oa + m.na * m.na].view(m.na, m.na).t()
            self.Q_g[mod] = self.qt_g[m.og:m.og + m.ng * m.ng].view(m.ng, m.ng).t()
            self.d_a[mod] = self.d_a_flat[m.da_off:m.da_off + m.na]
            self.d_g[mod] = self.d_g_flat[m.dg_off:m.dg_off + m.ng]
        self.v_flat = torch.zeros_like(self.model.flat_params)
        mx = max(m.na * m.ng for m in self._mods)
        self.tmp1, self.tmp2 = z(mx), z(mx)
        self.nu_buf = z(2)
        nmax = max(max(m.na, m.ng) for m in self._mods)
        self.e_work = z(nmax)      # re-sized below to the sum over this rank's rocSOLVER factors
        self.info = torch.zeros(1, dtype=torch.int32, device=dev)
        # eigendecomposition work list: (run bucket, qt bucket, d bucket, offset, d offset, n), one per factor,
        # sharded layer-wise over the data-parallel ranks (greedy by n^3) -- SURVEY 8(e)
        L = _hip.lib()
        jmax = L.acktr_jacobi_max_n()                     # LDS-resident batched Jacobi (one workgroup per factor)
        # blocked one-sided Jacobi above that (csrc/jacobi_block.hip) up to ACKTR_EIG_BLOCK_MAX_N (default 1024: north_star's
        # "hand-written batched Jacobi for factors of 1024 or less"), rocSOLVER syevd beyond -- NatureCNN's 1568 factor, whose
        # solve then runs CONCURRENTLY with the Jacobi launches.  ACKTR_EIG_BLOCK_MAX_N=2048 sends the 1568 factor through
        # the Jacobi too: the update lands 5x closer to fp64 (3.8e-4 against 2.0e-3 for the reference's own fp32 run) at
        # 66 ms per eigen step instead of ~33; =0 restores the round-1 path (rocSOLVER for everything above 128).
        bmax = min(int(os.environ.get('ACKTR_EIG_BLOCK_MAX_N', '1024')), L.acktr_jacobi_block_max_n())
        work = []
        for m in self._mods:
            work.append((self.aa_run, self.qt_a, self.d_a_flat, m.oa, m.da_off, m.na))
            work.append((self.gg_run, self.qt_g, self.d_g_flat, m.og, m.dg_off, m.ng))
        self._eig_work = work
        self._eig_owner = assign_eig_owners([w[5] for w in work], self.dp.world)
        descs, self._big, self._blk = [], [], []
        self._jac_max = 1
        for w, own in zip(work, self._eig_owner):
            if own != self.dp.rank:
                continue
            run, qt, dfl, off, doff, n = w
            if n <= jmax:       # batched Jacobi descriptors (built once: pointers are stable)
                descs.append(_JacobiDesc(run.data_ptr() + 4 * off, dfl.data_ptr() + 4 * doff, qt.data_ptr() + 4 * off, n))
                self._jac_max = max(self._jac_max, n)
            elif n <= bmax:
                self._blk.append(w)
            else:
                self._big.append(w)
        if self._blk:              # host job table + workspace of the blocked Jacobi (pointers are stable)
            self._blk_jobs = (_hip.JacobiJob * len(self._blk))(*[
                _hip.JacobiJob(run.data_ptr() + 4 * off, dfl.data_ptr() + 4 * doff, qt.data_ptr() + 4 * off, n)
                for run, qt, dfl, off, doff, n in self._blk])
            ns = (ctypes.c_int * len(self._blk))(*[w[5] for w in self._blk])
            nbytes = L.acktr_sym_eig_jacobi_block_workspace_bytes(ns, len(self._blk))
            if nbytes < 0:
                raise RuntimeError('acktr_sym_eig_jacobi_block_workspace_bytes rejected the factor sizes')
            self._blk_ws = torch.empty(int(nbytes), dtype=torch.uint8, device=dev)
            self._blk_sweeps = (ctypes.c_int * len(self._blk))()
        if self._big:              # host-side job table for the concurrent rocSOLVER solves (pointers are stable)
            self.e_work = z(sum(w[5] for w in self._big))
            self.info = torch.zeros(len(self._big), dtype=torch.int32, device=dev)
            self._big_jobs = (_hip.EigJob * len(self._big))(*[
                _hip.EigJob(qt.data_ptr() + 4 * off, dfl.data_ptr() + 4 * doff, n, 0)
                for run, qt, dfl, off, doff, n in self._big])
        self._jac_count = len(descs)
        if descs:
            arr = (_JacobiDesc * len(descs))(*descs)
            self._jac_descs = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
        self._ready_device = dev

    # ------------------------------------------------------------------ statistics (kfac.py:144-177)
    def _cov_dense(self, a2d, ptr, ld, B, D, alpha, out, share=False):
        # Two modules that see the same tensor get the same factor: critic_linear and dist.linear both read the FC features
        # (A), a Linear and the AddBias behind it receive the same grad_output (G; kfac.py:59-64 is a no-op on 2-D
        # gradients).  The reference computes each twice; here the second one is a copy of the first (same bits).
        # (share: only for operands that live on `acts` for the whole statistics pass -- a temporary's address can be
        #  reused by the next temporary)
        seen = self.__dict__.get('_cov_seen') if share else None
        key = (ptr, ld, B, D, float(alpha))
        if seen is not None:
            prev = seen.get(key)
            if prev is not None and D > 1:
                out.copy_(prev)
                return
            seen[key] = out
        L = _hip.lib()
        ws = ops.workspace(out.device, L.acktr_kfac_cov_workspace_bytes(D, B), 'kfac')
        with ops._Timed('kfac_cov_dense%d' % D, float(B) * D * (D + 1)):
            rc = L.acktr_kfac_cov_dense_f32(ptr, ld, B, D, float(alpha), _hip.dev_ptr(out), _hip.dev_ptr(ws), ws.numel(),
                                            _hip.cur_stream(out.device))
        _hip.check(rc, 'acktr_kfac_cov_dense_f32')

    def accumulate_stats(self, acts, actions, gather_idx, inv_batch, dp):
        """What the reference's hooks do during the forward (compute_cov_a on every layer input) and during
        the sampled-Fisher backward (compute_cov_g on every grad_output), a2c_acktr.py:53-68."""
        pol = self.model
        cnn = acts.cnn
        dev = acts.head.device if cnn else acts.v.device
        self._setup(dev)
        B = acts.head.shape[0] if cnn else acts.v.shape[0]       # local rows
        Bg = int(round(1.0 / inv_batch))                         # global batch (kfac.py:30,50 use the full batch)
        self._stats_a(acts, B, Bg)

        # ---- sampled Fisher loss and its data-gradient chain (a2c_acktr.py:56-68)
        rows_total = Bg
        if self._ext_noise is not None:                           # hipGraph capture: the caller refills this buffer
            noise = self._ext_noise.view(-1)                      # from the same CPU draw before every replay
        else:
            noise = torch.randn(rows_total, 1)                    # CPU default generator, like the reference (F7)
            noise = noise.to(dev, non_blocking=True).view(-1)
        dls_rows = None
        if pol.dist.kind == 1:
            dls_rows = torch.empty(B, pol.dist.num_outputs, dtype=f32, device=dev)
        scratch_g = self.v_flat                                   # d logstd sum lands here; overwritten in step()
        engine.policy_loss(pol, acts, engine.ALGO_FISHER, actions, None, inv_batch, adv=noise, gather_idx=gather_idx,
                           gflat=scratch_g, dlogstd_rows=dls_rows)
        engine.backward(pol, acts, scratch_g, keep=True, wgrad=False)
        self._stats_g(acts, B, Bg, dls_rows)

        # ---- data-parallel: SUM the sample factors (global scalings already applied), then running averages
        dp.all_reduce_sum_many_((self.aa_sample, self.gg_sample))
        self._running_stat('a')
        self._running_stat('g')

    # -- the reference's own call pattern (a2c_acktr.py:53-78 driven by user code): Policy.evaluate_actions and the
    # -- backward of its autograd node call these where the module hooks of kfac.py:144-177 fire
    def on_forward(self, acts):
        """kfac.py:144-159 (_save_input, a forward-pre hook on every known module): with grad enabled and
        steps % Ts == 0, EVERY forward updates the running input statistics m_aa."""
        cnn = acts.cnn
        dev = acts.head.device if cnn else acts.v.device
        self._setup(dev)
        B = acts.head.shape[0] if cnn else acts.v.shape[0]
        with torch.no_grad():
            self._stats_a(acts, B, B * self.dp.world)
            self.dp.all_reduce_sum_(self.aa_sample)
            self._running_stat('a')

    def on_backward(self, acts, dls_rows):
        """kfac.py:161-177 (_save_grad_output, a backward hook): while `acc_stats` is set (and steps % Ts == 0) a
        backward pass updates the running grad-output statistics m_gg.  `acts` carries the pre-activation gradients of
        that pass (engine.backward(keep=True)), dls_rows the logstd module's grad_output (Gaussian heads)."""
        cnn = acts.cnn
        B = acts.head.shape[0] if cnn else acts.v.shape[0]
        with torch.no_grad():
            self._stats_g(acts, B, B * self.dp.world, dls_rows)
            self.dp.all_reduce_sum_(self.gg_sample)
            self._running_stat('g')

    def _running_stat(self, which):
        """update_running_stat (kfac.py:67-71) over one flat bucket; the first update of a bucket initialises the
        running average to the sample (kfac.py:156-157,174-175: `steps == 0` clones)."""
        run, smp = (self.aa_run, self.aa_sample) if which == 'a' else (self.gg_run, self.gg_sample)
        init = 1 if self.steps == 0 else 0
        d = float(self.stat_decay)
        rc = _hip.lib().acktr_kfac_running_stat_f32(_hip.dev_ptr(run), _hip.dev_ptr(smp), run.numel(), d / (1 - d), 1 - d,
                                                    init, _hip.cur_stream(run.device))
        _hip.check(rc, 'acktr_kfac_running_stat_f32')
        self._n_stats[which] += 1

    def _stats_a(self, acts, B, Bg):
        """compute_cov_a of every known module into the sample bucket (kfac.py:29-46)."""
        pol = self.model
        cnn = acts.cnn
        dev = self.aa_sample.device
        L = _hip.lib()
        stream = _hip.cur_stream(dev)
        view_a = lambda m: self.aa_sample[m.oa:m.oa + m.na * m.na]
        self._cov_seen = {}

        # ---- A factors: compute_cov_a on the layer inputs (kfac.py:29-46)
        if cnn:
            conv_in = {'conv1': (acts.obs, acts.idx, 1.0 / 255.0, 4), 'conv2': (acts.y1, None, 1.0, 2),
                       'conv3': (acts.y2, None, 1.0, 1)}
            lin_in = {'fc': acts.flat3, 'critic': acts.feat, 'actor': acts.feat}
        else:
            conv_in = {}
            lin_in = {'a0': acts.obs, 'a2': acts.ha1, 'c0': acts.obs, 'c2': acts.hc1, 'critic': acts.hc2,
                      'mean': acts.ha2}
        for m in self._mods:
            out = view_a(m)
            if m.kind == 'AddBias':
                # ones^T (ones / B_global) over the local rows: a constant.  Single process: nothing else writes the slot,
                # so it is filled once per (slot, B, Bg); under data parallelism the in-place all-reduce sums into it every
                # step, so it is refilled every step
                done = self.__dict__.setdefault('_bias_a_done', set())
                key_ = (m.oa, B, Bg, out.data_ptr())
                if self.dp.active or key_ not in done:
                    out.fill_(float(B) / Bg)
                    done.add(key_)
            elif m.kind == 'Conv2d':
                x, idx, sc, stride = conv_in[m.layer]
                OC, C, KH, KW = m.param.shape
                H, W = x.shape[-2], x.shape[-1]
                OH, OW = (H - KH) // stride + 1, (W - KW) // stride + 1
                if x.dtype == torch.uint8:                          # frames kept as bytes: conv1 only
                    if self.fast_cnn:
                        raise NotImplementedError('KFACOptimizer(fast_cnn=True) reads float observations; uint8 frames '
                                                  '(RolloutStorage(obs_dtype=torch.uint8)) are not supported there')
                    alpha = (sc * sc) / (float(Bg) * float(OH * OW) ** 2)
                    D_, R_ = C * KH * KW, B * OH * OW
                    ws = ops.workspace(dev, L.acktr_kfac_cov_workspace_bytes(D_, R_), 'kfac')
                    with ops._Timed('kfac_cov_a_conv%ds%d' % (KH, stride), float(R_) * D_ * (D_ + 1)):
                        rc = L.acktr_kfac_cov_a_conv1_u8(_hip.dev_ptr(x, torch.uint8), _hip.dev_ptr(idx, torch.int64), B, alpha,
                                                         _hip.dev_ptr(out), _hip.dev_ptr(ws), ws.numel(), stream)
                    _hip.check(rc, 'acktr_kfac_cov_a_conv1_u8')
                    continue
                if self.fast_cnn:                                   # kfac.py:33-36: one mean patch per image
                    pm = torch.empty(B, C * KH * KW, dtype=f32, device=dev)
                    rc = L.acktr_kfac_patch_mean_f32(_hip.dev_ptr(x), _hip.dev_ptr(idx, torch.int64), B, C, H, W, KH, KW,
                                                     stride, sc, _hip.dev_ptr(pm), stream)
                    _hip.check(rc, 'acktr_kfac_patch_mean_f32')
                    self._cov_dense(pm, pm.data_ptr(), C * KH * KW, B, C * KH * KW, 1.0 / Bg, out)
                    continue
                alpha = (sc * sc) / (float(Bg) * float(OH * OW) ** 2)
                ws = ops.workspace(dev, L.acktr_kfac_cov_workspace_bytes(C * KH * KW, B * OH * OW), 'kfac')
                D_, R_ = C * KH * KW, B * OH * OW           # symmetric-half flop count R D (D+1), SURVEY 8(d)
                with ops._Timed('kfac_cov_a_conv%ds%d' % (KH, stride), float(R_) * D_ * (D_ + 1)):
                    rc = L.acktr_kfac_cov_a_conv_f32(_hip.dev_ptr(x), _hip.dev_ptr(idx, torch.int64), B, C, H, W, KH, KW,
                                                     stride, alpha, _hip.dev_ptr(out), _hip.dev_ptr(ws), ws.numel(), stream)
                _hip.check(rc, 'acktr_kfac_cov_a_conv_f32')
            else:
                a = lin_in[m.layer]
                self._cov_dense(a, a.data_ptr(), a.shape[1], B, a.shape[1], 1.0 / Bg, out, share=True)


    def _stats_g(self, acts, B, Bg, dls_rows):
        """compute_cov_g of every known module into the sample bucket (kfac.py:49-64)."""
        pol = self.model
        cnn = acts.cnn
        dev = self.gg_sample.device
        L = _hip.lib()
        stream = _hip.cur_stream(dev)
        gauss = pol.dist.kind == 1
        view_g = lambda m: self.gg_sample[m.og:m.og + m.ng * m.ng]
        self._cov_seen = {}
        # ---- G factors: compute_cov_g on the grad_outputs (kfac.py:49-64)
        if cnn:
            conv_g = {'conv1': acts.dy1, 'conv2': acts.dy2, 'conv3': acts.dy3}
            A = pol.dist.num_outputs
            hp, hl = acts.dhead.data_ptr(), acts.dhead.shape[1]
            lin_g = {'fc': (acts.dfeat.data_ptr(), acts.dfeat.shape[1], acts.dfeat.shape[1]),
                     'critic': (hp, hl, 1), 'actor': (hp + 4, hl, A)}
            if gauss:
                lin_g['logstd'] = (dls_rows.data_ptr(), A, A)
        else:
            conv_g = {}
            A = pol.dist.num_outputs
            lin_g = {'a0': (acts.dha1.data_ptr(), acts.dha1.shape[1], acts.dha1.shape[1]),
                     'a2': (acts.dha2.data_ptr(), acts.dha2.shape[1], acts.dha2.shape[1]),
                     'c0': (acts.dhc1.data_ptr(), acts.dhc1.shape[1], acts.dhc1.shape[1]),
                     'c2': (acts.dhc2.data_ptr(), acts.dhc2.shape[1], acts.dhc2.shape[1]),
                     'critic': (acts.dv.data_ptr(), 1, 1), 'mean': (acts.dmean.data_ptr(), A, A)}
            if gauss:
                lin_g['logstd'] = (dls_rows.data_ptr(), A, A)
        for m in self._mods:
            out = view_g(m)
            if m.layer in conv_g:
                dy = conv_g[m.layer]
                OC, SP = dy.shape[1], dy.shape[2] * dy.shape[3]
                if m.kind == 'Conv2d' and not self.fast_cnn:
                    ws = ops.workspace(dev, L.acktr_kfac_cov_workspace_bytes(OC, B * SP), 'kfac')
                    rc = L.acktr_kfac_cov_g_conv_f32(_hip.dev_ptr(dy), B, OC, SP, float(SP) * Bg, _hip.dev_ptr(out),
                                                     _hip.dev_ptr(ws), ws.numel(), stream)
                    _hip.check(rc, 'acktr_kfac_cov_g_conv_f32')
                else:                              # AddBias after a conv (and Conv2d with fast_cnn): sum over space first
                    ssum = torch.empty(B, OC, dtype=f32, device=dev)
                    rc = L.acktr_spatial_sum_f32(_hip.dev_ptr(dy), B * OC, SP, _hip.dev_ptr(ssum), stream)
                    _hip.check(rc, 'acktr_spatial_sum_f32')
                    self._cov_dense(ssum, ssum.data_ptr(), OC, B, OC, float(Bg), out)
            else:
                ptr, ld, D = lin_g[m.layer]
                self._cov_dense(None, ptr, ld, B, D, float(Bg), out, share=True)

    # ------------------------------------------------------------------ eigendecompositions (kfac.py:205-214)
    def _eigendecompose(self):
        L = _hip.lib()
        dev = self.aa_run.device
        stream = _hip.cur_stream(dev)
        side = None
        if self._jac_count:
            # the LDS-resident solver (one workgroup per small factor, ~2 ms of a handful of CUs) on a side stream, so that
            # neither the rocSOLVER fork point nor the blocked Jacobi's launches queue behind it
            cur = torch.cuda.current_stream(dev)
            if self._side_stream is None:
                self._side_stream = torch.cuda.Stream(device=dev)
            side = self._side_stream
            side.wait_stream(cur)
            rc = L.acktr_sym_eig_jacobi_batched_f32(_hip.dev_ptr(self._jac_descs, torch.uint8), self._jac_count,
                                                    self._jac_max, 1e-6, ctypes.c_void_p(side.cuda_stream))
            _hip.check(rc, 'acktr_sym_eig_jacobi_batched_f32')
        big_thread, big_err = None, []
        if self._big:
            for run, qt, dfl, off, doff, n in self._big:        # syevd works in place: running average -> Q^T slot
                qt[off:off + n * n].copy_(run[off:off + n * n])

            def solve_big():
                # side streams that fork from `stream` at this point; returns after they have drained (host join)
                torch.cuda.set_device(dev)                      # a new host thread starts on device 0
                rc = L.acktr_sym_eig_rocsolver_multi_f32(self._big_jobs, len(self._big), _hip.dev_ptr(self.e_work),
                                                         _hip.dev_ptr(self.info, torch.int32), 1e-6, stream)
                if rc != 0:
                    big_err.append((rc, L.acktr_last_error().decode('utf-8', 'replace')))
            if self._blk:      # overlap the host-driven rocSOLVER chain with the Jacobi launches (ctypes drops the GIL)
                import threading
                big_thread = threading.Thread(target=solve_big)
                big_thread.start()
            else:
                solve_big()
        if self._blk:
            rc = L.acktr_sym_eig_jacobi_block_f32(self._blk_jobs, len(self._blk), 1e-6, _hip.dev_ptr(self._blk_ws),
                                                  self._blk_ws.numel(), 0, self._blk_sweeps, stream)
            if big_thread is not None:
                big_thread.join()
            _hip.check(rc, 'acktr_sym_eig_jacobi_block_f32')
            # a factor that has not met the rotation criterion after the sweep limit is reported through a negative sweep
            # count, not through rc: the reference's symeig converges or raises, so such a factor is solved again by
            # rocSOLVER instead of passing a half-rotated basis to the preconditioner
            bad = [w for w, sw in zip(self._blk, self._blk_sweeps) if sw < 0]
            if bad:
                self._solve_rocsolver(bad)
        if big_err:
            raise RuntimeError('acktr_sym_eig_rocsolver_multi_f32 failed (rc=%d): %s' % big_err[0])
        if side is not None:
            torch.cuda.current_stream(dev).wait_stream(side)
        self._exchange_eig()

    def _exchange_eig(self):
        """Data-parallel return of the sharded eigendecompositions: every factor was decomposed by exactly
        one rank; each owner packs its (Q^T, d) slots into one contiguous message and broadcasts it -- 16.7 MB in total
        for NatureCNN, one collective per owner (round 1 zeroed the foreign slots and SUM all-reduced the four whole
        buckets: 33 MB through a reduction nobody needed).  Pure tensor logic: tests/test_parallel_cpu.py drives it with
        gloo at world 4 on CPU tensors."""
        if not self.dp.active:
            return
        dev = self.qt_a.device
        per_owner = {}
        for w, own in zip(self._eig_work, self._eig_owner):
            per_owner.setdefault(own, []).append(w)
        for own in sorted(per_owner):
            segs = []
            for run, qt, dfl, off, doff, n in per_owner[own]:
                segs.append(qt[off:off + n * n])
                segs.append(dfl[doff:doff + n])
            total = sum(t_.numel() for t_ in segs)
            if self._bcast_buf is None or self._bcast_buf.numel() < total or self._bcast_buf.device != dev:
                self._bcast_buf = torch.empty(total, dtype=f32, device=dev)
            buf = self._bcast_buf[:total]
            if own == self.dp.rank:
                torch.cat(segs, out=buf)
            self.dp.broadcast_(buf, src=own)
            if own != self.dp.rank:
                o = 0
                for t_ in segs:
                    t_.copy_(buf[o:o + t_.numel()])
                    o += t_.numel()

    def _solve_rocsolver(self, work):
        """syevd on the listed (run, qt, d, off, doff, n) factors on the current stream (fallback path)."""
        L = _hip.lib()
        dev = self.aa_run.device
        for run, qt, dfl, off, doff, n in work:
            qt[off:off + n * n].copy_(run[off:off + n * n])
        jobs = (_hip.EigJob * len(work))(*[_hip.EigJob(qt.data_ptr() + 4 * off, dfl.data_ptr() + 4 * doff, n, 0)
                                           for run, qt, dfl, off, doff, n in work])
        e_work = torch.zeros(sum(w[5] for w in work), dtype=f32, device=dev)
        info = torch.zeros(len(work), dtype=torch.int32, device=dev)
        rc = L.acktr_sym_eig_rocsolver_multi_f32(jobs, len(work), _hip.dev_ptr(e_work), _hip.dev_ptr(info, torch.int32),
                                                 1e-6, _hip.cur_stream(dev))
        _hip.check(rc, 'acktr_sym_eig_rocsolver_multi_f32')
        if int(info.abs().sum().item()) != 0:
            raise RuntimeError('KFAC eigendecomposition did not converge (blocked Jacobi and rocSOLVER syevd both failed)')

    def _precond_jobs(self, pol, gflat):
        """ctypes job table of the batched preconditioner: per layer its gradient slot, eigenbases / eigenvalues, the slot
        of the update vector, and two scratch slots of the same size (flat buffers laid out like the parameters)."""
        key = (gflat.data_ptr(), self.v_flat.data_ptr(), self.qt_g.data_ptr(), self.qt_a.data_ptr())
        if self.__dict__.get('_pj_key') != key:
            if self.__dict__.get('_pj_tmp') is None or self._pj_tmp[0].numel() != gflat.numel():
                self._pj_tmp = (torch.zeros_like(gflat), torch.zeros_like(gflat))
            arr = (_hip.PrecondJob * len(self._mods))()
            for i, m in enumerate(self._mods):
                arr[i] = _hip.PrecondJob(
                    _hip.dev_ptr(pol.grad_view(m.param, gflat)), ctypes.c_void_p(self.qt_g.data_ptr() + 4 * m.og),
                    ctypes.c_void_p(self.d_g_flat.data_ptr() + 4 * m.dg_off), ctypes.c_void_p(self.qt_a.data_ptr() + 4 * m.oa),
                    ctypes.c_void_p(self.d_a_flat.data_ptr() + 4 * m.da_off),
                    _hip.dev_ptr(pol.grad_view(m.param, self._pj_tmp[0])), _hip.dev_ptr(pol.grad_view(m.param, self._pj_tmp[1])),
                    _hip.dev_ptr(pol.grad_view(m.param, self.v_flat)), m.ng, m.na,
                    ctypes.c_void_p(self.q_g.data_ptr() + 4 * m.og), ctypes.c_void_p(self.q_a.data_ptr() + 4 * m.oa))
            self._pj_arr, self._pj_key = arr, key
        return self._pj_arr

    # ------------------------------------------------------------------ step (kfac.py:190-242)
    def step(self):
        pol = self.model
        flat, gflat = pol.flat_params, pol.flat_grads
        dev = flat.device
        self._setup(dev)
        L = _hip.lib()
        stream = _hip.cur_stream(dev)
        with torch.no_grad():
            pol.adopt_grads()
            if self.weight_decay > 0:
                gflat.add_(flat, alpha=self.weight_decay)         # kfac.py:192-194 (never enabled by the reference driver)
            la = self.damping + self.weight_decay
            if self._n_stats['a'] == 0 or self._n_stats['g'] == 0:
                # the reference fails here too (KeyError on the empty m_aa / m_gg dicts, kfac.py:205-214)
                raise RuntimeError('KFACOptimizer.step(): no Kronecker-factor statistics have been accumulated (m_aa: %d, '
                                   'm_gg: %d updates).  Drive it like a2c_acktr.py:53-78: evaluate_actions() with grad '
                                   'enabled, then `acc_stats = True; fisher_loss.backward(retain_graph=True); acc_stats = '
                                   'False` -- or call A2C_ACKTR.update().' % (self._n_stats['a'], self._n_stats['g']))
            if self.steps % self.Tf == 0:
                self._eigendecompose()
                # Q next to Q^T: with both, every operand of the four preconditioner products is contiguous along its output
                # index (stationary-accumulator stages, kfac.hip); 24 small transposes every Tf steps
                for m in self._mods:
                    for qt, q, off, n in ((self.qt_a, self.q_a, m.oa, m.na), (self.qt_g, self.q_g, m.og, m.ng)):
                        if n > 1:
                            _hip.check(L.acktr_transpose_f32(ctypes.c_void_p(qt.data_ptr() + 4 * off), n, n,
                                                             ctypes.c_void_p(q.data_ptr() + 4 * off), stream), 'acktr_transpose_f32')
                        else:
                            q[off:off + 1].copy_(qt[off:off + 1])
            # kfac.py:216-227 for every layer: stage s of all layers in one grouped launch (acktr_kfac_precondition_batched_f32)
            jobs = self._precond_jobs(pol, gflat)
            with ops._Timed('kfac_precondition_all', sum(4.0 * m.ng * m.na * (m.ng + m.na) for m in self._mods)):
                rc = L.acktr_kfac_precondition_batched_f32(jobs, len(self._mods), float(la), stream)
            _hip.check(rc, 'acktr_kfac_precondition_batched_f32')
            ws = ops.workspace(dev, L.acktr_optim_workspace_bytes(), 'optim')
            rc = L.acktr_kfac_kl_clip_f32(_hip.dev_ptr(self.v_flat), _hip.dev_ptr(gflat), flat.numel(), float(self.lr),
                                          float(self.kl_clip), _hip.dev_ptr(self.nu_buf), _hip.dev_ptr(ws), stream)
            _hip.check(rc, 'acktr_kfac_kl_clip_f32')
            first = self._mom_buf is None
            if first:
                self._mom_buf = torch.zeros_like(flat)
            rc = L.acktr_sgd_momentum_f32(_hip.dev_ptr(flat), _hip.dev_ptr(gflat), _hip.dev_ptr(self.v_flat),
                                          _hip.dev_ptr(self._mom_buf), flat.numel(),
                                          ctypes.c_void_p(self.nu_buf.data_ptr() + 4), float(self.sgd_lr),
                                          float(self.momentum), 1 if first else 0, stream)
            _hip.check(rc, 'acktr_sgd_momentum_f32')
        self.steps += 1

    def zero_grad(self, set_to_none=False):
        """optimizer.zero_grad() of a2c_acktr.py:70: afterwards every p.grad is a zeroed view of the flat gradient bucket,
        so the backward() that follows accumulates straight into what step() reads."""
        self.model.flat_grads.zero_()
        self.model.bind_grads()
