"""ctypes binding of libacktr_hip.so (the C-ABI declared in include/acktr_hip.h).

PyTorch is only plumbing here: device memory (``tensor.data_ptr()``) and the
current HIP stream.  There is NO fallback: if the library is missing or a
tensor is not on the GPU the call raises.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('ACKTR_HIP_LIB') or os.path.join(os.path.dirname(_HERE), 'lib', 'libacktr_hip.so')
ABI_VERSION = 1

_lib = None

c_p = ctypes.c_void_p
c_i = ctypes.c_int
c_l = ctypes.c_long
c_f = ctypes.c_float


class LossArgs(ctypes.Structure):
    """struct acktr_loss_args of include/acktr_hip.h (field order and types must match)."""
    _fields_ = [('algo', c_i), ('dist', c_i), ('B', c_i), ('A', c_i),
                ('values', c_p), ('ld_values', c_l), ('dist_in', c_p), ('ld_dist', c_l),
                ('logstd', c_p), ('actions', c_p),
                ('old_logp', c_p), ('old_values', c_p), ('returns', c_p), ('adv', c_p),
                ('gather_idx', c_p),
                ('clip', c_f), ('value_coef', c_f), ('entropy_coef', c_f), ('inv_batch', c_f),
                ('use_clipped_value_loss', c_i),
                ('dvalues', c_p), ('ld_dvalues', c_l), ('ddist', c_p), ('ld_ddist', c_l),
                ('dlogstd', c_p), ('dlogstd_rows', c_p), ('logp_out', c_p), ('loss_acc', c_p),
                ('workspace', c_p), ('workspace_bytes', c_l)]

class MlpStepArgs(ctypes.Structure):
    """struct acktr_mlp_step_args (include/acktr_hip.h)"""
    _fields_ = ([(n, c_p) for n in ('wc0', 'bc0', 'wc2', 'bc2', 'wa0', 'ba0', 'wa2', 'ba2', 'wv', 'bv', 'wm', 'bm', 'logstd',
                                    'gwc0', 'gbc0', 'gwc2', 'gbc2', 'gwa0', 'gba0', 'gwa2', 'gba2', 'gwv', 'gbv', 'gwm', 'gbm',
                                    'glogstd', 'obs', 'gather_idx')] +
                [('B', c_i), ('IN', c_i), ('H', c_i), ('A', c_i), ('algo', c_i)] +
                [(n, c_p) for n in ('actions', 'old_logp', 'old_values', 'returns', 'adv')] +
                [('clip', c_f), ('value_coef', c_f), ('entropy_coef', c_f), ('inv_batch', c_f),
                 ('use_clipped_value_loss', c_i), ('loss_acc', c_p)] +
                [(n, c_p) for n in ('opt_params', 'opt_grads', 'opt_exp_avg', 'opt_exp_avg_sq')] +
                [('opt_n', c_l), ('opt_max_norm', c_f), ('opt_hyper_dev', c_p), ('opt_step_dev', c_p),
                 ('categorical', c_i), ('actions_i64', c_p)])


class ConvPackJob(ctypes.Structure):
    """struct acktr_conv_pack_job (include/acktr_hip.h)"""
    _fields_ = [('w', c_p), ('workspace', c_p), ('workspace_bytes', c_l),
                ('C', c_i), ('H', c_i), ('W', c_i), ('OC', c_i), ('KH', c_i), ('KW', c_i), ('S', c_i), ('dgrad', c_i)]


class JacobiJob(ctypes.Structure):
    """struct acktr_jacobi_job (include/acktr_hip.h); also the element of the device descriptor array of the LDS kernel"""
    _fields_ = [('a', c_p), ('d', c_p), ('qt', c_p), ('n', c_i)]


class HeadFin(ctypes.Structure):
    """struct acktr_head_fin (include/acktr_hip.h): a head step's pending finalize"""
    _fields_ = [('slabs', c_p), ('nb', c_i), ('stride', c_l), ('n4', c_i), ('nw', c_i), ('OUT', c_i),
                ('gw', c_p), ('gb', c_p), ('loss_acc', c_p), ('inv_batch', c_f)]


class SlabFin(ctypes.Structure):
    """struct acktr_slab_fin (include/acktr_hip.h): a weight gradient's pending slab sum"""
    _fields_ = [('dst', c_p), ('slabs', c_p), ('nslabs', c_i), ('n4', c_l), ('stride', c_l), ('scale', c_f)]


class PackMap(ctypes.Structure):
    """struct acktr_pack_map (include/acktr_hip.h): where the optimiser also stores conv weights (weight-image slots)"""
    _fields_ = [('lo', c_l), ('hi', c_l), ('t1', c_p), ('t2', c_p), ('base', c_p)]


class PrecondJob(ctypes.Structure):
    """struct acktr_precond_job (include/acktr_hip.h)"""
    _fields_ = [('grad', c_p), ('qt_g', c_p), ('d_g', c_p), ('qt_a', c_p), ('d_a', c_p), ('tmp1', c_p), ('tmp2', c_p),
                ('v_out', c_p), ('OUT', c_i), ('IN', c_i), ('q_g', c_p), ('q_a', c_p)]


class EigJob(ctypes.Structure):
    """struct acktr_eig_job (include/acktr_hip.h)"""
    _fields_ = [('a_inout_qt', c_p), ('d', c_p), ('n', c_i), ('reserved', c_i)]


# name -> (restype, argtypes); mirrors include/acktr_hip.h one to one
_PROTOS = {
    'acktr_abi_version': (c_i, []),
    'acktr_target_arch': (ctypes.c_char_p, []),
    'acktr_last_error': (ctypes.c_char_p, []),
    'acktr_compute_returns_f32': (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_f, c_f, c_i, c_i, c_p]),
    'acktr_adv_normalize_workspace_bytes': (c_l, []),
    'acktr_adv_normalize_f32': (c_i, [c_p, c_p, c_p, c_l, c_p, c_p]),
    'acktr_adv_normalize_noalloc_max_n': (c_l, []),
    'acktr_adv_normalize_noalloc_init': (c_i, []),
    'acktr_adv_normalize_noalloc_f32': (c_i, [c_p, c_p, c_p, c_l, c_p]),
    'acktr_gather_rows': (c_i, [c_p, c_p, c_p, c_l, c_l, c_l, c_p]),
    'acktr_conv2d_fwd_workspace_bytes': (c_l, [c_i] * 7),
    'acktr_conv2d_fwd_f32': (c_i, [c_p, c_p, c_p, c_p, c_p] + [c_i] * 8 + [c_f, c_i, c_p, c_l, c_i, c_p]),
    'acktr_conv2d_gate_mask_bytes': (c_l, [c_i] * 4),
    'acktr_conv2d_fwd_gate_f32': (c_i, [c_p, c_p, c_p, c_p, c_p] + [c_i] * 8 + [c_f, c_i, c_p, c_l, c_i, c_p, c_p]),
    'acktr_conv2d_wgrad_workspace_bytes': (c_l, [c_i] * 8),
    'acktr_conv2d_wgrad_f32': (c_i, [c_p, c_p, c_p, c_p, c_p] + [c_i] * 8 + [c_f, c_p, c_l, c_p]),
    'acktr_conv1_fwd_u8': (c_i, [c_p, c_p, c_p, c_p, c_p, c_i, c_f, c_i, c_p, c_l, c_i, c_p]),
    'acktr_conv1_fwd_u8_gate': (c_i, [c_p, c_p, c_p, c_p, c_p, c_i, c_f, c_i, c_p, c_l, c_i, c_p, c_p]),
    'acktr_conv1_wgrad_u8_workspace_bytes': (c_l, [c_i]),
    'acktr_conv1_wgrad_u8': (c_i, [c_p, c_p, c_p, c_p, c_p, c_i, c_f, c_p, c_l, ctypes.POINTER(SlabFin), c_p]),
    'acktr_kfac_cov_a_conv1_u8': (c_i, [c_p, c_p, c_i, c_f, c_p, c_p, c_l, c_p]),
    'acktr_conv2d_wgrad_deferred_f32': (c_i, [c_p, c_p, c_p, c_p, c_p] + [c_i] * 8 + [c_f, c_p, c_l, ctypes.POINTER(SlabFin),
                                              c_p]),
    'acktr_conv2d_dgrad_workspace_bytes': (c_l, [c_i] * 8),
    'acktr_conv2d_dgrad_f32': (c_i, [c_p, c_p, c_p, c_p] + [c_i] * 8 + [c_p, c_l, c_i, c_p]),
    'acktr_conv2d_dgrad_padded_dy_offset_bytes': (c_l, [c_i] * 7),
    'acktr_conv2d_dgrad_ex_f32': (c_i, [c_p, c_p, c_p, c_p] + [c_i] * 8 + [c_p, c_l, c_i, c_p, c_i, c_p]),
    'acktr_conv2d_dgrad_carry_f32': (c_i, [c_p, c_p, c_p, c_p] + [c_i] * 8 + [c_p, c_l, c_i, c_p, c_i, ctypes.POINTER(SlabFin), c_p]),
    'acktr_linear_dgrad_ex_f32': (c_i, [c_p, c_p, c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_p, c_i, c_i, c_i, c_i, c_p]),
    'acktr_conv2d_pack_weights_f32': (c_i, [c_p, c_i, c_p]),
    'acktr_linear_fwd_workspace_bytes': (c_l, [c_i, c_i, c_i]),
    'acktr_linear_fwd_f32': (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_l, c_p]),
    'acktr_linear_dgrad_f32': (c_i, [c_p, c_p, c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_p]),
    'acktr_linear_wgrad_workspace_bytes': (c_l, [c_i, c_i, c_i]),
    'acktr_linear_wgrad_f32': (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_p, c_l, c_p]),
    'acktr_gemm_f32': (c_i, [c_p, c_l, c_l, c_p, c_l, c_l, c_p, c_l, c_i, c_i, c_i, c_f, c_i, c_p]),
    'acktr_policy_loss_workspace_bytes': (c_l, [c_i, c_i]),
    'acktr_policy_loss_f32': (c_i, [ctypes.POINTER(LossArgs), c_p]),
    'acktr_linear_fwd_slabs_workspace_bytes': (c_l, [c_i, c_i, c_i]),
    'acktr_linear_fwd_slabs_f32': (c_i, [c_p, c_p, c_i, c_i, c_i, c_p, c_l, ctypes.POINTER(ctypes.c_int), c_p]),
    'acktr_head_step_slabs_f32': (c_i, [ctypes.POINTER(LossArgs), c_p, c_i, c_l, c_p, c_p, c_p, c_p, c_p, c_p, c_i,
                                        ctypes.POINTER(HeadFin), c_p]),
    'acktr_head_step_supported': (c_i, [c_i, c_i, c_i, c_i]),
    'acktr_head_step_workspace_bytes': (c_l, [c_i, c_i, c_i]),
    'acktr_head_step_f32': (c_i, [ctypes.POINTER(LossArgs), c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_p]),
    'acktr_head_step_deferred_f32': (c_i, [ctypes.POINTER(LossArgs), c_p, c_p, c_p, c_p, c_p, c_p, c_i,
                                           ctypes.POINTER(HeadFin), c_p]),
    'acktr_dist_fwd_f32': (c_i, [c_i, c_p, c_l, c_p, c_p, c_i, c_i, c_p, c_p, c_p, c_p]),
    'acktr_dist_act_f32': (c_i, [c_i, c_p, c_l, c_p, c_p, c_i, c_i, c_i, c_p, c_p, c_p]),
    'acktr_dist_bwd_f32': (c_i, [c_i, c_p, c_l, c_p, c_p, c_p, c_p, c_i, c_i, c_p, c_l, c_p, c_p, c_p, c_l, c_p]),
    'acktr_optim_workspace_bytes': (c_l, []),
    'acktr_clip_adam_f32': (c_i, [c_p, c_p, c_p, c_p, c_l, c_f, c_f, c_f, c_f, c_f, c_f, c_p, c_p, c_p]),
    'acktr_clip_adam_dev_f32': (c_i, [c_p, c_p, c_p, c_p, c_l, c_f, c_p, c_p, c_p, c_p]),
    'acktr_clip_adam_dev_head_f32': (c_i, [c_p, c_p, c_p, c_p, c_l, c_f, c_p, c_p, ctypes.POINTER(HeadFin),
                                           ctypes.POINTER(SlabFin), c_p, c_p]),
    'acktr_clip_adam_dev_pack_f32': (c_i, [c_p, c_p, c_p, c_p, c_l, c_f, c_p, c_p, ctypes.POINTER(HeadFin),
                                           ctypes.POINTER(SlabFin), ctypes.POINTER(PackMap), c_p, c_p]),
    'acktr_clip_rmsprop_f32': (c_i, [c_p, c_p, c_p, c_l, c_f, c_f, c_f, c_f, c_p, c_p, c_p]),
    'acktr_sgd_momentum_f32': (c_i, [c_p, c_p, c_p, c_p, c_l, c_p, c_f, c_f, c_i, c_p]),
    'acktr_kfac_cov_workspace_bytes': (c_l, [c_i, c_l]),
    'acktr_kfac_cov_a_conv_f32': (c_i, [c_p, c_p] + [c_i] * 7 + [c_f, c_p, c_p, c_l, c_p]),
    'acktr_kfac_cov_g_conv_f32': (c_i, [c_p, c_i, c_i, c_i, c_f, c_p, c_p, c_l, c_p]),
    'acktr_kfac_cov_dense_f32': (c_i, [c_p, c_l, c_i, c_i, c_f, c_p, c_p, c_l, c_p]),
    'acktr_kfac_patch_mean_f32': (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_p, c_p]),
    'acktr_spatial_sum_f32': (c_i, [c_p, c_i, c_i, c_p, c_p]),
    'acktr_kfac_running_stat_f32': (c_i, [c_p, c_p, c_l, c_f, c_f, c_i, c_p]),
    'acktr_jacobi_max_n': (c_i, []),
    'acktr_sym_eig_jacobi_batched_f32': (c_i, [c_p, c_i, c_i, c_f, c_p]),
    'acktr_jacobi_block_max_n': (c_i, []),
    'acktr_sym_eig_jacobi_block_workspace_bytes': (c_l, [ctypes.POINTER(c_i), c_i]),
    'acktr_sym_eig_jacobi_block_f32': (c_i, [ctypes.POINTER(JacobiJob), c_i, c_f, c_p, c_l, c_i, ctypes.POINTER(c_i), c_p]),
    'acktr_sym_eig_rocsolver_f32': (c_i, [c_p, c_i, c_p, c_p, c_p, c_f, c_p]),
    'acktr_sym_eig_rocsolver_multi_f32': (c_i, [ctypes.POINTER(EigJob), c_i, c_p, c_p, c_f, c_p]),
    'acktr_kfac_precondition_f32': (c_i, [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_f, c_p, c_p, c_p, c_p]),
    'acktr_kfac_precondition_batched_f32': (c_i, [ctypes.POINTER(PrecondJob), c_i, c_f, c_p]),
    'acktr_kfac_kl_clip_f32': (c_i, [c_p, c_p, c_l, c_f, c_f, c_p, c_p, c_p]),
    'acktr_gail_mix_concat_f32': (c_i, [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_p, c_p]),
    'acktr_gail_mix_concat_idx_f32': (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_p]),
    'acktr_gail_bce_seed_f32': (c_i, [c_p, c_i, c_i, c_p, c_p, c_p, c_p]),
    'acktr_gail_param_grads_f32': (c_i, [c_p] * 6 + [c_i] * 5 + [c_p] * 7),
    'acktr_gail_pen_seed_f32': (c_i, [c_p, c_i, c_i, c_f, c_p, c_p, c_p, c_p]),
    'acktr_gail_tanh_link_f32': (c_i, [c_i, c_p, c_p, c_p, c_p, c_l, c_i, c_p, c_p, c_p]),
    'acktr_gail_reward_f32': (c_i, [c_p, c_l, c_p, c_p]),
    'acktr_gail_returns_moments_f32': (c_i, [c_p, c_p, c_i, c_i, c_f, c_p, c_p, c_p, c_p]),
    'acktr_gail_rms_merge_f64': (c_i, [c_p, c_i, ctypes.c_double, c_p, c_p, c_p]),
    'acktr_gail_scale_rows_f32': (c_i, [c_p, c_p, c_i, c_i, c_p]),
    'acktr_mlp_fused_step_supported': (c_i, [c_i, c_i, c_i, c_i]),
    'acktr_mlp_fused_step_f32': (c_i, [ctypes.POINTER(MlpStepArgs), c_p]),
    'acktr_gru_mask_rows_f32': (c_i, [c_p, c_p, c_p, c_i, c_i, c_p]),
    'acktr_gru_gates_fwd_f32': (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_p]),
    'acktr_gru_gates_bwd_f32': (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_p]),
    'acktr_gru_step_supported': (c_i, [c_i, c_i]),
    'acktr_gru_step_fwd_f32': (c_i, [c_p] * 8 + [c_i, c_i, c_p]),
    'acktr_gru_step_bwd_f32': (c_i, [c_p] * 8 + [c_i, c_i, c_i, c_p]),
    'acktr_transpose_f32': (c_i, [c_p, c_i, c_i, c_p, c_p]),
}


def register(name, restype, argtypes):
    """Used by sibling modules to declare further entry points of the same library."""
    _PROTOS[name] = (restype, argtypes)
    if _lib is not None:
        fn = getattr(_lib, name)
        fn.restype, fn.argtypes = restype, argtypes


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                'libacktr_hip.so not found at %s -- build it with '
                '`python pytorch-a2c-ppo-acktr-gail_amd/csrc/build.py` (hipcc, gfx950). '
                'There is no CPU fallback for this package.' % LIB_PATH)
        _lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOS.items():
            fn = getattr(_lib, name)          # AttributeError if the .so lacks a declared symbol
            fn.restype, fn.argtypes = res, args
        v = _lib.acktr_abi_version()
        if v != ABI_VERSION:
            raise RuntimeError('libacktr_hip.so ABI %d != expected %d (stale build?)' % (v, ABI_VERSION))
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().acktr_last_error().decode('utf-8', 'replace')
        raise RuntimeError('%s failed (rc=%d): %s' % (what, rc, msg))


def dev_ptr(t, dtype=None, name='tensor'):
    """Device pointer of a contiguous GPU tensor; raises for anything else (no silent copies)."""
    if t is None:
        return c_p(None)
    if not t.is_cuda:
        raise RuntimeError('%s must live on the MI355X device (got %s): this package has no CPU path'
                           % (name, t.device))
    if not t.is_contiguous():
        raise RuntimeError('%s must be contiguous' % name)
    if dtype is not None and t.dtype != dtype:
        raise RuntimeError('%s must be %s (got %s)' % (name, dtype, t.dtype))
    return c_p(t.data_ptr())


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def cur_stream(device=None):
    """torch's current stream on `device` as a hipStream_t.  (The raw-handle getter when this torch has it: building a
    torch.cuda.Stream object per launch was 8 % of the host time of an eager PPO step.)"""
    if _raw_stream is not None:
        if device is None:
            idx = torch.cuda.current_device()
        elif isinstance(device, int):
            idx = device
        else:
            idx = device.index if isinstance(device, torch.device) else torch.device(device).index
            if idx is None:
                idx = torch.cuda.current_device()
        return c_p(_raw_stream(idx))
    return c_p(torch.cuda.current_stream(device).cuda_stream)
