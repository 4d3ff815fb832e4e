"""ctypes binding of csrc/libmmidet_hip.so (C ABI: include/mmidet_hip.h).  Fails loudly when the library is missing:
there is no CPU/ATen fallback for any op of the hot path."""
import ctypes
import os

import torch  # noqa: F401  -- must come first: torch brings its own libamdhip64; loading ours first puts a second,
#                              device-less HIP runtime in the process ("no ROCm-capable device is detected")
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_uint64, c_void_p

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'csrc')
LIB_PATH = os.environ.get('MMIDET_HIP_LIB', os.path.join(CSRC, 'libmmidet_hip.so'))  # override: kernel A/B experiments

if not os.path.exists(LIB_PATH):
    raise ImportError('%s is missing: the HIP extension is not built (run `python __graft_entry__.py` or '
                      '`python mmi-det_amd/mmidet_hip/build.py`); the MI355X hot path has no fallback.' % LIB_PATH)

_lib = ctypes.CDLL(LIB_PATH)

ACT_NONE, ACT_SILU, ACT_LEAKY = 0, 1, 2


class ConvDesc(Structure):
    _fields_ = [(n, c_int32) for n in ('N', 'H', 'W', 'Cin', 'Ho', 'Wo', 'Cout', 'KH', 'KW', 'stride', 'pad', 'ldx',
                                       'ldy')]


class BnStats(Structure):
    """mmi_bn_stats (include/mmidet_hip.h)."""
    _fields_ = [('eps', c_float), ('momentum', c_float), ('running_mean', c_void_p), ('running_var', c_void_p),
                ('num_batches_tracked', c_void_p), ('num_batches_tracked2', c_void_p), ('mean_invstd', c_void_p)]


class LinearEpilogue(Structure):
    """mmi_linear_epilogue (include/mmidet_hip.h)."""
    _fields_ = [('kind', c_int32), ('ldaux', c_int32), ('ldaux_out', c_int32), ('p_drop', c_float), ('aux', c_void_p),
                ('aux_out', c_void_p), ('seed', c_uint64), ('seed_dev', c_void_p)]


EPI_NONE, EPI_DROPOUT_RESIDUAL, EPI_GELU, EPI_GELU_GRAD, EPI_ACCUMULATE = 0, 1, 2, 3, 4


class BnReduceHook(Structure):
    """mmi_bn_reduce_hook (include/mmidet_hip.h)."""
    _fields_ = [('y', c_void_p), ('ldy', c_int32), ('mean_invstd', c_void_p), ('mi_stride', c_int32), ('gamma', c_void_p),
                ('beta', c_void_p), ('act', c_int32), ('partials', c_void_p)]


BnReduceHookPair = BnReduceHook * 2


class BnMap(Structure):
    """mmi_bn_map (include/mmidet_hip.h): parameter blocks + output scatter of a BatchNorm pass over a multi-module buffer."""
    _fields_ = [('gamma', c_void_p * 4), ('beta', c_void_p * 4), ('dgamma', c_void_p * 4), ('dbeta', c_void_p * 4),
                ('nblk', c_int32), ('blk', c_int32), ('period', c_int32), ('split', c_int32), ('ls0', c_int32), ('ls1', c_int32)]


PtrPair = c_void_p * 2
BnStatsPair = BnStats * 2


def ptr_pair(a, b):
    return PtrPair(a, b)


P = c_void_p
_SIGS = {
    'mmi_version': (c_int, []),
    'mmi_last_error': (c_char_p, []),
    'mmi_workspace_header_bytes': (c_size_t, [c_int]),
    'mmi_split_t8': (c_int, [P, c_int, P, c_int, c_int64, c_int, P]),
    'mmi_gemm_operands_t8': (c_int, [P, P, P, P]),
    'mmi_conv_fwd_row_blocks': (c_int, [POINTER(ConvDesc)]),
    'mmi_set_streamk_slots': (c_int, [c_int]),
    'mmi_set_tile_override': (c_int, [c_int, c_int]),
    'mmi_set_wgrad_override': (c_int, [c_int, c_int, c_int]),
    'mmi_set_gemm_precision': (c_int, [c_int]),
    'mmi_set_uniform_loaders': (c_int, [c_int]),
    'mmi_set_deep_prefetch': (c_int, [c_int]),
    'mmi_conv_fwd_workspace': (c_size_t, [POINTER(ConvDesc)]),
    'mmi_conv_fwd': (c_int, [P, P, P, P, P, P, c_size_t, POINTER(ConvDesc), P]),
    'mmi_conv_bn_fwd': (c_int, [P, P, P, P, POINTER(BnStats), P, c_size_t, POINTER(ConvDesc), P]),
    'mmi_bn_act_fwd_split': (c_int, [P, c_int, P, P, P, P, c_int, P, c_int, P, c_int, c_int, c_int64, c_int, c_int, P]),
    'mmi_bn_act_bwd_workspace': (c_size_t, [c_int64, c_int]),
    'mmi_bn_act_bwd': (c_int, [P, c_int, P, c_int, P, c_int, c_int, P, P, P, P, c_size_t, P, c_int, P, P, P, P, c_int64, c_int,
                               c_int, c_int, P]),
    'mmi_conv_bias_act_fwd': (c_int, [P, P, P, P, c_int, c_int, P, P, c_size_t, POINTER(ConvDesc), P]),
    'mmi_detect_decode': (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, c_int64, c_int64, c_float, P, P]),
    'mmi_nms_workspace': (c_size_t, [c_int, c_int64, c_int, c_int]),
    'mmi_nms': (c_int, [P, c_int, c_int64, c_int, c_float, c_float, P, c_int, c_int, c_int, c_float, P, c_size_t, P, P,
                        P]),
    'mmi_linear_fwd_fused': (c_int, [P, P, P, P, P, c_size_t, POINTER(ConvDesc), POINTER(LinearEpilogue), P]),
    'mmi_linear_dgrad_fused': (c_int, [P, P, P, P, c_size_t, POINTER(ConvDesc), POINTER(LinearEpilogue), P]),
    'mmi_conv_dgrad_workspace': (c_size_t, [POINTER(ConvDesc)]),
    'mmi_conv_dgrad': (c_int, [P, P, P, P, c_size_t, POINTER(ConvDesc), P]),
    'mmi_conv_wgrad_workspace': (c_size_t, [POINTER(ConvDesc)]),
    'mmi_conv_wgrad': (c_int, [P, P, P, P, P, c_size_t, POINTER(ConvDesc), P]),
    'mmi_conv_wgrad_table_bytes': (c_size_t, [POINTER(ConvDesc)]),
    'mmi_conv_wgrad_table_build': (c_int, [P, POINTER(ConvDesc), P]),
    'mmi_conv_wgrad_tab': (c_int, [P, P, P, P, P, c_size_t, P, POINTER(ConvDesc), P]),
    'mmi_bn_act_fwd_map': (c_int, [P, c_int, P, POINTER(BnMap), P, c_int, P, c_int, P, c_int, c_int64, c_int, c_int, P]),
    'mmi_bn_act_bwd_map': (c_int, [P, c_int, P, c_int, P, c_int, P, POINTER(BnMap), P, c_size_t, P, c_int, c_int64, c_int, c_int, c_int, P]),
    'mmi_conv_fwd_row_blocks_n': (c_int, [POINTER(ConvDesc), c_int]),
    'mmi_conv_fwd_workspace_n': (c_size_t, [POINTER(ConvDesc), c_int]),
    'mmi_conv_dgrad_workspace_n': (c_size_t, [POINTER(ConvDesc), c_int]),
    'mmi_conv_wgrad_workspace_n': (c_size_t, [POINTER(ConvDesc), c_int]),
    'mmi_conv_bn_fwd2': (c_int, [POINTER(PtrPair), POINTER(PtrPair), POINTER(PtrPair), POINTER(PtrPair), POINTER(BnStatsPair), c_int, P, c_size_t,
                                 POINTER(ConvDesc), P]),
    'mmi_conv_dgrad2': (c_int, [POINTER(PtrPair), POINTER(PtrPair), POINTER(PtrPair), POINTER(PtrPair), c_int, P, c_size_t, POINTER(ConvDesc), P]),
    'mmi_conv_dgrad_row_blocks_n': (c_int, [POINTER(ConvDesc), c_int]),
    'mmi_conv_dgrad_bnred': (c_int, [P, P, P, P, c_int, POINTER(BnReduceHook), P, c_size_t, POINTER(ConvDesc), P]),
    'mmi_conv_dgrad2_bnred': (c_int, [POINTER(PtrPair), POINTER(PtrPair), POINTER(PtrPair), POINTER(PtrPair), c_int, POINTER(BnReduceHookPair), P,
                                      c_size_t, POINTER(ConvDesc), P]),
    'mmi_bn_act_bwd_apply_map': (c_int, [P, c_int, P, c_int, P, c_int, P, POINTER(BnMap), POINTER(PtrPair), c_int, P, c_int, c_int64, c_int,
                                         c_int, c_int, P]),
    'mmi_conv_wgrad2': (c_int, [POINTER(PtrPair), POINTER(PtrPair), POINTER(PtrPair), POINTER(PtrPair), P, c_size_t, P, POINTER(ConvDesc), P]),
    'mmi_bn_finalize': (c_int, [P, c_int, c_int64, c_int, c_float, c_float, P, P, P, P, P]),
    'mmi_bn_eval_stats': (c_int, [P, P, c_int, c_float, P, P]),
    'mmi_bn_act_fwd': (c_int, [P, c_int, P, P, P, P, c_int, P, c_int, c_int64, c_int, c_int, P]),
    'mmi_bn_bwd_parts': (c_int, [c_int64]),
    'mmi_bn_act_bwd_reduce': (c_int, [P, c_int, P, c_int, P, P, P, P, c_int64, c_int, c_int, P]),
    'mmi_bn_act_bwd_apply': (c_int, [P, c_int, P, c_int, P, P, P, P, c_int, P, c_int, P, P, c_int64, c_int, c_int,
                                     c_int, P]),
    'mmi_colsum': (c_int, [P, c_int, c_int64, c_int, P, P, P]),
    'mmi_nchw_to_nhwc': (c_int, [P, c_int64, c_int64, c_int64, c_int64, P, c_int, c_int, c_int, c_int, P]),
    'mmi_u8_pair_to_nhwc': (c_int, [P, P, P, c_int, c_int, c_int, P]),
    'mmi_nhwc_to_nchw': (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    'mmi_space_to_depth': (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, P]),
    'mmi_space_to_depth_ld': (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    'mmi_head_permute': (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, P]),
    'mmi_add': (c_int, [P, c_int, P, c_int, P, c_int, c_int64, c_int, P]),
    'mmi_copy2d': (c_int, [P, c_int, P, c_int, c_int64, c_int, P]),
    'mmi_upsample2x': (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    'mmi_upsample2x_ld': (c_int, [P, c_int, P, c_int, c_int, c_int, c_int, c_int, P]),
    'mmi_upsample2x_bwd': (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    'mmi_upsample2x_bwd_acc': (c_int, [P, c_int, P, c_int, P, c_int, c_int, c_int, c_int, P]),
    'mmi_spp_pool_fwd': (c_int, [P, c_int, P, c_int, c_int, c_int, c_int, c_int, P]),
    'mmi_spp_pool_bwd': (c_int, [P, c_int, P, c_int, P, c_int, c_int, c_int, c_int, c_int, P]),
    'mmi_build_targets': (c_int, [P, c_int, P, c_int, c_int, P, c_float, P, P, P, P, P, P]),
    'mmi_dropout': (c_int, [P, P, c_int64, P, c_int64, c_float, c_uint64, P, P]),
    'mmi_seed_advance': (c_int, [P, P]),
    'mmi_gelu_fwd': (c_int, [P, P, c_int64, P]),
    'mmi_gelu_bwd': (c_int, [P, P, P, c_int64, P]),
    'mmi_sigmoid_fwd': (c_int, [P, P, c_int64, P]),
    'mmi_sigmoid_bwd': (c_int, [P, P, P, c_int64, P]),
    'mmi_mul': (c_int, [P, P, P, c_int64, P]),
    'mmi_scale': (c_int, [P, P, P, c_int64, P]),
    'mmi_layernorm_fwd': (c_int, [P, P, P, P, P, c_int, c_int, c_float, P]),
    'mmi_layernorm_bwd_parts': (c_int, [c_int]),
    'mmi_layernorm_bwd': (c_int, [P, P, P, P, P, P, P, P, c_int, c_int, P]),
    'mmi_layernorm_bwd_input': (c_int, [P, P, P, P, P, P, P, c_float, c_uint64, P, c_int, c_int, P]),
    'mmi_layernorm_bwd_params': (c_int, [P, P, P, P, P, P, c_int, c_int, P]),
    'mmi_attention_fwd': (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_float, c_uint64, P, P]),
    'mmi_attention_bwd': (c_int, [P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_float, c_uint64, P, P]),
    'mmi_attention_fwd_strided': (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_float, c_uint64, P, P]),
    'mmi_attention_bwd_strided': (c_int, [P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_float, c_uint64, P,
                                          P]),
    'mmi_avgpool8_fwd': (c_int, [P, c_int, c_int, c_int, c_int, c_int, P, c_int64, c_int, P]),
    'mmi_avgpool8_bwd': (c_int, [P, c_int64, c_int, P, c_int, c_int, c_int, c_int, c_int, P]),
    'mmi_avgpool8_bwd_acc': (c_int, [P, c_int64, c_int, P, c_int, P, c_int, c_int, c_int, c_int, c_int, P]),
    'mmi_upsample_add_fwd': (c_int, [P, c_int, P, c_int64, c_int, P, c_int, c_int, c_int, c_int, c_int, P]),
    'mmi_upsample_add_bwd': (c_int, [P, c_int, P, c_int64, c_int, c_int, c_int, c_int, c_int, P]),
    'mmi_ffm_highpass': (c_int, [P, P, c_int, c_int, c_uint64, P]),
    'mmi_separation_loss': (c_int, [P, P, P, P, c_int, P, P]),
    'mmi_fusion_stats_workspace': (c_size_t, []),
    'mmi_fusion_stats': (c_int, [P, c_int, P, c_int, P, c_int, c_int, c_int, c_int, P, P, P]),
    'mmi_conv_fwd_row_blocks_bf16': (c_int, [POINTER(ConvDesc)]),
    'mmi_conv_fwd_workspace_bf16': (c_size_t, [POINTER(ConvDesc)]),
    'mmi_conv_fwd_bf16': (c_int, [P, P, P, P, P, POINTER(BnStats), P, c_size_t, POINTER(ConvDesc), P]),
    'mmi_conv_dgrad_bf16': (c_int, [P, P, P, P, c_int, POINTER(ConvDesc), P]),
    'mmi_conv_wgrad_bf16': (c_int, [P, P, P, P, P, c_size_t, POINTER(ConvDesc), P]),
    'mmi_bn_act_fwd_split_bf16': (c_int, [P, c_int, P, P, P, P, c_int, P, c_int, P, c_int, c_int, c_int64, c_int, c_int, P]),
    'mmi_bn_act_bwd_bf16': (c_int, [P, c_int, P, c_int, P, c_int, c_int, P, P, P, P, c_size_t, P, c_int, P, P, P, P, c_int64, c_int,
                                    c_int, c_int, P]),
    'mmi_cast_f32_bf16': (c_int, [P, c_int, P, c_int, c_int64, c_int, P]),
    'mmi_cast_bf16_f32': (c_int, [P, c_int, P, c_int, c_int64, c_int, P]),
    'mmi_add_bf16': (c_int, [P, c_int, P, c_int, P, c_int, c_int64, c_int, P]),
    'mmi_copy2d_bf16': (c_int, [P, c_int, P, c_int, c_int64, c_int, P]),
    'mmi_upsample2x_bf16': (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    'mmi_upsample2x_bwd_bf16': (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    'mmi_cem_blocks': (c_int, [c_int, c_int, c_int]),
    'mmi_cem_bwd_mid_workspace': (c_size_t, [c_int, c_int, c_int]),
    'mmi_cem_bwd_mid_blocks': (c_int, [c_int, c_int, c_int]),
    'mmi_cem_bwd_mid': (c_int, [P, P, P, P, P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, P]),
    'mmi_cem_conv2_stats': (c_int, [P, c_int, P, P, c_int, c_int, c_int, P]),
    'mmi_cem_conv2_fwd_blocks': (c_int, [c_int, c_int, c_int]),
    'mmi_cem_conv2_wgrad_bn_workspace': (c_size_t, [c_int, c_int, c_int]),
    'mmi_cem_conv2_wgrad_bn': (c_int, [P, P, P, c_int, P, P, P, P, P, c_int, P, P, c_size_t, c_int, c_int, c_int, P]),
    'mmi_cem_conv2_fwd': (c_int, [P, c_int, P, P, P, c_int, c_int, c_int, P]),
    'mmi_cem_fwd_from_y2': (c_int, [P, P, P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, P]),
    'mmi_cem_fused_fwd': (c_int, [P, c_int, P, P, P, P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, P]),
    'mmi_comm_available': (c_int, []),
    'mmi_comm_unique_id': (c_int, [P]),
    'mmi_comm_init': (c_int, [c_int, c_int, P]),
    'mmi_comm_world': (c_int, []),
    'mmi_comm_rank': (c_int, []),
    'mmi_allreduce_bucket': (c_int, [P, c_int64, c_int, P]),
    'mmi_broadcast_bytes': (c_int, [P, c_int64, c_int, P]),
    'mmi_comm_destroy': (c_int, []),
    'mmi_sgd_ema_step': (c_int, [P, P, c_int, P, P]),
    'mmi_sobel_add_fwd': (c_int, [P, c_int, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    'mmi_sobel_add_bwd_workspace': (c_size_t, [c_int, c_int, c_int, c_int]),
    'mmi_sobel_add_bwd': (c_int, [P, c_int, P, P, P, c_int, P, P, P, c_int, c_int, c_int, c_int, P]),
    'mmi_detect_loss_workspace': (c_size_t, [c_int, c_int64, c_int64]),
    'mmi_detect_loss': (c_int, [P, P, P, c_int, c_int, c_int, c_int, P, P, P, P, P, c_int64, P, c_float, c_float, c_float,
                                c_float, c_float, c_float, P, c_int, c_float, c_int, P, c_size_t, P, P]),
}
_UNCHECKED = ('mmi_version', 'mmi_conv_dgrad_row_blocks_n', 'mmi_cem_blocks', 'mmi_cem_conv2_fwd_blocks', 'mmi_cem_bwd_mid_blocks', 'mmi_conv_fwd_row_blocks_bf16', 'mmi_comm_available', 'mmi_comm_world', 'mmi_comm_rank', 'mmi_conv_fwd_row_blocks', 'mmi_conv_fwd_row_blocks_n', 'mmi_set_streamk_slots', 'mmi_set_uniform_loaders', 'mmi_set_deep_prefetch', 'mmi_bn_bwd_parts', 'mmi_layernorm_bwd_parts')

EXPORTS = sorted(_SIGS)


_SPIN_US = float(os.environ.get('MMIDET_HOST_SPIN_US', '0'))


class MMIError(RuntimeError):
    pass


def _bind(name, restype, argtypes):
    fn = getattr(_lib, name)
    fn.restype = restype
    fn.argtypes = argtypes
    if restype is not c_int or name in _UNCHECKED:
        return fn

    def checked(*args):
        rc = fn(*args)
        if rc != 0:
            raise MMIError('%s failed (%d): %s' % (name, rc, _lib.mmi_last_error().decode()))
    if _SPIN_US > 0:        # experiment (tools/ab_env.sh MMIDET_HOST_SPIN_US=n): how sensitive is the step to host enqueue time?
        import time as _time

        def checked(*args):  # noqa: F811
            rc = fn(*args)
            if rc != 0:
                raise MMIError('%s failed (%d): %s' % (name, rc, _lib.mmi_last_error().decode()))
            t = _time.perf_counter() + _SPIN_US * 1e-6
            while _time.perf_counter() < t:
                pass
    checked.__name__ = name
    return checked


def register(sigs):
    """Bind further entry points (other modules of this package extend the table as kernels are added)."""
    for name, (restype, argtypes) in sigs.items():
        _SIGS[name] = (restype, argtypes)
        globals()[name[4:]] = _bind(name, restype, argtypes)
    EXPORTS[:] = sorted(_SIGS)


register(dict(_SIGS))

# Planner settings change which schedule (and how much workspace) a shape gets: callers that cache planner answers per shape
# (ops.fwd_plan ...) key them on this counter.
plan_epoch = [0]


def _bump(fn):
    def wrapped(*a):
        plan_epoch[0] += 1
        return fn(*a)
    wrapped.__name__ = getattr(fn, '__name__', 'setter')
    return wrapped


for _name in ('set_streamk_slots', 'set_tile_override', 'set_gemm_precision', 'set_uniform_loaders', 'set_wgrad_override', 'set_deep_prefetch'):
    globals()[_name] = _bump(globals()[_name])
