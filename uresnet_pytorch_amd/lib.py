"""ctypes binding of liburesnet_hip.so (C ABI declared in include/uresnet_hip.h).

The product path has no CPU fallback: if the library is missing or a call fails,
a RuntimeError is raised.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'liburesnet_hip.so')
_lib = None

c_void_p, c_int, c_i64, c_double = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_double

class _FinBN(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in ('gamma', 'beta', 'mean', 'invstd', 'scale', 'shift', 'running_mean',
                                                 'running_var')]


class GConvArgs(ctypes.Structure):
    """urn_gconv_args of include/uresnet_hip.h"""
    _fields_ = [('x', ctypes.c_void_p), ('wt', ctypes.c_void_p), ('tbl', ctypes.c_void_p), ('ld', ctypes.c_int64),
                ('K', ctypes.c_int), ('flip', ctypes.c_int), ('n_out', ctypes.c_int64), ('cin', ctypes.c_int),
                ('cout', ctypes.c_int), ('res', ctypes.c_void_p), ('y', ctypes.c_void_p),
                ('xf_scale', ctypes.c_void_p), ('xf_shift', ctypes.c_void_p), ('epilogue', ctypes.c_int),
                ('part', ctypes.c_void_p), ('e_x', ctypes.c_void_p), ('e_scale', ctypes.c_void_p),
                ('e_shift', ctypes.c_void_p), ('e_mean', ctypes.c_void_p), ('e_invstd', ctypes.c_void_p),
                ('sync_word', ctypes.c_void_p), ('fin_n', ctypes.c_int64), ('fin_eps', ctypes.c_double),
                ('fin_momentum', ctypes.c_double), ('fin_bn', _FinBN * 2), ('fin_dgamma', ctypes.c_void_p),
                ('fin_dbeta', ctypes.c_void_p), ('fin_coef0', ctypes.c_void_p), ('fin_coef1', ctypes.c_void_p),
                ('part_slots', ctypes.c_int), ('xs_slots', ctypes.c_int), ('xs_split', ctypes.c_int),
                ('xs_ld', ctypes.c_int * 2), ('xs_sums', ctypes.c_void_p * 2), ('xs_n', ctypes.c_int64),
                ('xs_gamma', ctypes.c_void_p), ('xs_beta', ctypes.c_void_p), ('xs_mean', ctypes.c_void_p),
                ('xs_invstd', ctypes.c_void_p), ('xs_scale', ctypes.c_void_p), ('xs_shift', ctypes.c_void_p),
                ('xs_running_mean', ctypes.c_void_p), ('xs_running_var', ctypes.c_void_p), ('precision', ctypes.c_int),
                ('ldx', ctypes.c_int64), ('ldy', ctypes.c_int64), ('pairs', ctypes.c_void_p), ('pairs_tile', ctypes.c_int),
                ('wt_frag', ctypes.c_void_p), ('wt_frag_prec', ctypes.c_int)]


class DenseGeom(ctypes.Structure):
    """urn_dense_geom of include/uresnet_hip.h (dimensions in z, y, x order)"""
    _fields_ = [('In', ctypes.c_int * 3), ('Out', ctypes.c_int * 3), ('Sub', ctypes.c_int * 3), ('p', ctypes.c_int * 3),
                ('os', ctypes.c_int * 3), ('s', ctypes.c_int * 3), ('nt', ctypes.c_int * 3), ('e', (ctypes.c_int * 3) * 3),
                ('wi', (ctypes.c_int * 3) * 3), ('kdim', ctypes.c_int * 3), ('mode', ctypes.c_int)]


# name -> (restype, argtypes); must list every symbol of include/uresnet_hip.h
SIGNATURES = {
    'urn_version': (c_int, []),
    'urn_last_error': (ctypes.c_char_p, []),
    'urn_hash_capacity': (c_i64, [c_i64]),
    'urn_hash_bytes': (c_i64, [c_i64]),
    'urn_unique_scratch_bytes': (c_i64, [c_i64]),
    'urn_hash_clear': (c_int, [c_void_p, c_i64, c_void_p]),
    'urn_sites_build': (c_int, [c_void_p, c_i64, c_int, c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_void_p,
                                c_void_p, c_void_p]),
    'urn_input_features': (c_int, [c_void_p, c_void_p, c_i64, c_int, c_void_p, c_i64, c_void_p, c_void_p,
                                   c_void_p]),
    'urn_rulebook_subm': (c_int, [c_void_p, c_void_p, c_i64, c_int, c_void_p, c_i64, c_void_p, c_i64, c_void_p,
                                  c_void_p]),
    'urn_level_down': (c_int, [c_void_p, c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_void_p,
                               c_void_p, c_void_p, c_void_p]),
    'urn_level_down_tables': (c_int, [c_void_p, c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_void_p,
                                      c_void_p, c_void_p, c_void_p, c_i64, c_void_p, c_i64, c_void_p]),
    'urn_levels_scratch_bytes': (c_i64, [c_i64, c_int]),
    'urn_sites_build_levels': (c_int, [c_void_p, c_i64, c_int, c_int, c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_void_p,
                                       c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_void_p]),
    'urn_rulebook_subm_multi': (c_int, [c_int, c_void_p, c_void_p, c_i64, c_void_p, c_void_p, c_i64, c_void_p, c_i64,
                                        c_void_p]),
    'urn_down_tables': (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_i64, c_void_p]),
    'urn_fill_i32': (c_int, [c_void_p, c_i64, ctypes.c_int32, c_void_p]),
    'urn_pairs_bytes': (c_i64, [c_i64, c_int, c_int]),
    'urn_pairs_build': (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'urn_gconv_fwd': (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_int, c_int, c_i64, c_int, c_int, c_void_p,
                              c_void_p, c_void_p]),
    'urn_gconv_bwd_dw': (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_int, c_i64, c_int, c_int, c_void_p,
                                 c_void_p]),
    'urn_gconv_dw_pairs_scratch_bytes': (c_i64, [c_i64, c_int, c_int, c_int, c_int]),
    'urn_gconv_bwd_dw_pairs': (c_int, [c_void_p, c_i64, c_void_p, c_void_p, c_void_p, c_i64, c_void_p, c_int, c_int, c_i64,
                                       c_int, c_int, c_void_p, c_void_p, c_i64, c_void_p]),
    'urn_weight_fragments': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    'urn_weight_fragments16': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    'urn_transpose_w': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    'urn_bn_scratch_bytes': (c_i64, [c_int]),
    'urn_bn_relu_fwd': (c_int, [c_void_p, c_i64, c_int, c_void_p, c_void_p, c_double, c_int, c_void_p, c_void_p,
                                c_void_p, c_void_p, c_void_p, c_double, c_void_p, c_void_p]),
    'urn_bn_relu_apply': (c_int, [c_void_p, c_i64, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                                  c_void_p]),
    'urn_bn_relu_bwd': (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_int, c_void_p, c_void_p, c_void_p, c_int,
                                c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'urn_gconv_part_bytes': (c_i64, [c_i64, c_int]),
    'urn_gconv_fwd_ex': (c_int, [c_void_p, ctypes.POINTER(c_int), c_void_p]),
    'urn_gconv_bwd_dw_ex': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_int, c_i64, c_int, c_int,
                                    c_void_p, c_void_p]),
    'urn_bn_stats_partial': (c_int, [c_void_p, c_i64, c_int, c_void_p, ctypes.POINTER(c_int), c_void_p]),
    'urn_bn_finalize_fwd': (c_int, [c_void_p, c_int, c_i64, c_int, c_int, c_double, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_void_p, c_void_p, c_void_p, c_double, c_void_p]),
    'urn_bn_finalize_bwd': (c_int, [c_void_p, c_int, c_i64, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'urn_bn_bwd_apply': (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_void_p, c_void_p]),
    'urn_bn_bwd_apply_sums': (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_i64, c_int, c_void_p, c_void_p, c_void_p,
                                      c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    'urn_gconv_bwd_dw_strided': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_void_p, c_i64, c_int, c_i64,
                                         c_int, c_int, c_void_p, c_void_p]),
    'urn_gconv_dw_2stage_scratch_bytes': (c_i64, [c_int, c_i64, c_int, c_int]),
    'urn_gconv_dw_2stage_scratch_max': (c_i64, []),
    'urn_gconv_bwd_dw_2stage': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_void_p, c_i64, c_int, c_i64,
                                        c_int, c_int, c_void_p, c_void_p, c_i64, c_void_p]),
    'urn_adam_flat': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_double, c_double, c_double, c_double,
                              c_double, c_i64, c_void_p]),
    'urn_rows_gather': (c_int, [c_void_p, c_void_p, c_i64, c_int, c_void_p, c_void_p]),
    'urn_rows_scatter_add': (c_int, [c_void_p, c_void_p, c_i64, c_int, c_void_p, c_void_p]),
    'urn_head_fwd': (c_int, [c_void_p, c_void_p, c_i64, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    'urn_head_bwd': (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                             c_void_p]),
    'urn_ce_scratch_bytes': (c_i64, []),
    'urn_ce_fwd': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_i64, c_int, c_void_p, c_void_p, c_void_p,
                           c_void_p]),
    'urn_ce_bwd': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_int,
                           c_void_p, c_void_p]),
    'urn_dense_conv_scratch_bytes': (c_i64, [c_int, c_int, c_void_p]),
    'urn_dense_conv': (c_int, [c_void_p, c_i64, c_int, c_void_p, c_void_p, c_void_p, c_i64, c_int, c_int, c_void_p, c_int, c_void_p,
                               c_int, c_void_p, c_void_p, c_void_p, c_i64, c_void_p]),
    'urn_dense_dw_scratch_bytes': (c_i64, [c_int, c_void_p, c_void_p, c_int, c_int]),
    'urn_dense_dw': (c_int, [c_void_p, c_i64, c_int, c_void_p, c_i64, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                             c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_i64, c_int, c_void_p]),
    'urn_dense_weight_layouts': (c_int, [c_int, c_void_p, c_void_p]),
    'urn_dense_fold': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    'urn_dense_conv_dgrad_fold': (c_int, [c_void_p, c_i64, c_int, c_void_p, c_void_p, c_void_p, c_i64, c_int, c_int, c_void_p, c_void_p,
                                          c_void_p, c_void_p, c_int, c_void_p, c_i64, c_void_p]),
    'urn_dense_bn_act_fwd': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_i64, c_int,
                                     c_void_p]),
    'urn_dense_bn_act_bwd_reduce': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_i64,
                                            c_int, c_void_p, c_void_p, c_int, c_void_p]),
    'urn_dense_ce_fwd': (c_int, [c_void_p, c_i64, c_void_p, c_void_p, c_void_p, c_i64, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    'urn_dense_ce_bwd': (c_int, [c_void_p, c_i64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_int, c_void_p,
                                 c_void_p]),
    'urn_dense_bn_bwd_finalize': (c_int, [c_void_p, c_int, c_int, c_i64, c_int, c_void_p, c_void_p]),
    'urn_dense_bn_act_bwd_apply': (c_int, [c_void_p] * 16 + [c_i64, c_int, c_void_p]),
    'urn_net_create': (c_int, [c_int, c_int, c_int, c_int, c_double, c_double, c_int, ctypes.POINTER(c_void_p)]),
    'urn_net_destroy': (None, [c_void_p]),
    'urn_net_prepare_weights': (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_void_p]),
    'urn_net_param_count': (c_i64, [c_void_p]),
    'urn_net_running_count': (c_i64, [c_void_p]),
    'urn_net_num_tensors': (c_int, [c_void_p]),
    'urn_net_tensor': (c_int, [c_void_p, c_int, ctypes.POINTER(c_i64), ctypes.POINTER(c_i64)]),
    'urn_net_workspace_bytes': (c_i64, [c_void_p, c_int, ctypes.POINTER(c_i64), c_i64, c_int]),
    'urn_net_set_pairs': (c_int, [c_void_p, c_int, ctypes.POINTER(c_void_p), ctypes.POINTER(c_void_p),
                                  ctypes.POINTER(c_void_p), ctypes.POINTER(c_int), ctypes.POINTER(c_int),
                                  ctypes.POINTER(c_int)]),
    'urn_net_forward': (c_int, [c_void_p, c_int, c_i64, ctypes.POINTER(c_i64), ctypes.POINTER(c_void_p),
                                ctypes.POINTER(c_void_p), ctypes.POINTER(c_void_p), c_void_p, c_i64, c_void_p,
                                c_void_p, c_void_p, c_void_p, c_i64, c_void_p, c_int, c_void_p]),
    'urn_net_num_bn': (c_int, [c_void_p]),
    'urn_net_bn_info': (c_int, [c_void_p, c_int, ctypes.POINTER(c_i64), ctypes.POINTER(c_i64), ctypes.POINTER(c_int)]),
    'urn_net_bn_export': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    'urn_net_probe': (c_int, [c_void_p, c_void_p]),
    'urn_net_backward': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    'urn_net_suffix_offset': (c_i64, [c_void_p]),
    'urn_net_side_stream': (c_void_p, [c_void_p]),
    'urn_net_backward_cb': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'urn_net_set_head': (c_int, [c_void_p, c_void_p, c_void_p]),
    'urn_tail_fwd': (c_int, [c_void_p, c_void_p, c_i64, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_i64, ctypes.c_double,
                             c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_double,
                             c_void_p, c_void_p]),
    'urn_tail_bwd': (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                             c_void_p, c_i64, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    'urn_set_option': (c_int, [ctypes.c_char_p, c_i64]),
    'urn_prof_enable': (c_int, [c_int]),
    'urn_prof_read': (c_int, [c_int, ctypes.POINTER(c_double), ctypes.POINTER(c_i64)]),
}


def load():
    """Load the shared library and bind every entry point (no GPU needed)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError('liburesnet_hip.so not built: run `python -c "import __graft_entry__ as g; g.build()"` '
                               '(expected at %s)' % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
        # developer switch for A/B runs of whole programs: URN_OPTIONS="key=value,key=value" -> urn_set_option
        for kv in filter(None, os.environ.get('URN_OPTIONS', '').split(',')):
            k, v = kv.split('=')
            if L.urn_set_option(k.strip().encode(), int(v)) != 0:
                raise RuntimeError('URN_OPTIONS: unknown option %r' % k)
    return _lib


def check(rc, what=''):
    if rc != 0:
        msg = load().urn_last_error()
        raise RuntimeError('liburesnet_hip %s failed (%d): %s' % (what, rc, msg.decode() if msg else ''))


def stream():
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  Tensors must be contiguous."""
    if t is None:
        return None
    assert t.is_contiguous(), 'non-contiguous tensor handed to the C ABI'
    return t.data_ptr()


def require_gpu(t):
    if not t.is_cuda:
        raise RuntimeError('uresnet_pytorch_amd: the HIP path needs tensors on the GPU (got %s); '
                           'there is no CPU fallback' % t.device)


PRECISIONS = {'fp32': 0, 'bf16': 1, 'fp16': 2}


def set_precision(name):
    """MFMA operand precision of the gather convolutions (forward, input gradient, weight gradient): the library
    default that calls without an explicit urn_gconv_args.precision use.  BASELINE configs[1] = 'bf16', configs[4] =
    'fp16'; tensors in HBM and the accumulation stay fp32."""
    global _PRECISION
    load().urn_set_option(b'gconv_precision', PRECISIONS[name])
    _PRECISION = PRECISIONS[name]


_PRECISION = 0


def precision():
    """the library default set by set_precision: 0 fp32, 1 bf16, 2 fp16"""
    return _PRECISION
