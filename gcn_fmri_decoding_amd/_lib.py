"""ctypes binding of libchebgcn.so (include/chebgcn.h).

The library is the product: there is no CPU or PyTorch fallback.  If the shared object
is missing or a call fails, an exception is raised -- never a silent alternative path.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('CHEBGCN_LIB') or os.path.join(_HERE, 'libchebgcn.so')   # override: kernel experiments only

BIAS_NONE, BIAS_FILTER, BIAS_VERTEX = 0, 1, 2
POOL_MAX, POOL_AVG = 0, 1


class ChebgcnError(RuntimeError):
    pass


_p = C.c_void_p
_i = C.c_int
_i64 = C.c_int64
_f = C.c_float

# name -> (restype, argtypes); mirrors include/chebgcn.h one to one
SIGNATURES = {
    'chebgcn_version': (_i, []),
    'chebgcn_last_error': (C.c_char_p, []),
    'chebgcn_last_dispatch': (C.c_char_p, []),
    'chebgcn_plane_stride': (_i, [_i]),
    'chebgcn_graph_create': (_i, [_i, _i64, _p, _p, _p, C.POINTER(_p)]),
    'chebgcn_graph_create_planes': (_i, [_i, _i64, _p, _p, _p, _i, C.POINTER(_p)]),
    'chebgcn_graph_destroy': (None, [_p]),
    'chebgcn_graph_query': (_i, [_p, _i, C.POINTER(_i64)]),
    'chebgcn_recurrence_fwd': (_i, [_p, _p, _p, _i, _i, _i, _p]),
    'chebgcn_recurrence_bwd': (_i, [_p, _p, _p, _i, _i, _i, _p]),
    'chebgcn_recurrence_fwd_t': (_i, [_p, _p, _p, _i, _i, _i, _p]),
    'chebgcn_reindex_weights': (_i, [_p, _p, _i, _i, _i, _p]),
    'chebgcn_reindex_weights_batch': (_i, [_i, _p, _p, _p, _p, _p, _p]),
    'chebgcn_contract_fwd': (_i, [_p, _p, _p, _i, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    'chebgcn_contract_fwd_bf16_workspace': (C.c_size_t, [_i, _i, _i]),
    'chebgcn_contract_fwd_bf16': (_i, [_p, _p, _p, _i, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, C.c_size_t, _p]),
    'chebgcn_contract_bwd_x_bf16_workspace': (C.c_size_t, [_i, _i, _i]),
    'chebgcn_contract_bwd_x_bf16': (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _p, C.c_size_t, _p]),
    'chebgcn_contract_bwd_w_bf16_workspace': (C.c_size_t, [_i, _i, _i, _i, _i]),
    'chebgcn_contract_bwd_w_bf16': (_i, [_p, _p, _p, _p, C.c_size_t, _i, _i, _i, _i, _i, _i, _p]),
    'chebgcn_adam_partials': (_i, [_i64]),
    'chebgcn_adam_step_sq': (_i, [_p, _p, _p, _p, _i64, _f, _p, _f, _f, _f, _f, _f, _p, _p]),
    'chebgcn_adam_step_sq_all': (_i, [_p, _p, _p, _p, _i64, _i64, _f, _p, _f, _f, _f, _f, _f, _p, _p]),
    'chebgcn_loss_bookkeeping': (_i, [_p, _p, _i, _f, _p, _f, _f, _p, _p, _p, _p]),
    'chebgcn_set_scalars': (_i, [_p, C.c_float, C.c_float, _p]),
    'chebgcn_softmax_xent': (_i, [_p, _p, _i, _p, _p, _i, _i, _p]),
    'chebgcn_relu_grad_bf16': (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p, C.c_size_t, _p]),
    'chebgcn_bf16_dy16_supported': (_i, [_i, _i, _i, _i, _i]),
    'chebgcn_contract_bwd_w_bf16_dy16': (_i, [_p, _p, _p, _p, C.c_size_t, _i, _i, _i, _i, _i, _p]),
    'chebgcn_contract_bwd_x_bf16_dy16': (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p, C.c_size_t, _p]),
    'chebgcn_brelu_pool_fwd': (_i, [_p, _p, _i, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    'chebgcn_brelu_pool_bwd_workspace': (C.c_size_t, [_i, _i, _i, _i, _i]),
    'chebgcn_brelu_pool_bwd': (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p, C.c_size_t, _p]),
    'chebgcn_pool_gather_fwd': (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    'chebgcn_pool_scatter_bwd_workspace': (C.c_size_t, [_i, _i, _i, _i, _i]),
    'chebgcn_pool_scatter_bwd': (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p, C.c_size_t, _p]),
    'chebgcn_contract_bwd_w_workspace': (C.c_size_t, [_i, _i, _i, _i, _i]),
    'chebgcn_contract_bwd_w': (_i, [_p, _p, _p, _p, C.c_size_t, _i, _i, _i, _i, _i, _p]),
    'chebgcn_contract_bwd_x': (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p]),
    'chebgcn_contract_bwd_x_relu': (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    'chebgcn_contract_bwd_w_relu': (_i, [_p, _p, _p, _p, _p, C.c_size_t, _i, _i, _i, _i, _i, _p]),
    'chebgcn_contract_bwd_w_relu_bias_merged': (_i, [_i, _i, _i, _i, _i]),
    'chebgcn_contract_bwd_w_relu_bias': (_i, [_p, _p, _p, _p, _p, _p, C.c_size_t, _i, _i, _i, _i, _i, _p]),
    'chebgcn_contract_fwd_mean_supported': (_i, [_i, _i, _i, _i, _i]),
    'chebgcn_contract_fwd_gated_supported': (_i, [_i, _i, _i, _i, _i]),
    'chebgcn_contract_fwd_gated': (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    'chebgcn_contract_fwd_mean': (_i, [_p, _p, _p, _i, _p, _p, _i, _i, _i, _i, _i, _p]),
    'chebgcn_contract_bwd_w_relu_mean': (_i, [_p, _p, _p, _p, _p, C.c_size_t, _i, _i, _i, _i, _i, _p]),
    'chebgcn_contract_bwd_x_relu_mean': (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    'chebgcn_bias_grad_relu_mean': (_i, [_p, _p, _p, _i, _i, _i, _i, _p, C.c_size_t, _p]),
    'chebgcn_relu_grad_mean': (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p, C.c_size_t, _p]),
    'chebgcn_fused_layer_supported': (_i, [_p, _i, _i, _i, _i]),
    'chebgcn_fused_layer_workspace': (C.c_size_t, [_p, _i, _i, _i, _i]),
    'chebgcn_fused_layer_fwd': (_i, [_p, _p, _p, _p, _i, _p, _p, _p, _p, C.c_size_t, _i, _i, _i, _i, _i, _p]),
    'chebgcn_fused_layer_bwd_x': (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    'chebgcn_perm_data': (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p]),
    'chebgcn_to_plane': (_i, [_p, _p, _i, _i, _i, _p]),
    'chebgcn_from_plane': (_i, [_p, _p, _i, _i, _i, _p]),
    'chebgcn_feature_mean_fwd': (_i, [_p, _p, _i, _i, _i, _p]),
    'chebgcn_feature_mean_bwd': (_i, [_p, _p, _i, _i, _i, _p]),
    'chebgcn_fc_fwd_supported': (_i, [_i, _i, _i]),
    'chebgcn_fc_fwd_workspace': (C.c_size_t, [_i, _i, _i]),
    'chebgcn_fc_fwd': (_i, [_p, _i64, _p, _p, _p, _p, C.c_size_t, _i, _i, _i, _i, _p]),
    'chebgcn_fc_bwd': (_i, [_p, _i64, _p, _p, _p, _p, _p, _p, _i64, _i, _i, _i, _p]),
    'chebgcn_adam_step': (_i, [_p, _p, _p, _p, _i64, _f, _f, _f, _f, _f, _f, _p]),
    'chebgcn_adam_step_dev': (_i, [_p, _p, _p, _p, _i64, _p, _f, _f, _f, _f, _f, _p]),
    'chebgcn_metis_one_level_f32': (_i, [_i64, _p, _p, _p, _p, _p, _i64, _p]),
    'chebgcn_metis_one_level_f32p': (_i, [_i64, _p, _p, _p, _p, _p, _i64, _p]),
    'chebgcn_metis_one_level_f64': (_i, [_i64, _p, _p, _p, _p, _p, _i64, _p]),
    'chebgcn_compute_perm_level': (_i, [_p, _i64, _p, _i64, _p]),
    'chebgcn_bank_order': (_i, [_i, _p, _p, _i, _p, _p]),
}

_lib = None


def lib():
    """The loaded library; raises ChebgcnError with build instructions if absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ChebgcnError(
                'libchebgcn.so not found at %s: build it with '
                '`python -c "import __graft_entry__ as g; g.build()"` or '
                '`make -C gcn_fmri_decoding_amd/csrc` (needs hipcc, --offload-arch=gfx950). '
                'There is no CPU fallback.' % LIB_PATH)
        # PyTorch first: libchebgcn.so needs libamdhip64, and the copy PyTorch bundles must be the one
        # in the process -- loaded the other way round, two HIP runtimes coexist and ours sees no device
        import torch  # noqa: F401
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)        # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


# When a list: every checked launching call appends (what, chebgcn_last_dispatch()) -- the kernel templates the
# dispatchers chose for that call.  Tests assert the instantiation they claim to reach with it; None = off.
dispatch_log = None


def check(rc, what):
    if rc != 0:
        msg = lib().chebgcn_last_error()
        raise ChebgcnError('%s failed (%d): %s' % (what, rc, msg.decode() if msg else '?'))
    if dispatch_log is not None:
        dispatch_log.append((what, lib().chebgcn_last_dispatch().decode()))


def last_dispatch():
    """Kernel templates the calling thread's last launching entry point enqueued (chebgcn_last_dispatch)."""
    return lib().chebgcn_last_dispatch().decode()


def plane_stride(M):
    """Padded plane length; pure arithmetic, identical to chebgcn_plane_stride()."""
    return (int(M) + 31) & ~31
