"""ctypes binding of libfov360_hip.so (the C ABI declared in include/fov360.h).

There is NO CPU fallback: if the HIP library is missing the import of any compute entry point
fails loudly.  Loading the library and querying its symbols does not need a GPU.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# FOV_LIB_PATH: another build of the SAME library (A/B timing of kernel variants, diagnostic builds) - never a fallback
LIB_PATH = os.environ.get("FOV_LIB_PATH") or os.path.join(_HERE, "lib", "libfov360_hip.so")

ACT_SIGMOID = 0
ACT_HARD_SIGMOID = 1
IMPL_AUTO = 0
IMPL_GENERIC = 1
IMPL_CLUSTER = 2

OK = 0
ERR_INVALID = -1
ERR_UNSUPPORTED = -2
ERR_WORKSPACE = -3
ERR_LAUNCH = -4
ERR_TIMEOUT = -5

_P = ctypes.c_void_p
_I = ctypes.c_int
_SZ = ctypes.c_size_t

# name -> (restype, argtypes); mirrors include/fov360.h one to one
SIGNATURES = {
    "fov_last_error": (ctypes.c_char_p, []),
    "fov_version": (_I, []),
    "fov_cluster_supported": (_I, [_I, _I]),
    "fov_lstm_seq_workspace_bytes": (_SZ, [_I, _I, _I, _I, _I]),
    "fov_lstm_seq_fwd": (_I, [_P] * 9 + [_I] * 6 + [_P, _SZ, _P]),
    "fov_dense_fwd": (_I, [_P] * 4 + [_I] * 4 + [_P]),
    "fov_dense_add_fwd": (_I, [_P] * 4 + [ctypes.c_int64, _P] + [_I] * 4 + [_P]),
    "fov_mix_head_fwd": (_I, [_P] * 5 + [ctypes.c_int64, _P, _P] + [_I] * 3 + [_P]),
    "fov_mix_head_bwd": (_I, [_P] * 9 + [_I] * 3 + [_P]),
    "fov_mix_decoder_workspace_bytes": (_SZ, [_I] * 2),
    "fov_mix_decoder_fwd": (_I, [_P] * 6 + [ctypes.c_int64] * 2 + [_P] * 21 + [_I] * 5 + [_P, _SZ, _P]),
    "fov_mix_decoder_fwd_bf16": (_I, [_P] * 6 + [ctypes.c_int64] * 2 + [_P] * 21 + [_I] * 5 + [_P, _SZ, _P]),
    "fov_lstm_seq_fwd_bf16": (_I, [_P] * 10 + [_I] * 5 + [_P, _SZ, _P]),
    "fov_mix_decoder_bwd_workspace_bytes": (_SZ, [_I] * 2),
    "fov_mix_decoder_bwd": (_I, [_P] * 21 + [_I] * 5 + [_P, _SZ, _P]),
    "fov_mix_decoder_bwd_bf16": (_I, [_P] * 21 + [_I] * 5 + [_P, _SZ, _P]),
    "fov_matmul_workspace_bytes": (_SZ, [_I] * 3),
    "fov_matmul": (_I, [_P] * 3 + [_I] * 3 + [_P, _SZ, _P]),
    "fov_lstm_seq_fwd_zx": (_I, [_P] * 9 + [_I] * 5 + [_P, _SZ, _P]),
    "fov_seq2seq_decode_workspace_bytes": (_SZ, [_I] * 7),
    "fov_seq2seq_decode_fwd": (_I, [_P] * 13 + [_I] * 8 + [_P, _SZ, _P]),
    "fov_seq2seq_decoder_fwd": (_I, [_P] * 11 + [_I] * 6 + [_P, _SZ, _P]),
    "fov_seq2seq_tf_workspace_bytes": (_SZ, [_I] * 7),
    "fov_seq2seq_tf_fwd": (_I, [_P] * 11 + [_I] * 8 + [_P, _SZ, _P]),
    "fov_meanvar_xyz": (_I, [_P, _P, ctypes.c_int64, _I, _P]),
    "fov_lstm_seq_fwd_train": (_I, [_P] * 10 + [_I] * 6 + [_P, _SZ, _P]),
    "fov_lstm_seq_bwd_workspace_bytes": (_SZ, [_I] * 4),
    "fov_lstm_seq_bwd": (_I, [_P] * 17 + [_I] * 6 + [_P, _SZ, _P]),
    "fov_lstm_seq_bwd_bf16": (_I, [_P] * 17 + [_I] * 6 + [_P, _SZ, _P]),
    "fov_dense_bwd_bf16": (_I, [_P] * 6 + [_I] * 4 + [_P, _SZ, _P]),
    "fov_dense_bwd_workspace_bytes": (_SZ, [_I] * 3),
    "fov_mix_head_wgrad_workspace_bytes": (_SZ, [_I] * 5),
    "fov_mix_head_wgrad": (_I, [_P] * 6 + [_I] * 6 + [_P, _SZ, _P]),
    "fov_wgrad_fused_workspace_bytes": (_SZ, [ctypes.c_int64, _I, _I, _I]),
    "fov_wgrad_fused": (_I, [_P, _I, _P, _I, _P, _P, ctypes.c_int64, _I, _I, _I, _I, _P, _SZ, _P]),
    "fov_dense_bwd": (_I, [_P] * 6 + [_I] * 4 + [_P, _SZ, _P]),
    "fov_mse_dense_grad": (_I, [_P] * 4 + [ctypes.c_int64, _I, _P, _SZ, _P]),
    "fov_mse_dense_grad_w": (_I, [_P] * 4 + [ctypes.c_int64, _I, ctypes.c_float, _I, _I, _I, _P, _SZ, _P]),
    "fov_mse_dense_grad_db": (_I, [_P] * 5 + [ctypes.c_int64, _I, _I, ctypes.c_float, _P, _SZ, _P]),
    "fov_scale": (_I, [_P, ctypes.c_int64, ctypes.c_float, _P]),
    "fov_act_bwd": (_I, [_P] * 4 + [ctypes.c_int64, _I, _P]),
    "fov_act_fwd": (_I, [_P, _P, ctypes.c_int64, _I, _P]),
    "fov_tf_head_supported": (_I, [_I] * 4),
    "fov_tf_head_fwd": (_I, [_P] * 13 + [_I] * 4 + [_P]),
    "fov_tf_head_bwd": (_I, [_P] * 20 + [_I] * 5 + [_P]),
    "fov_mlp_head_supported": (_I, [_I, _I, _P]),
    "fov_mlp_head_fwd": (_I, [_P] * 7 + [_I] * 4 + [_P]),
    "fov_mlp_head_bwd_workspace_bytes": (_SZ, [_I, _I, _P]),
    "fov_mlp_head_bwd": (_I, [_P] * 10 + [_I] * 3 + [_P, _SZ, _P]),
    "fov_gmm3d_loss_grad": (_I, [_P, _P, ctypes.c_int64, _P, _P, _I, _I, _I, ctypes.c_float, _I, _P, _SZ, _P]),
    "fov_gmm3d_sample": (_I, [_P, _P, _P, _P, ctypes.c_int64, _I, _I, _I, _P]),
    "fov_gauss_nll_grad": (_I, [_P] * 6 + [_I] * 3 + [ctypes.c_float, _P, _SZ, _P]),
    "fov_rmsprop_tf_step": (_I, [_P] * 3 + [ctypes.c_int64] + [ctypes.c_float] * 4 + [_P]),
    "fov_dense_mse_head_supported": (_I, [ctypes.c_int64, _I, _I]),
    "fov_dense_mse_head_workspace_bytes": (_SZ, [ctypes.c_int64, _I, _I]),
    "fov_dense_mse_head": (_I, [_P] * 9 + [ctypes.c_int64, _I, _I, _I, ctypes.c_float, _P, _SZ, _P]),
    "fov_rmsprop_tf_step_guarded": (_I, [_P] * 3 + [ctypes.c_int64] + [ctypes.c_float] * 4 + [_P] * 5),
    "fov_categorical_crossentropy_grad": (_I, [_P] * 4 + [ctypes.c_int64, _I, _P, _SZ, _P]),
    "fov_xyz_sum1_grad": (_I, [_P] * 3 + [ctypes.c_int64, _I, _P, _SZ, _P]),
    "fov_sample_refeed_fwd": (_I, [_P] * 4 + [ctypes.c_int64] + [_I] * 4 + [_P]),
    "fov_sample_refeed_bwd": (_I, [_P, ctypes.c_int64] + [_P] * 4 + [_I] * 5 + [_P]),
    "fov_adam_step": (_I, [_P] * 4 + [ctypes.c_int64] + [ctypes.c_float] * 4 + [ctypes.c_int64, _P]),
    "fov_rmsprop_step": (_I, [_P] * 3 + [ctypes.c_int64] + [ctypes.c_float] * 3 + [_P]),
    "fov_lstm_stack2_supported_bf16": (_I, [_I] * 4),
    "fov_lstm_stack2_supported": (_I, [_I] * 4),
    "fov_lstm_stack2_fwd": (_I, [_P] * 19 + [_I] * 5 + [_P, _SZ, _P]),
    "fov_lstm_stack2_fwd_bf16": (_I, [_P] * 15 + [_I] * 5 + [_P, ctypes.c_size_t, _P]),
    "fov_mix_decoder_prepack": (_I, [_P, _P, ctypes.c_size_t, _P, ctypes.c_size_t, _I, _P]),
    "fov_guard_flag": (_I, [_P] * 5),
    "fov_reduce_defer_begin": (_I, [_P, ctypes.c_size_t, _P, ctypes.c_size_t, _P]),
    "fov_reduce_defer_flush": (_I, [_P, _P]),
    "fov_reduce_defer_end": (_I, [_P, _P]),
    "fov_adam_step_guarded": (_I, [_P] * 4 + [ctypes.c_int64] + [ctypes.c_float] * 4 + [ctypes.c_int64] + [_P] * 5),
    "fov_rmsprop_step_guarded": (_I, [_P] * 3 + [ctypes.c_int64] + [ctypes.c_float] * 3 + [_P] * 5),
    "fov_conv2d_fwd": (_I, [_P, ctypes.c_int64, ctypes.c_int64, _P, _P, _P, _P] + [_I] * 8 + [_P]),
    "fov_convlstm_cell_fwd": (_I, [_P, ctypes.c_int64, ctypes.c_int64, _I, _P, ctypes.c_int64, ctypes.c_int64] + [_P] * 5 + [ctypes.c_int64, _P] + [_I] * 7 + [_P]),
    "fov_conv2d_fwd2": (_I, [_P, ctypes.c_int64, ctypes.c_int64, _I, _P, ctypes.c_int64, ctypes.c_int64, _I, _P, _P, _P, _P] + [_I] * 7 + [_P]),
    "fov_lstm_stack2_bwd_supported": (_I, [_I] * 4),
    "fov_lstm_stack2_bwd_workspace_bytes": (_SZ, [_I] * 4),
    "fov_lstm_stack2_bwd": (_I, [_P] * 29 + [_I] * 6 + [_P, _SZ, _P]),
    "fov_lstm_seq_wgrad": (_I, [_P] * 7 + [_I] * 6 + [_P, _SZ, _P]),
    "fov_lstm_seq_wgrad_pair_one_launch": (_I, [_I] * 4),
    "fov_lstm_seq_wgrad_pair": (_I, [_P] * 7 + [_I] * 2 + [_P] * 7 + [_I] * 2 + [_I] * 3 + [_P, _SZ, _P]),
    "fov_stream_create": (_I, [_I, ctypes.POINTER(ctypes.c_void_p)]),
    "fov_stream_destroy": (_I, [_P]),
    "fov_conv2d_dilated_fwd": (_I, [_P, ctypes.c_int64, ctypes.c_int64, _P, _P, _P, _P] + [_I] * 9 + [_P]),
    "fov_convlstm_cell_dilated_fwd": (_I, [_P, ctypes.c_int64, ctypes.c_int64, _I, _P, ctypes.c_int64, ctypes.c_int64] + [_P] * 5 + [ctypes.c_int64, _P] + [_I] * 8 + [_P]),
    "fov_conv2d_dilated_wgrad": (_I, [_P, ctypes.c_int64, _P, _P] + [_I] * 9 + [_P, _SZ, _P]),
    "fov_convlstm_gates": (_I, [_P, _P, _P, ctypes.c_int64, ctypes.c_int64, _I, _I, _P]),
    "fov_softmax_lastdim": (_I, [_P, _P, ctypes.c_int64, _I, _P]),
    "fov_convlstm_gates_train": (_I, [_P] * 4 + [ctypes.c_int64, _P, ctypes.c_int64, _I, _I, _P]),
    "fov_convlstm_gates_bwd": (_I, [_P, ctypes.c_int64] + [_P] * 5 + [ctypes.c_int64, _I, _I, _P]),
    "fov_conv2d_wgrad_workspace_bytes": (_SZ, [_I] * 4),
    "fov_conv2d_wgrad": (_I, [_P, ctypes.c_int64, _P, _P] + [_I] * 8 + [_P, _SZ, _P]),
    "fov_conv2d_weight_transpose": (_I, [_P, _P] + [_I] * 4 + [_P]),
    "fov_softmax_lastdim_bwd": (_I, [_P] * 3 + [ctypes.c_int64, _I, _P]),
    "fov_colsum": (_I, [_P, _P, ctypes.c_int64, _I, _I, _P, _SZ, _P]),
    "fov_window_count": (ctypes.c_int64, [_I, _I, _I]),
    "fov_window_stacks": (_I, [_P] * 4 + [_I] * 6 + [_P]),
    "fov_fov_hit_rate": (_I, [_P, ctypes.c_int64, _P, ctypes.c_int64, _P, ctypes.c_int64, ctypes.c_float, ctypes.c_float, _P]),
    "fov_workspace_init": (_I, [_P, _SZ, _P]),
    "fov_check_status": (_I, [_P, _SZ, _P]),
    "fov_workspace_force_safe": (_I, [_P, _SZ, _I, _P]),
    "fov_reload_env": (None, []),
    "fov_debug_generic_launches": (ctypes.c_int64, []),
    "fov_debug_set_epoch": (_I, [_P, _SZ, ctypes.c_uint, _P]),
    "fov_exchange_mode": (_I, [_P, _SZ, _P]),
}

_lib = None


class FovError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libfov360_hip: error %d: %s" % (code, msg))
        self.code = code


def lib():
    """The loaded library (raises if the HIP extension has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "HIP extension missing: %s (build it with `python __graft_entry__.py` or "
                "`make -C longterm360fov_amd/csrc`); there is no CPU fallback" % LIB_PATH)
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(code):
    if code != OK:
        raise FovError(code, lib().fov_last_error().decode("utf-8", "replace"))
