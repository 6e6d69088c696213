"""ctypes binding of libtimediff_hip.so (include/timediff_hip.h).

The library is built in-tree by `make -C timediff_amd/csrc` (or __graft_entry__.build()).  There is
no CPU or eager-PyTorch fallback: if the shared object is missing, loading fails loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TIMEDIFF_HIP_LIB") or os.path.join(_HERE, "libtimediff_hip.so")

TD_MAX_CAT = 32
TD_MISC_COUNT = 10
TAB_ROWS = ["sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod",
            "sqrt_recipm1_alphas_cumprod", "posterior_mean_coef1", "posterior_mean_coef2",
            "posterior_log_variance_clipped", "log_alpha", "log_1_min_alpha", "log_cumprod_alpha",
            "log_1_min_cumprod_alpha"]
MISC_KEYS = ["time_emb.1.weight", "time_emb.1.bias", "time_emb.3.weight", "time_emb.3.bias",
             "rnn.time_mlp.1.weight", "rnn.time_mlp.1.bias", "rnn.norm.weight", "rnn.norm.bias",
             "fc.weight", "fc.bias"]


LAYER_KERNEL_AUTO, LAYER_KERNEL_BATCH, LAYER_KERNEL_UNITS, LAYER_KERNEL_CLUSTER = 0, 1, 2, 3     # td_set_layer_kernel
PROFILE_LAYER_INFER, PROFILE_LAYER_SAVE, PROFILE_BPTT_SWEEP = 1, 2, 4     # td_profile_enable
HEAD_FUSION_AUTO, HEAD_FUSION_OFF, HEAD_FUSION_ON = 0, 1, 2             # td_set_head_fusion
STEP_LAUNCH_AUTO, STEP_LAUNCH_OFF, STEP_LAUNCH_ON = 0, 1, 2             # td_set_step_launch


class RnnConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("c_in", "hidden", "c_out", "layers", "dirs", "gru", "embed_dim", "time_dim")]


class DiffusionConfig(C.Structure):
    _fields_ = [("n_num", C.c_int32), ("n_cat", C.c_int32), ("seq_len", C.c_int32), ("timesteps", C.c_int32),
                ("num_classes", C.c_int32 * TD_MAX_CAT), ("num_index", C.c_int32 * (2 * TD_MAX_CAT)),
                ("cat_index", C.c_int32 * TD_MAX_CAT), ("data_channels", C.c_int32), ("raw_range", C.c_int32),
                ("tables", C.c_void_p)]


class Rng(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("patient0", C.c_uint32), ("tag", C.c_uint32), ("seed_dev", C.c_void_p)]


class StepRecord(C.Structure):
    """td_step_record: what changes from one optimizer step to the next (read on the device by a captured training step)."""
    _fields_ = [("seed", C.c_uint64), ("step_size", C.c_float), ("inv_sqrt_bc2", C.c_float), ("ema_decay", C.c_float),
                ("ema_mode", C.c_int32), ("reserved", C.c_uint64)]


class TimediffError(RuntimeError):
    pass


_lib = None

_P = C.c_void_p
_I = C.c_int32
# td_layer_done_fn (include/timediff_hip.h): host callback of td_rnn_backward_cb
LAYER_DONE_FN = C.CFUNCTYPE(None, C.c_int32, C.c_void_p)

_SIGS = {
    "td_version": (C.c_int, []),
    "td_last_error": (C.c_char_p, []),
    "td_rnn_packed_bytes": (C.c_size_t, [C.POINTER(RnnConfig)]),
    "td_rnn_workspace_bytes": (C.c_size_t, [C.POINTER(RnnConfig), _I, _I]),
    "td_rnn_pack_weights": (C.c_int, [C.POINTER(RnnConfig), C.POINTER(_P), _P, _P]),
    "td_rnn_time_film": (C.c_int, [C.POINTER(RnnConfig), C.POINTER(_P), _P, _I, _P, _P, _P]),
    "td_rnn_time_film_backward_workspace_bytes": (C.c_size_t, [C.POINTER(RnnConfig), _I]),
    "td_rnn_time_film_backward": (C.c_int, [C.POINTER(RnnConfig), C.POINTER(_P), _P, _I, _P, C.POINTER(_P), _P, _P]),
    "td_rnn_forward": (C.c_int, [C.POINTER(RnnConfig), _P, C.POINTER(_P), _P, _P, _I, _P, _I, _I, _P, _P, _P]),
    "td_rnn_read_hidden": (C.c_int, [C.POINTER(RnnConfig), _P, _I, _I, _P, _P]),
    "td_rnn_train_workspace_bytes": (C.c_size_t, [C.POINTER(RnnConfig), _I, _I]),
    "td_rnn_set_dropout": (C.c_int, [C.c_float, C.c_uint64]),
    "td_rnn_dropout_mask": (C.c_int, [C.POINTER(RnnConfig), _I, _I, _I, _P, _P]),
    "td_rnn_packT_bytes": (C.c_size_t, [C.POINTER(RnnConfig)]),
    "td_rnn_pack_weights_T": (C.c_int, [C.POINTER(RnnConfig), C.POINTER(_P), _P, _P]),
    "td_rnn_forward_train": (C.c_int, [C.POINTER(RnnConfig), _P, C.POINTER(_P), _P, _P, _I, _I, _P, _P, _P]),
    "td_rnn_backward": (C.c_int, [C.POINTER(RnnConfig), _P, C.POINTER(_P), _P, _P, _I, _I, _P, C.POINTER(_P),
                                  C.POINTER(_P), _P, _P]),
    "td_rnn_backward_cb": (C.c_int, [C.POINTER(RnnConfig), _P, C.POINTER(_P), _P, _P, _I, _I, _P, C.POINTER(_P),
                                     C.POINTER(_P), _P, LAYER_DONE_FN, _P, _P]),
    "td_rnn_backward_all": (C.c_int, [C.POINTER(RnnConfig), _P, C.POINTER(_P), _P, _P, _P, _I, _I, _P, _P, C.POINTER(_P),
                                      C.POINTER(_P), _P, LAYER_DONE_FN, _P, _P]),
    "td_q_sample_mixed": (C.c_int, [C.POINTER(DiffusionConfig), _I, _P, _P, _P, _P, Rng, _P, _P, _P, _P, _P]),
    "td_mixed_loss_workspace_bytes": (C.c_size_t, [C.POINTER(DiffusionConfig), _I]),
    "td_mixed_loss": (C.c_int, [C.POINTER(DiffusionConfig), _I, _P, _P, _P, _P, _P, C.c_float, C.c_float, _P, _P, _P, _P]),
    "td_mixed_loss_noalloc": (C.c_int, [C.POINTER(DiffusionConfig), _I, _P, _P, _P, _P, _P, C.c_float, C.c_float, _P, _P, _P]),
    "td_sample_prior": (C.c_int, [C.POINTER(DiffusionConfig), _I, _P, _P, Rng, _P, _P, _P]),
    "td_build_denoiser_input": (C.c_int, [C.POINTER(DiffusionConfig), _I, _P, _P, _P, _P]),
    "td_reverse_step_mixed": (C.c_int, [C.POINTER(DiffusionConfig), _I, _P, _P, _P, _I, _P, _P, _P, Rng, _P]),
    "td_reverse_step_terms": (C.c_int, [C.POINTER(DiffusionConfig), _I, _P, _P, _P, _I, _P, _P, _P, _P]),
    "td_decode_samples": (C.c_int, [C.POINTER(DiffusionConfig), _I, _P, _P, _P, _P]),
    "td_sample_chain_workspace_bytes": (C.c_size_t, [C.POINTER(RnnConfig), C.POINTER(DiffusionConfig), _I]),
    "td_sample_chain": (C.c_int, [C.POINTER(RnnConfig), C.POINTER(DiffusionConfig), _P, C.POINTER(_P), _P, _I, _P, _P,
                                  Rng, _I, _P, _P, _P]),
    "td_sample_chain_release": (C.c_int, []),
    "td_grad_norm_workspace_bytes": (C.c_size_t, []),
    "td_grad_norm": (C.c_int, [_P, C.c_size_t, _P, _P, _P]),
    "td_clip_adam_ema": (C.c_int, [_P, _P, _P, _P, _P, C.c_size_t, _P, C.c_float, C.c_float, C.c_float, C.c_float,
                                   C.c_float, _I, _I, C.c_float, _P]),
    "td_step_record_set": (C.c_int, [_P, C.c_uint64, C.c_float, C.c_float, C.c_float, _I, _I, C.c_float, _P]),
    "td_clip_adam_ema_rec": (C.c_int, [_P, _P, _P, _P, _P, C.c_size_t, _P, C.c_float, C.c_float, C.c_float, C.c_float, _P, _P]),
    "td_set_layer_kernel": (C.c_int, [_I]),
    "td_rnn_layer_kernel_choice": (C.c_int, [C.POINTER(RnnConfig), _I, _I]),
    "td_rnn_step_launch_choice": (C.c_int, [C.POINTER(RnnConfig), _I, _I]),
    "td_cls_workspace_bytes": (C.c_size_t, [C.POINTER(RnnConfig), _I, _I]),
    "td_cls_forward": (C.c_int, [C.POINTER(RnnConfig), _P, _P, _P, _P, _I, _I, _P, _P, _P]),
    "td_cls_backward": (C.c_int, [C.POINTER(RnnConfig), _P, _P, _P, _P, _P, _I, _I, _P, C.POINTER(_P), _P, _P, _P]),
    "td_set_head_fusion": (C.c_int, [_I]),
    "td_set_side_gemm": (C.c_int, [_I]),
    "td_set_step_launch": (C.c_int, [_I]),
    "td_check_async_errors": (C.c_int, []),
    "td_profile_enable": (C.c_int, [_I]),
    "td_profile_read": (C.c_int, [C.POINTER(C.c_int32), C.POINTER(C.c_float)]),
}


def lib():
    """The loaded shared library (raises if it has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise TimediffError(
                f"{LIB_PATH} is missing: build it with `make -C timediff_amd/csrc` (there is no fallback path)")
        # torch's wheel carries its own libamdhip64: it has to be in the process first, so that the
        # library binds to the same HIP runtime that owns the tensors' memory and streams
        import torch  # noqa: F401
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
        # measurement switches of the tools (tools/perf_probe.py): applied once at load through the explicit knobs
        mode = os.environ.get("TIMEDIFF_LAYER_KERNEL", "")
        if mode[:1] in ("b", "u", "c"):
            handle.td_set_layer_kernel({"b": LAYER_KERNEL_BATCH, "u": LAYER_KERNEL_UNITS, "c": LAYER_KERNEL_CLUSTER}[mode[0]])
        if os.environ.get("TIMEDIFF_SIDE_GEMM", "") == "0":
            handle.td_set_side_gemm(0)
    return _lib


def check(rc):
    if rc != 0:
        raise TimediffError(lib().td_last_error().decode())


def check_async():
    """After results have been waited for (a loss read back, samples copied to the host): did a cluster-persistent kernel
    give up on a hand-off (GPU shared or partitioned: its workgroups were not co-resident)?  Raises instead of handing on NaNs."""
    check(lib().td_check_async_errors())


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def ptr_table(tensors):
    return (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
