"""ctypes binding of libaurppo_hip.so (declarations mirror include/aurppo.h one to one)."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libaurppo_hip.so")

# every symbol include/aurppo.h declares (tests/test_abi.py checks the header against this list)
SYMBOLS = (
    "aurppo_version", "aurppo_last_error", "aurppo_device_count", "aurppo_k7_variant", "aurppo_reload_knobs", "aurppo_k7w_kernel", "aurppo_gae_f32", "aurppo_gae_pack_f32",
    "aurppo_mt19937_create", "aurppo_mt19937_destroy", "aurppo_mt19937_seed", "aurppo_mt19937_get_state",
    "aurppo_mt19937_set_state", "aurppo_mt19937_status_f32", "aurppo_arange_i32", "aurppo_shuffle_i32", "aurppo_shuffle_epochs_i32",
    "aurppo_gather_f32", "aurppo_loss_workspace_bytes", "aurppo_loss_fwd_bwd_f32", "aurppo_loss_fwd_bwd_packed_f32",
    "aurppo_clip_workspace_bytes", "aurppo_grad_norm_clip_f32", "aurppo_mlp_workspace_bytes", "aurppo_mlp_ppo_step_f32",
    "aurppo_mlp_ppo_step_ev_f32", "aurppo_mlp_ppo_minibatch_f32", "aurppo_mlp_ppo_grad_f32", "aurppo_mlp_ppo_apply_f32", "aurppo_mlp_ppo_apply_parts_f32", "aurppo_p2p_handle_bytes", "aurppo_p2p_parts", "aurppo_p2p_create",
    "aurppo_p2p_get_handle", "aurppo_p2p_open_peers", "aurppo_p2p_allreduce_mean_f32", "aurppo_p2p_status", "aurppo_p2p_destroy", "aurppo_pack_records_f32", "aurppo_mlp_act_f32", "aurppo_clip_adam_f32",
    "aurppo_bias_relu_pool2_fwd_f32", "aurppo_bias_relu_pool2_bwd_f32", "aurppo_weighted_batch_sum_f32",
    "aurppo_first_block_fwd_f32", "aurppo_first_block_bwd_f32", "aurppo_conv3x3_wop_bytes", "aurppo_conv3x3_f32", "aurppo_linear_f32", "aurppo_linear_bias_act_f32",
    "aurppo_linear_wgrad_ws_bytes", "aurppo_linear_wgrad_f32", "aurppo_conv3x3_wgrad_ws_bytes", "aurppo_conv3x3_wgrad_f32",
    "aurppo_mlp_wide_workspace_bytes", "aurppo_mlp_wide_ppo_step_f32", "aurppo_mlp_wide_ppo_minibatch_f32",
    "aurppo_mlp_wide_act_f32",
)

_lib = None


class AurppoLibraryMissing(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load the HIP library or fail loudly -- there is no fallback implementation."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AurppoLibraryMissing(
            f"{LIB_PATH} not found. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). aur_ppo_amd has no CPU or PyTorch fallback for its kernels.")
    # torch ships its own libamdhip64 (same SONAME as /opt/rocm's).  Import torch FIRST so this
    # library binds to the runtime torch uses: streams and device pointers are only meaningful
    # inside one HIP runtime instance.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    vp, i32, f64 = C.c_void_p, C.c_int, C.c_double
    lib.aurppo_version.restype = i32
    lib.aurppo_last_error.restype = C.c_char_p
    lib.aurppo_device_count.restype = i32
    lib.aurppo_gae_f32.argtypes = [vp] * 7 + [i32, i32, f64, f64, i32, vp]
    lib.aurppo_gae_pack_f32.argtypes = [vp] * 9 + [i32, i32, f64, f64, i32, vp]
    lib.aurppo_mt19937_create.argtypes = [C.POINTER(vp), C.c_uint32, i32, vp]
    lib.aurppo_mt19937_destroy.argtypes = [vp]
    lib.aurppo_mt19937_seed.argtypes = [vp, C.c_uint32, vp]
    lib.aurppo_mt19937_get_state.argtypes = [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_int32), vp]
    lib.aurppo_mt19937_set_state.argtypes = [vp, C.POINTER(C.c_uint32), C.c_int32, vp]
    lib.aurppo_mt19937_status_f32.argtypes = [vp, vp, vp]
    lib.aurppo_arange_i32.argtypes = [vp, i32, vp]
    lib.aurppo_shuffle_i32.argtypes = [vp, vp, i32, vp]
    lib.aurppo_shuffle_epochs_i32.argtypes = [vp, vp, i32, i32, vp]
    lib.aurppo_gather_f32.argtypes = [vp, i32, C.POINTER(vp), C.POINTER(vp), C.POINTER(i32), i32, vp]
    lib.aurppo_loss_workspace_bytes.argtypes = [i32]
    lib.aurppo_loss_workspace_bytes.restype = C.c_size_t
    lib.aurppo_loss_fwd_bwd_f32.argtypes = [vp] * 7 + [i32, f64, f64, f64, i32, i32] + [vp] * 6
    lib.aurppo_loss_fwd_bwd_packed_f32.argtypes = [vp] * 4 + [i32, f64, f64, f64, i32, i32] + [vp] * 6
    lib.aurppo_clip_adam_f32.argtypes = [vp] * 4 + [C.c_int64, C.c_int64, f64, vp, vp, f64, f64, f64, vp, vp, vp]
    lib.aurppo_mlp_act_f32.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp, C.POINTER(i32), i32, vp, vp, vp, vp]
    lib.aurppo_mlp_wide_workspace_bytes.argtypes = [i32, i32, i32]
    lib.aurppo_mlp_wide_workspace_bytes.restype = C.c_size_t
    lib.aurppo_mlp_wide_ppo_step_f32.argtypes = ([vp] * 4 + [i32] * 6 + [vp, C.POINTER(i32), i32, vp, f64, f64, f64, i32, i32] +
                                                 [vp] * 5)
    lib.aurppo_mlp_wide_ppo_minibatch_f32.argtypes = ([vp] * 4 + [i32] * 6 + [vp, C.POINTER(i32), i32, vp, f64, f64, f64, i32, i32] +
                                                      [vp, vp, vp, f64, vp, vp, f64, f64, f64, vp, vp, i32, i32, vp, vp])
    lib.aurppo_mlp_wide_act_f32.argtypes = [vp, vp] + [i32] * 6 + [vp, C.POINTER(i32), i32] + [vp] * 5
    lib.aurppo_mlp_workspace_bytes.argtypes = [i32]
    lib.aurppo_mlp_workspace_bytes.restype = C.c_size_t
    lib.aurppo_mlp_ppo_step_f32.argtypes = [vp] * 4 + [i32] * 5 + [vp, C.POINTER(i32), i32, vp, f64, f64, f64, i32, i32, vp, vp, vp]
    lib.aurppo_mlp_ppo_step_ev_f32.argtypes = lib.aurppo_mlp_ppo_step_f32.argtypes + [vp, vp]
    lib.aurppo_mlp_ppo_grad_f32.argtypes = [vp] * 4 + [i32] * 5 + [vp, C.POINTER(i32), i32, vp, f64, f64, f64, i32, i32, vp, vp, i32, vp, vp]
    lib.aurppo_mlp_ppo_apply_f32.argtypes = [vp] * 4 + [C.POINTER(i32), i32, i32, f64, f64, vp, vp, f64, f64, f64, vp, vp, i32, vp, i32, vp, vp]
    lib.aurppo_mlp_ppo_apply_parts_f32.argtypes = [vp] * 4 + [C.POINTER(i32), i32, i32, vp, i32, f64, vp, vp, f64, f64, f64, vp, vp, i32, vp, i32, vp, vp]
    lib.aurppo_k7w_kernel.argtypes = [i32, i32]
    lib.aurppo_p2p_parts.argtypes = [i32]
    lib.aurppo_p2p_create.argtypes = [C.POINTER(vp), i32, i32, i32, vp]
    lib.aurppo_p2p_get_handle.argtypes = [vp, vp]
    lib.aurppo_p2p_open_peers.argtypes = [vp, vp]
    lib.aurppo_p2p_allreduce_mean_f32.argtypes = [vp, vp, i32, vp, vp, f64, vp]
    lib.aurppo_p2p_status.argtypes = [vp, C.POINTER(i32), vp]
    lib.aurppo_p2p_destroy.argtypes = [vp]
    lib.aurppo_pack_records_f32.argtypes = [vp, vp, i32, i32, vp, vp]
    lib.aurppo_mlp_ppo_minibatch_f32.argtypes = ([vp] * 4 + [i32] * 5 + [vp, C.POINTER(i32), i32, vp, f64, f64, f64, i32, i32, vp] +
                                                 [vp, vp, f64, vp, vp, f64, f64, f64, vp, vp, i32, i32, vp, vp])
    lib.aurppo_bias_relu_pool2_fwd_f32.argtypes = [vp] * 6 + [i32] * 4 + [vp]
    lib.aurppo_bias_relu_pool2_bwd_f32.argtypes = [vp] * 4 + [i32] * 4 + [vp]
    lib.aurppo_weighted_batch_sum_f32.argtypes = [vp, vp, vp, i32, C.c_int64, vp]
    lib.aurppo_first_block_fwd_f32.argtypes = [vp] * 6 + [i32] * 5 + [vp]
    lib.aurppo_first_block_bwd_f32.argtypes = [vp] * 6 + [i32] * 5 + [vp]
    lib.aurppo_conv3x3_wop_bytes.argtypes = [i32, i32]
    lib.aurppo_conv3x3_wop_bytes.restype = C.c_size_t
    lib.aurppo_conv3x3_f32.argtypes = [vp, vp, vp] + [i32] * 7 + [vp, vp]
    lib.aurppo_linear_f32.argtypes = [vp, vp, vp, C.c_longlong, i32, i32, i32, vp, vp]
    lib.aurppo_linear_bias_act_f32.argtypes = [vp, vp, vp, vp, C.c_longlong, i32, i32, i32, vp, vp]
    lib.aurppo_linear_wgrad_ws_bytes.argtypes = [C.c_longlong, i32, i32]
    lib.aurppo_linear_wgrad_ws_bytes.restype = C.c_size_t
    lib.aurppo_linear_wgrad_f32.argtypes = [vp, vp, vp, C.c_longlong, i32, i32, vp, vp]
    lib.aurppo_conv3x3_wgrad_ws_bytes.argtypes = [i32] * 6
    lib.aurppo_conv3x3_wgrad_ws_bytes.restype = C.c_size_t
    lib.aurppo_conv3x3_wgrad_f32.argtypes = [vp, vp, vp] + [i32] * 6 + [vp, vp]
    lib.aurppo_clip_workspace_bytes.argtypes = [C.c_int64]
    lib.aurppo_clip_workspace_bytes.restype = C.c_size_t
    lib.aurppo_grad_norm_clip_f32.argtypes = [vp, C.c_int64, f64, vp, vp, vp]
    for name in SYMBOLS:
        fn = getattr(lib, name)
        if name not in ("aurppo_last_error", "aurppo_loss_workspace_bytes", "aurppo_clip_workspace_bytes",
                        "aurppo_mlp_workspace_bytes", "aurppo_conv3x3_wop_bytes", "aurppo_linear_wgrad_ws_bytes",
                        "aurppo_conv3x3_wgrad_ws_bytes"):
            fn.restype = i32
    _lib = lib
    return lib
