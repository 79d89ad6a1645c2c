"""ctypes binding of libeda_hip.so (C ABI declared in include/eda_hip.h).

There is no fallback: if the HIP library is missing or an entry point fails,
this raises.  (The CPU oracle under oracle/ is test infrastructure and is never
imported from here.)
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# EDA_HIP_LIB selects an experiment build of the same sources (eda_amd.build.build_variant: instrumented or
# differently tuned kernels); the product path is the default
LIB_PATH = os.environ.get("EDA_HIP_LIB") or os.path.join(_HERE, "csrc", "libeda_hip.so")

_i, _f, _p, _sz = ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t
_l, _u = ctypes.c_long, ctypes.c_uint

# name -> (restype, argtypes); mirrors include/eda_hip.h one to one
SIGNATURES = {
    "eda_version": (_i, []),
    "eda_last_error_string": (ctypes.c_char_p, []),
    "eda_set_fma_mode": (_i, [_i]),
    "eda_get_fma_mode": (_i, []),
    "eda_reload_env": (_i, []),
    "eda_set_deterministic": (_i, [_i]),
    "eda_get_deterministic": (_i, []),
    "eda_index_add_rows_ordered_f32": (_i, [_p, _p, _l, _i, _i, _p, _p]),
    "eda_fps_workspace_bytes": (_sz, [_i, _i, _i]),
    "eda_furthest_point_sampling_f32": (_i, [_p, _i, _i, _i, _p, _p, _sz, _p]),
    "eda_fps_prefix_workspace_bytes": (_sz, [_i, _i, _i]),
    "eda_furthest_point_sampling_prefix_f32": (_i, [_p, _i, _i, _i, _p, _p, _sz, _p]),
    "eda_gather_points_f32": (_i, [_p, _p, _i, _i, _i, _i, _p, _p]),
    "eda_gather_points_grad_f32": (_i, [_p, _p, _i, _i, _i, _i, _p, _p]),
    "eda_ball_query_workspace_bytes": (_sz, [_i, _i, _i]),
    "eda_ball_query_f32": (_i, [_p, _p, _i, _i, _i, _f, _i, _p, _p, _sz, _p]),
    "eda_group_points_f32": (_i, [_p, _p, _i, _i, _i, _i, _i, _p, _p]),
    "eda_group_points_grad_f32": (_i, [_p, _p, _i, _i, _i, _i, _i, _p, _p]),
    "eda_three_nn_f32": (_i, [_p, _p, _i, _i, _i, _p, _p, _p]),
    "eda_three_interpolate_f32": (_i, [_p, _p, _p, _i, _i, _i, _i, _p, _p]),
    "eda_three_interpolate_grad_f32": (_i, [_p, _p, _p, _i, _i, _i, _i, _p, _p]),
    "eda_group_concat_cl_f32": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _i, _p, _p]),
    "eda_group_concat_cl_grad_f32": (_i, [_p, _p, _i, _i, _i, _i, _i, _p, _p]),
    "eda_bn_relu_dropout_max_rows": (_l, []),
    "eda_bn_relu_fwd_f32": (_i, [_p, _l, _i, _p, _p, _f, _f, _i, _p, _p, _i, _p, _p, _p, _p, _p, _p,
                                _p, _f, _p, _u, _p]),
    "eda_bn_relu_bwd_f32": (_i, [_p, _p, _p, _l, _i, _i, _p, _p, _p, _p, _p, _i, _p, _p, _p, _p, _f, _p, _u,
                                _p]),
    "eda_add_dropout_ln_fwd_f32": (_i, [_p, _p, _p, _p, _p, _l, _i, _f, _f, _p, _u, _p, _p, _p, _p, _p, _p]),
    "eda_linear_add_dropout_ln_supported": (_i, [_i, _i]),
    "eda_linear_add_dropout_ln_fwd_f32": (_i, [_p, _l, _l, _i, _p, _l, _i, _p, _p, _p, _p, _f, _f, _p, _u, _p, _p, _p, _p,
                                               _p, _p, _p]),
    "eda_linear_add_dropout_ln_workspace_bytes": (_sz, [_l, _i, _i]),
    "eda_linear_add_dropout_ln_fwd_ws_f32": (_i, [_p, _l, _l, _i, _p, _l, _i, _p, _p, _p, _p, _f, _f, _p, _u, _p, _p, _p, _p,
                                                  _p, _p, _p, _sz, _p]),
    "eda_add_dropout_ln_bwd_workspace_bytes": (_sz, [_l, _i]),
    "eda_add_dropout_ln_bwd_blocks": (_i, [_l]),
    "eda_ln_reduce_grouped_f32": (_i, [_p, _i, _i, _p]),
    "eda_add_dropout_ln_bwd_f32": (_i, [_p, _p, _p, _p, _p, _p, _p, _l, _i, _f, _p, _u, _p, _p, _p, _p, _sz,
                                       _p, _p]),
    "eda_mha_fwd_f32": (_i, [_p, _p, _p, _l, _l, _l, _l, _l, _l, _p, _i, _i, _i, _i, _i, _f, _f,
                            _p, _u, _p, _p, _p]),
    "eda_mha_bwd_f32": (_i, [_p, _p, _p, _l, _l, _l, _l, _l, _l, _p, _i, _i, _i, _i, _i, _f, _f,
                            _p, _u, _p, _p, _p, _l, _l, _p, _p, _p, _p, _l, _l, _l, _l, _l, _l, _p, _sz, _p]),
    "eda_mha_bwd_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "eda_set_bn_sync": (_i, [_p, _p, _i]),
    "eda_peer_slab_bytes": (_sz, []),
    "eda_peer_create": (_i, [_p]),
    "eda_peer_connect": (_i, [_i, _i, _p]),
    "eda_peer_alloc_kind": (_i, []),
    "eda_peer_reset": (_i, []),
    "eda_peer_selftest": (_i, [_p, _i]),
    "eda_peer_disconnect": (_i, []),
    "eda_peer_connected": (_i, []),
    "eda_peer_timeouts": (_l, []),
    "eda_peer_allreduce_f64": (_i, [_p, _l, _p]),
    "eda_peer_bn_hook": (_i, [_p, _p, _l, _p]),
    "eda_set_bn_sync_native": (_i, [_i]),
    "eda_fps_set_cu_reserve": (_i, [_i]),
    "eda_fps_set_background": (_i, [_i]),
    "eda_fps_set_policy": (_i, [_i]),
    "eda_add_n_f32": (_i, [_p, _i, _sz, _p, _p]),
    "eda_gemm_set_dma": (_i, [_i]),
    "eda_mha_fwd_hd64_f32": (_i, [_p, _p, _p, _l, _l, _l, _l, _l, _l, _p, _i, _i, _i, _i, _f, _p, _p]),
    "eda_mha_fwd_hd64_drop_f32": (_i, [_p, _p, _p, _l, _l, _l, _l, _l, _l, _p, _i, _i, _i, _i, _f, _f, _p, _u, _p, _p]),
    "eda_dropout_f32": (_i, [_p, _l, _f, _p, _u, _p, _p]),
    "eda_mha_fwd": (_i, [_p, _p, _p, _l, _l, _l, _l, _l, _l, _p, _i, _i, _i, _i, _i, _f, _f,
                        _p, _u, _p, _p, _i, _p]),
    "eda_mha_fwd_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "eda_mha_fwd_ws": (_i, [_p, _p, _p, _l, _l, _l, _l, _l, _l, _p, _i, _i, _i, _i, _i, _f, _f,
                           _p, _u, _p, _p, _i, _p, _sz, _p]),
    "eda_mha_qproj_supported": (_i, [_i, _i, _i]),
    "eda_mha_qproj_fwd": (_i, [_p, _l, _l, _p, _l, _p, _p, _p, _l, _l, _l, _l, _p, _i, _i, _i, _i, _i, _f, _f, _p, _u,
                              _p, _l, _l, _p, _p, _i, _p]),
    "eda_mha_bwd": (_i, [_p, _p, _p, _l, _l, _l, _l, _l, _l, _p, _i, _i, _i, _i, _i, _f, _f,
                        _p, _u, _p, _p, _p, _l, _l, _p, _p, _p, _p, _l, _l, _l, _l, _l, _l, _p, _sz, _i, _p]),
    "eda_mha_bwd_ticket_bytes": (_sz, [_i, _i, _i, _i]),
    "eda_mha_bwd_tk": (_i, [_p, _p, _p, _l, _l, _l, _l, _l, _l, _p, _i, _i, _i, _i, _i, _f, _f,
                           _p, _u, _p, _p, _p, _l, _l, _p, _p, _p, _p, _l, _l, _l, _l, _l, _l, _p, _sz, _p, _sz, _i, _p]),
    "eda_match_cost_f32": (_i, [_p, _p, _p, _p, _l, _p, _p, _i, _i, _i, _i, _i, _f, _f, _f, _p, _p]),
    "eda_match_slots_i64": (_i, [_p, _p, _i, _i, _i, _i, _p, _p]),
    "eda_box_loss_fwd_f32": (_i, [_p, _l, _l, _p, _p, _p, _p, _i, _i, _i, _i, _p, _p, _p, _p, _p]),
    "eda_box_loss_bwd_f32": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _p, _p]),
    "eda_pos_align_chunk": (_i, [_i, _i]),
    "eda_pos_align_fwd_f32": (_i, [_p, _p, _p, _p, _l, _l, _p, _i, _i, _i, _i, _i, _i, _f, _p, _p, _p]),
    "eda_scale_by_scene_f32": (_i, [_p, _p, _p, _i, _l, _l, _i, _p, _p]),
    "eda_center_query_pos_f32": (_i, [_p, _p, _p, _l, _p, _p, _p]),
    "eda_grad_sumsq_workspace_bytes": (_sz, []),
    "eda_grad_sumsq_f32": (_i, [_p, _l, _p, _p, _i, _p, _p, _p, _p, _p, _p]),
    "eda_adamw_flat_f32": (_i, [_p, _p, _l, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _f, _f, _p]),
    "eda_compact_targets": (_i, [_p, _i, _p, _p, _p, _p, _p, _p, _p, _i, _i, _p, _p, _p, _p]),
    "eda_loss_combine_fwd_f32": (_i, [_p, _p, _p, _p, _f, _f, _i, _i, _p, _p, _p, _p]),
    "eda_loss_combine_bwd_f32": (_i, [_p, _p, _f, _f, _p, _p, _p, _i, _p]),
    "eda_sem_align_lds_bytes": (_sz, [_i, _i]),
    "eda_sem_align_supported": (_i, [_i, _i]),
    "eda_sem_align_fwd_f32": (_i, [_p, _p, _p, _l, _l, _p, _p, _i, _i, _i, _i, _i, _f, _p, _p, _p]),
    "eda_seed_objectness_lds_bytes": (_sz, [_i]),
    "eda_seed_objectness_fwd_f32": (_i, [_p, _p, _p, _p, _l, _p, _l, _p, _l, _p, _i, _i, _i, _i, _p, _p, _p]),
    "eda_bf16x3_planes_bytes": (_sz, [_i, _i]),
    "eda_bf16x3_split_f32": (_i, [_p, _l, _i, _i, _p, _p]),
    "eda_linear_frozen_b3_supported": (_i, [_l, _i, _i]),
    "eda_linear_frozen_b3_f32": (_i, [_p, _l, _l, _i, _p, _i, _p, _i, _p, _l, _p]),
    "eda_wgrad_workspace_bytes": (_sz, [_l, _i, _i]),
    "eda_wgrad_f32": (_i, [_p, _l, _p, _l, _l, _i, _i, _p, _p, _p, _sz, _p]),
    "eda_wgrad_grouped_f32": (_i, [_p, _i, _p, _p, _p]),
    "eda_wgrad_set_arith": (_i, [_i]),
    "eda_lsa_f32": (_i, [_p, _l, _l, _l, _i, _i, _i, _p, _p, _p]),
    "eda_wcolsum_workspace_bytes": (_sz, [_l, _i, _i]),
    "eda_wcolsum_f32": (_i, [_p, _l, _i, _l, _p, _l, _i, _p, _p, _p, _sz, _p, _p]),
    "eda_colsum_workspace_bytes": (_sz, [_l, _i]),
    "eda_colsum_f32": (_i, [_p, _l, _i, _l, _p, _p, _sz, _p, _p]),
    "eda_device_copy_f32": (_i, [_p, _p, _sz, _p]),
    "eda_transpose_batch_f32": (_i, [_p, _i, ctypes.c_longlong, _p]),
    "eda_linear_ex_f32": (_i, [_p, _l, _l, _i, _p, _l, _i, _p, _i, _f, _p, _u, _p, _l, _f, _p, _l, _p]),
    "eda_linear_splitk_workspace_bytes": (_sz, [_l, _i, _i]),
    "eda_linear_ex_ws_f32": (_i, [_p, _l, _l, _i, _p, _l, _i, _p, _i, _f, _p, _u, _p, _l, _f, _p, _l, _p, _sz, _p]),
    "eda_linear_dgrad_ws_f32": (_i, [_p, _l, _l, _i, _p, _l, _i, _p, _l, _p, _sz, _p]),
    "eda_linear_addend_ws_f32": (_i, [_p, _l, _l, _i, _p, _l, _i, _p, _p, _l, _p, _l, _p, _sz, _p]),
    "eda_linear_dgrad_addend_ws_f32": (_i, [_p, _l, _l, _i, _p, _l, _i, _p, _l, _p, _l, _p, _sz, _p]),
    "eda_l2norm_rows_fwd_f32": (_i, [_p, _l, _i, _f, _p, _p, _p]),
    "eda_l2norm_rows_bwd_f32": (_i, [_p, _p, _p, _l, _i, _f, _p, _p]),
    "eda_linear_fwd_f32": (_i, [_p, _l, _l, _i, _p, _l, _i, _p, _i, _p, _l, _p]),
    "eda_linear_dgrad_f32": (_i, [_p, _l, _l, _i, _p, _l, _i, _p, _l, _p]),
    "eda_linear_grouped_fwd_f32": (_i, [_i, _p, _p, _l, _p, _p, _p, _p, _p, _i, _p, _p, _p]),
    "eda_linear_grouped_dgrad_f32": (_i, [_i, _p, _p, _l, _p, _p, _p, _p, _p, _p, _p]),
    "eda_bn_relu_grouped_fwd_f32": (_i, [_p, _l, _i, _i, _p, _p, _p, _p, _f, _f, _i, _p, _p, _p, _p, _p, _f, _p, _p, _p]),
    "eda_bn_relu_grouped_bwd_f32": (_i, [_p, _p, _l, _i, _i, _p, _p, _p, _p, _p, _i, _p, _p, _p, _f, _p, _p, _p]),
    "eda_tiny_out_bwd_workspace_bytes": (ctypes.c_size_t, [_i, _i, _i]),
    "eda_tiny_out_bwd_multi_f32": (_i, [_i, _l, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, ctypes.c_size_t, _p]),
    "eda_bn_relu_grouped_bwd_multi_f32": (_i, [_i, _p, _p, _l, _i, _i, _p, _p, _i, _p, _p, _f, _p, _p, _p]),
    "eda_sa_fused_fwd_f32": (_i, [_p, _l, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _i, _l, _i, _p, _p, _p, _p, _p, _p,
                                 _f, _f, _i, _i, _p, _p, _p, _p, _p, _p]),
    "eda_sa_fused_eval_supported": (_i, [_i, _i, _p, _i]),
    "eda_sa_fused_eval_f32": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _i, _i, _p, _p, _p, _p, _p, _p, _f, _p, _p]),
    "eda_sa_fused_bwd_workspace_bytes": (_sz, [_l, _i, _p, _i]),
    "eda_sa_fused_bwd_f32": (_i, [_p, _p, _p, _l, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _i, _l, _i, _p, _p, _p,
                                 _p, _p, _i, _i, _p, _p, _p, _sz, _p, _p, _p, _p, _l, _p, _p]),
    "eda_sa_fused_bwd_wt_f32": (_i, [_p, _p, _p, _l, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _i, _l, _i, _p, _p, _p,
                                    _p, _p, _p, _i, _i, _p, _p, _p, _sz, _p, _p, _p, _p, _l, _p, _p]),
}

_lib = None


class EdaHipError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle of libeda_hip.so."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise EdaHipError(
                f"{LIB_PATH} is missing: build it with `python -m eda_amd.build` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)          # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        if L.eda_version() != 1:
            raise EdaHipError(f"libeda_hip.so ABI version {L.eda_version()} != 1")
        _lib = L
    return _lib


def last_error():
    msg = lib().eda_last_error_string()
    return msg.decode() if msg else ""


def check(rc, what):
    if rc != 0:
        msg = lib().eda_last_error_string()
        raise EdaHipError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")
