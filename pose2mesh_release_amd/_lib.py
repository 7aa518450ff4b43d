"""ctypes binding of libp2m_hip.so / libp2m_host.so (the C ABI declared in include/p2m.h).

The product path has NO CPU fallback: if the HIP library is missing, or a call returns a
negative status, a P2MError is raised.  torch is used only to own device memory and streams;
nothing but raw pointers, sizes and the stream handle crosses the boundary.
"""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIBDIR = os.path.join(HERE, "lib")


class P2MError(RuntimeError):
    pass


_c = ctypes
_i32, _i64, _f32, _vp = _c.c_int32, _c.c_int64, _c.c_float, _c.c_void_p

# name -> (restype, argtypes); must list every symbol of include/p2m.h (tests check this)
HIP_SYMBOLS = {
    "p2m_last_error_string": (_c.c_char_p, []),
    "p2m_version": (_c.c_char_p, []),
    "p2m_stream_capture_id": (_c.c_int, [_vp, _c.POINTER(_c.c_uint64)]),
    "p2m_graph_create": (_c.c_int, [_vp, _vp, _vp, _i32, _i32, _c.POINTER(_vp)]),
    "p2m_graph_destroy": (_c.c_int, [_vp]),
    "p2m_graph_info": (_c.c_int, [_vp, _c.POINTER(_i32 * 4)]),
    "p2m_cheb_basis_fwd": (_c.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "p2m_cheb_basis_bwd": (_c.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "p2m_cheb_combine_small": (_c.c_int, [_vp, _vp, _i32, _i32, _vp, _vp, _i32, _vp]),
    "p2m_cheb_combine_small_real": (_c.c_int, [_vp, _vp, _i32, _i32, _vp, _vp, _i32, _vp, _i32, _f32, _vp]),
    "p2m_cheb_expand_small": (_c.c_int, [_vp, _vp, _i32, _vp, _i32, _i32, _vp]),
    "p2m_weight_pack": (_c.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "p2m_weight_grad_unpack": (_c.c_int, [_vp, _vp, _i32, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "p2m_graph_split_info": (_c.c_int, [_vp, _c.POINTER(_i32 * 2), _c.POINTER(_f32 * 2)]),
    "p2m_cheb_basis_fwd_real": (_c.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp]),
    "p2m_graph_pair_info": (_c.c_int, [_vp, _c.POINTER(_i32 * 2)]),
    "p2m_graph_plan_info": (_c.c_int, [_vp, _c.POINTER(_i32 * 3)]),
    "p2m_cheb_basis_pair": (_c.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _vp]),
    "p2m_gemm_planes_rows": (_c.c_int, [_vp, _i32, _i32, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _vp, _i32,
                                        _vp, _vp, _vp, _i32, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp]),
    "p2m_rows_tiles_per_sample": (_i32, [_vp, _i32]),
    "p2m_gemm_tn_rows": (_c.c_int, [_vp, _i32, _i32, _vp, _i32, _i32, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp,
                                    _i32, _vp, _vp, _i32, _vp, _vp, _vp]),
    "p2m_weight_eff": (_c.c_int, [_vp, _vp, _i32, _i32, _f32, _f32, _vp]),
    "p2m_weight_grad_unpack2": (_c.c_int, [_vp, _vp, _i32, _vp, _vp, _i32, _f32, _f32, _vp, _vp, _i32, _i32, _i32,
                                           _i32, _vp]),
    "p2m_bn_finalize_rows": (_c.c_int, [_vp, _i32, _i32, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _f32, _f32, _vp,
                                        _vp, _vp, _vp, _i32, _vp]),
    "p2m_bn_finalize_split": (_c.c_int, [_vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _f32, _f32, _vp, _vp, _vp, _vp, _i32,
                                         _vp]),
    "p2m_cheb_tile_gemm_supported": (_i32, [_vp, _i32, _i32, _i32]),
    "p2m_cheb_tile_gemm_mg": (_i32, [_i32, _i32]),
    "p2m_cheb_tile_gemm": (_c.c_int, [_vp, _i32, _vp, _vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp,
                                      _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp]),
    "p2m_cheb_tile_gemm_bnr_slots": (_i32, [_vp, _i32, _i32, _i32, _i32]),
    "p2m_bn_bwd_blocks_fake": (_i32, [_vp, _i32, _i32]),
    "p2m_bn_bwd_reduce_fake": (_c.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _i32, _i32, _vp]),
    "p2m_act_bound": (_c.c_int, [_vp, _vp, _i32, _vp, _vp, _vp]),
    "p2m_graph_plane_bits": (_i32, [_vp, _i32]),
    "p2m_amax": (_c.c_int, [_vp, _i64, _vp, _vp]),
    "p2m_conv_weights_prepare": (_c.c_int, [_vp, _i32, _i32, _vp]),
    "p2m_amax_rows": (_c.c_int, [_vp, _i32, _vp, _i32, _i32, _vp, _vp]),
    "p2m_bn_finalize_tiles": (_c.c_int, [_vp, _i32, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _f32, _f32, _vp, _vp, _vp, _vp,
                                         _i32, _vp]),
    "p2m_graph_fake_ids": (_c.c_int, [_vp, _vp]),
    "p2m_graph_real_ids": (_c.c_int, [_vp, _vp]),
    "p2m_graph_set_classes": (_c.c_int, [_vp, _vp]),
    "p2m_graph_class_info": (_c.c_int, [_vp, _c.POINTER(_i32 * 3)]),
    "p2m_stats_rows_w": (_c.c_int, [_vp, _vp, _i32, _i32, _vp, _vp]),
    "p2m_class_reduce": (_c.c_int, [_vp, _vp, _vp, _i32, _i32, _vp]),
    "p2m_gemm_planes": (_c.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _vp,
                                   _i32, _i32, _i32, _i64, _vp, _vp, _vp, _i32, _vp, _vp]),
    "p2m_weight_split_elems": (_i64, [_i32, _i32, _i32]),
    "p2m_weight_split": (_c.c_int, [_vp, _i32, _i32, _i32, _vp, _i32, _vp, _vp]),
    "p2m_stats_tile_rows": (_i32, []),
    "p2m_gemm_tn": (_c.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _i32, _i32, _i64, _i64, _vp, _vp,
                               _i32, _vp, _i32, _vp, _i32, _vp]),
    "p2m_bn_finalize": (_c.c_int, [_vp, _i32, _i64, _vp, _vp, _vp, _vp, _f32, _f32, _vp, _vp, _vp, _vp,
                                   _i32, _i32, _vp]),
    "p2m_bn_eval_coeffs": (_c.c_int, [_vp, _vp, _vp, _vp, _f32, _vp, _vp, _vp, _vp, _i32, _vp]),
    "p2m_bn_act_fwd": (_c.c_int, [_vp, _vp, _vp, _i32, _vp, _i32, _i32, _vp, _i64, _i32, _vp, _i32, _vp, _vp]),
    "p2m_bn_bwd_blocks": (_i32, [_i64, _i32]),
    "p2m_bn_bwd_blocks_classes": (_i32, [_vp, _i64, _i32]),
    "p2m_bn_bwd_reduce": (_c.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _i64, _i32, _vp, _vp]),
    "p2m_bn_bwd_finalize": (_c.c_int, [_vp, _i32, _i64, _vp, _vp, _vp, _i32, _i32, _vp]),
    "p2m_bn_bwd_apply": (_c.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _i64, _i32, _vp,
                                    _i32, _vp, _vp]),
    "p2m_pair_sum": (_c.c_int, [_vp, _vp, _i64, _i32, _vp, _vp]),
    "p2m_lerp_bwd_add": (_c.c_int, [_vp, _vp, _i64, _i32, _i32, _vp]),
    "p2m_coord_loss": (_c.c_int, [_vp, _vp, _vp, _i32, _i64, _f32, _vp, _vp, _vp]),
    "p2m_mesh_loss_workspace": (_i64, [_i32, _i32, _i32, _i32]),
    "p2m_mesh_loss": (_c.c_int, [_vp, _i32, _vp, _i32, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                 _i32, _vp, _vp, _f32, _f32, _f32, _f32, _vp, _vp, _vp, _i32, _vp]),
    "p2m_mesh_epilogue": (_c.c_int, [_vp, _i32, _vp, _i32, _f32, _vp, _vp, _vp, _i32, _vp, _vp, _i32, _vp]),
    "p2m_adam_step": (_c.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, _f32, _f32, _f32, _f32, _f32, _vp]),
    "p2m_rmsprop_step": (_c.c_int, [_vp, _vp, _vp, _i64, _f32, _f32, _f32, _f32, _vp]),
    "p2m_adam_step_dev": (_c.c_int, [_vp, _vp, _vp, _vp, _i64, _vp, _f32, _f32, _f32, _vp]),
    "p2m_rmsprop_step_dev": (_c.c_int, [_vp, _vp, _vp, _i64, _vp, _f32, _f32, _vp]),
    "p2m_gemm_tn_acc": (_c.c_int, [_vp, _i32, _vp, _i32, _i64, _vp, _i32, _vp, _vp, _vp]),
    "p2m_pn_stage_fwd": (_c.c_int, [_vp, _i32, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _f32, _f32, _vp, _f32, _vp,
                                    _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "p2m_pn_stage_bwd": (_c.c_int, [_vp, _i32, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _f32, _vp, _vp, _vp, _vp,
                                    _vp, _i32, _vp, _i32, _i32, _i32, _vp]),
    "p2m_chebconv_fwd": (_c.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
}

HOST_SYMBOLS = {
    "p2m_hem_match": (_i64, [_vp, _vp, _vp, _i64, _vp, _i64, _vp, _vp]),
    "p2m_tree_order_level": (_i64, [_vp, _i64, _vp, _i64, _vp]),
    "p2m_host_version": (_c.c_char_p, []),
}

_hip = None
_host = None


def _load(path, symbols):
    if not os.path.exists(path):
        raise P2MError(
            f"{os.path.basename(path)} not found at {path}: build it with "
            f"`python -m pose2mesh_release_amd.build` (there is no CPU/PyTorch fallback for this path)")
    lib = ctypes.CDLL(path)
    for name, (res, args) in symbols.items():
        fn = getattr(lib, name)  # AttributeError if the ABI is incomplete
        fn.restype = res
        fn.argtypes = args
    return lib


def hip():
    """The HIP library (loaded on first use)."""
    global _hip
    if _hip is None:
        # torch ships its own libamdhip64: import it FIRST so that this library binds to the same HIP
        # runtime instance (two runtimes in one process do not see each other's device memory).
        import torch  # noqa: F401
        _hip = _load(os.environ.get("P2M_HIP_LIB") or os.path.join(LIBDIR, "libp2m_hip.so"), HIP_SYMBOLS)
    return _hip


def host():
    """The CPU-only helper library."""
    global _host
    if _host is None:
        _host = _load(os.path.join(LIBDIR, "libp2m_host.so"), HOST_SYMBOLS)
    return _host


def check(rc, what=""):
    if rc != 0:
        msg = hip().p2m_last_error_string()
        raise P2MError(f"{what} failed (status {rc}): {msg.decode() if msg else ''}")
