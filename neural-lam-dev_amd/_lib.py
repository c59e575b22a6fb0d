"""ctypes binding of libnlam_hip.so (C ABI: include/nlam_hip.h).

The library is the product: there is no CPU or eager-PyTorch fallback.  If the
shared object is missing or does not export a symbol, importing this module
raises with the build command to run.
"""
import ctypes
import os

import torch  # noqa: F401  (loads torch's libamdhip64.so.7 first, so that the
# library binds to the same HIP runtime as the streams/tensors it is handed)

_HERE = os.path.dirname(os.path.abspath(__file__))
# (NLAM_LIB_PATH: another build of the same library, for same-box A/B timing of kernel variants)
LIB_PATH = os.environ.get("NLAM_LIB_PATH") or os.path.join(_HERE, "libnlam_hip.so")

_i64, _i32, _p, _f = ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_float

# name -> argtypes  (restype is int unless listed in _RESTYPES)
SIGNATURES = {
    "nlam_last_error": [],
    "nlam_abi_version": [],
    "nlam_mfma_mode": [],
    "nlam_set_mfma_mode": [_i32],
    "nlam_edge_bwd_forms_batch_sum": [_i64, _i64, _i32],
    "nlam_graph_build_host": [_p, _p, _i64, _i64, _i64, _p, _p, _p, _p, _p, _p, _p, _p],
    "nlam_gemm": [_i64, _i64, _i64, _p, _i64, _i64, _p, _i64, _i64, _p, _p, _i64, _i32, _i32, _p, _p],
    "nlam_silu_fwd": [_p, _p, _i64, _p],
    "nlam_silu_bwd": [_p, _p, _p, _i64, _p],
    "nlam_layernorm_fwd": [_p, _i64, _p, _p, _p, _i64, _p, _i64, _i64, _i64, _p],
    "nlam_layernorm_bwd_blocks": [_i64],
    "nlam_layernorm_bwd": [_p, _i64, _p, _p, _i64, _p, _i64, _p, _p, _i32, _p, _i64, _i64, _p],
    "nlam_colsum_blocks": [_i64],
    "nlam_colsum": [_p, _i64, _p, _i32, _p, _i64, _i64, _p],
    "nlam_gather_rows": [_p, _i64, _i64, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _p],
    "nlam_segment_sum": [_p, _i64, _i64, _p, _p, _p, _p, _i64, _i64, _i32, _i64, _i64, _i64, _p],
    "nlam_segment_sum_bsum": [_p, _i64, _i64, _p, _p, _p, _i64, _i64, _p, _i64, _i64, _i64, _i64, _p],
    "nlam_add_rows": [_p, _i64, _p, _i64, _p, _i64, _i64, _i64, _p],
    "nlam_copy_rows": [_p, _i64, _i64, _p, _i64, _i64, _i64, _i64, _i64, _p],
    "nlam_sum_batch": [_p, _i64, _p, _i64, _i64, _p],
    "nlam_mlp_fwd": [_p, _i64, _i64, _i32, _p, _i64, _i64, _i32, _p, _i64, _p, _p, _i64, _p, _p, _p,
                     _p, _i64, _i64, _p, _i64, _i64, _i64, _i64, _i32, _i32, _p],
    "nlam_bwd_grid": [_i64],
    "nlam_mlp_bwd_slab_stride": [_i32, _i32, _i32],
    "nlam_mlp_bwd": [_p, _i64, _i64, _i32, _p, _i64, _i64, _i32, _p, _i64, _p, _p, _i64, _p, _p,
                     _p, _i64, _i64, _p, _i64, _i64, _p, _i64, _i64, _i32, _p, _i64, _p,
                     _i64, _i64, _i32, _i32, _p],
    "nlam_outer_bwd_slab_stride": [_i32, _i32],
    "nlam_outer_bwd": [_p, _i64, _i64, _i32, _p, _i64, _i64, _i32, _p, _i64, _i64, _i32, _p, _p,
                       _i64, _i64, _i64, _p],
    "nlam_inet_supported": [_p],
    "nlam_inet_fwd": [_p, _p],
    "nlam_inet_bwd_workspace": [_p],
    "nlam_inet_bwd": [_p, _p, _p, _i64, _p],
    "nlam_mlp_multi_supported": [],
    "nlam_mlp_fwd_multi": [_i32] + [_p] * 17 + [_i32, _i32, _p],
    "nlam_mlp_bwd_multi": [_i32] + [_p] * 21 + [_i32, _i32, _p],
    "nlam_lin_multi_supported": [],
    "nlam_lin_bwd_multi": [_i32, _i32] + [_p] * 28 + [_p],
    "nlam_node_chain_supported": [],
    "nlam_node_fwd": [_p, _i64, _i64, _p, _i64, _i64, _p, _i64, _p, _p, _i64, _p, _p, _p,
                      _p, _i64, _i64, _p, _i64, _p, _p, _i64, _p, _p, _i64, _i64, _i64, _i64, _p],
    "nlam_node_bwd_slab_stride": [],
    "nlam_node_bwd_grid": [_i64, _i64],
    "nlam_node_bwd": [_p, _i64, _p, _p, _i64, _p, _i64, _i64, _p, _i64, _i64, _p, _i64, _p, _i64,
                      _p, _i64, _i64, _p, _i64, _i64, _p, _i64, _p, _p, _i64, _p, _p,
                      _p, _i64, _i64, _p, _i64, _i64, _p, _p, _i64, _i64, _i64, _p],
    "nlam_node_outer_slab_stride": [],
    "nlam_node_outer_grid": [_i64, _i64],
    "nlam_node_outer": [_p, _p, _i64, _i64, _p, _i64, _i64, _p, _i64, _i64, _p, _i64, _i64,
                        _p, _i64, _i64, _i64, _p],
    "nlam_reduce_slabs": [_p, _i64, _i64, _i64, _p, _i32, _p],
    "nlam_reduce_slabs_multi": [_p, _i64, _i64, _i32, _p, _p, _p, _p, _p, _p, _p],
    "nlam_reduce_slabs_batch": [_i32, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p],
    "nlam_lin_fwd": [_p, _i64, _i64, _i32, _p, _i64, _p, _i32, _p, _i64, _p, _i32, _p, _i64, _i64,
                     _i64, _i64, _i32, _p],
    "nlam_edge_fwd": [_p, _i64, _p, _p, _p, _p, _p, _p, _i64, _i64, _i32, _p, _i64, _i64, _p, _i64,
                      _i64, _p, _i64, _p, _i64, _p, _p, _p, _p, _i64, _i64, _p, _i64, _i64, _i64,
                      _i32, _p],
    "nlam_graph_tiles_host": [_p, _i64, _i32, _i32, _p, _i64],
    "nlam_lin_bwd_slab_stride": [_i32, _i32],
    "nlam_lin_bwd": [_p, _i64, _i64, _i32, _p, _i64, _i64, _p, _i64, _i32, _p, _i64, _i32, _p, _i64,
                     _i64, _p, _i64, _i64, _i64, _i64, _p, _i64, _i64, _i64, _p],
    "nlam_edge_bwd_slab_stride": [_i32],
    "nlam_edge_bwd": [_p, _i64, _p, _p, _p, _p, _p, _p, _i64, _i64, _i32, _p, _i64, _i64, _p, _i64,
                      _i64, _p, _i64, _p, _i64, _p, _p, _p, _i64, _i64, _p, _i64, _i64, _p, _i64,
                      _p, _i64, _i64, _p, _i64, _i64, _p, _i64, _i64, _i32, _p],
    "nlam_adamw_step": [_p, _p, _p, _p, _i64, _f, _f, _f, _f, _f, _i64, _f, _p],
    "nlam_affine_residual": [_p, _p, _p, _p, _p, _i64, _i32, _p],
    "nlam_scale_cols": [_p, _p, _p, _i64, _i32, _p],
    "nlam_boundary_mix": [_p, _p, _p, _p, _i64, _i64, _i32, _p],
    "nlam_wmse_blocks": [],
    "nlam_wmse_fwd": [_p, _p, _p, _p, _p, _p, _i64, _i64, _i32, _f, _p],
    "nlam_wmse_bwd": [_p, _p, _p, _p, _p, _f, _p, _i64, _i64, _i32, _p],
    "nlam_tail_fwd": [_p, _i64, _i64, _p, _p,
                      _p, _i64, _i64, _p, _p, _i64, _i64, _p, _p, _i64, _i64, _p,
                      _p, _i64, _p, _p, _p, _i32, _p, _i64, _p, _i64,
                      _p, _i64, _i64, _p, _p, _i64, _i64,
                      _p, _i64, _i64, _p, _i64, _i32, _i32, _p],
    "nlam_edge_bwd_parts_supported": [_i64, _i64, _i32],
    "nlam_edge_bwd_parts": [_p, _i64, _p, _p, _p, _p, _p, _p, _i64, _p, _i64, _i64, _p, _i64, _i64,
                            _p, _i64, _p, _p, _p, _i64, _i64, _p, _p, _i64, _p, _i64, _i64, _p, _i64,
                            _p, _i64, _i64, _i32, _p],
    "nlam_tail_fwd_pre_supported": [_i32, _i64, _i64],
    "nlam_tail_fwd_pre": [_i64, _p, _i64, _i64, _p, _i64, _i64, _p, _i64, _p, _i64, _p, _p, _p,
                          _p, _i64, _p, _i64, _i64, _p, _i64, _i64, _i64, _i32, _p],
    "nlam_mlp_tail_multi_shares": [_i32, _p, _p, _p],
    "nlam_mlp_tail_fwd_multi": [_i32, _i32] + [_p] * 9 + [_p],
    "nlam_mlp_tail_bwd_multi": [_i32, _i32] + [_p] * 12 + [_p],
    "nlam_tail_bwd_slab_stride": [_i32],
    "nlam_tail_bwd": [_p, _i64, _i64, _p, _p, _p, _i64, _p, _i64,
                      _p, _i64, _i64, _p, _p, _p, _i64, _i64, _p,
                      _p, _i64, _p, _p, _i32, _p, _i64,
                      _p, _i64, _i64, _p, _p, _i64, _i64,
                      _p, _i64, _i64, _i32, _i32, _p],
    "nlam_lin_bwd_data": [_p, _i64, _i64, _i32, _p, _i64, _i32, _p, _i64, _i64, _p, _i64, _i64,
                          _i64, _i64, _p],
    "nlam_wide_outer": [_p, _i64, _i64, _i32, _p, _i64, _i64, _i32, _i32, _p, _i64, _i64, _i64, _i32,
                        _p],
    "nlam_lin_fwd_multi": [_i32, _i32] + [_p] * 11 + [_i32, _p],
    "nlam_lin_bwd_data_multi": [_i32, _i32] + [_p] * 13 + [_p],
    "nlam_wide_outer_multi": [_i32, _i32] + [_p] * 13 + [_p],
    "nlam_wide_outer_multi_nx": [_i32, _i32] + [_p] * 12 + [_p],
    "nlam_concat_rows": [_i32, _p, _p, _p, _p, _p, _i64, _i64, _p],
    "nlam_sizeof_inet_args": [],
    "nlam_sizeof_inet_grads": [],
    "nlam_grid_encode_supported": [],
    "nlam_grid_encode_fwd": [_i32, _p, _p, _p, _p,                      # sources
                             _p, _i64, _p, _p, _i64, _p, _p, _p,        # grid_embedder
                             _p, _i64,                                  # Ws
                             _p, _i64, _p, _p, _i64, _p, _p, _p,        # encoding MLP
                             _p, _i64, _p,                              # Wr, br
                             _p, _p, _p, _p, _p, _i64, _i64, _p],
    "nlam_state_step": [_p, _i64, _p, _p, _i64, _p, _p, _p, _p, _i64, _i64, _i32, _p],
    "nlam_state_step_bwd": [_p, _p, _p, _p, _p, _i64, _i64, _i32, _p],
    "nlam_state_step_wmse_blocks": [],
    "nlam_state_step_wmse_fwd": [_p, _i64, _p, _p, _i64, _p, _p, _p, _p, _p, _p, _p, _p, _f, _i64, _i64,
                                 _i32, _p],
    "nlam_state_step_wmse_bwd": [_p, _p, _i64, _p, _p, _p, _p, _p, _f, _p, _p, _p, _i64, _i64, _i32, _p],
    "nlam_sum_many": [_i32, _p, _p, _i64, _p],
    "nlam_pack_chunk": [],
    "nlam_pack_segments": [_p, _i32, _i64, _p, _p],
    "nlam_std_head_fwd": [_p, _p, _p, _p, _p, _p, _i64, _i32, _p],
    "nlam_std_head_bwd": [_p, _p, _p, _p, _p, _i64, _i32, _p],
    "nlam_nll_fwd": [_p, _p, _p, _p, _p, _p, _i64, _i64, _i32, _f, _p],
    "nlam_nll_bwd": [_p, _p, _p, _p, _p, _f, _p, _p, _i64, _i64, _i32, _p],
    "nlam_debug_edge_bwd_stamps": [_p, _i32],
    "nlam_debug_mlp_bwd_stamps": [_p, _i32],
    "nlam_debug_fs_stamps": [_p, _i32],
    "nlam_debug_multi_shares": [_i32, _p, _i64, _p],
    "nlam_debug_lin_fwd_timeline": [_p],
    "nlam_debug_node_timeline": [_p],
    "nlam_mfma_probe": [_p, _p],
    "nlam_set_k16": [_i32],
}
_RESTYPES = {
    "nlam_last_error": ctypes.c_char_p,
    "nlam_layernorm_bwd_blocks": _i64,
    "nlam_colsum_blocks": _i64,
    "nlam_graph_tiles_host": _i64,
    "nlam_bwd_grid": _i64,
    "nlam_lin_bwd_slab_stride": _i64,
    "nlam_edge_bwd_slab_stride": _i64,
    "nlam_wmse_blocks": _i64,
    "nlam_pack_chunk": _i64,
    "nlam_mlp_bwd_slab_stride": _i64,
    "nlam_outer_bwd_slab_stride": _i64,
    "nlam_inet_bwd_workspace": _i64,
    "nlam_sizeof_inet_args": _i64,
    "nlam_sizeof_inet_grads": _i64,
    "nlam_node_bwd_slab_stride": _i64,
    "nlam_node_bwd_grid": _i64,
    "nlam_node_outer_slab_stride": _i64,
    "nlam_node_outer_grid": _i64,
    "nlam_tail_bwd_slab_stride": _i64,
}


class NlamError(RuntimeError):
    pass


ABI_VERSION = 4   # include/nlam_hip.h NLAM_ABI_VERSION (tests/test_host_logic.py compares the two)
MFMA_MODE_NAMES = ("fp32", "bf16x3", "b3", "bf16")


def _load():
    mode = os.environ.get("NLAM_MFMA", "")
    if mode and mode.lower() not in MFMA_MODE_NAMES:
        raise NlamError(f"NLAM_MFMA={mode!r} is not one of fp32 | bf16x3 | bf16")
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (hipcc --offload-arch=gfx950); there is no fallback path."
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise ImportError(f"{LIB_PATH} does not export {name}; rebuild it") from e
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, ctypes.c_int)
    got = lib.nlam_abi_version()
    if got != ABI_VERSION:
        raise ImportError(
            f"{LIB_PATH} has ABI version {got}, this binding was written against {ABI_VERSION} "
            "(include/nlam_hip.h NLAM_ABI_VERSION): rebuild the library"
        )
    return lib


lib = _load()


def check(code, what):
    if code != 0:
        raise NlamError(f"{what} failed ({code}): {lib.nlam_last_error().decode()}")
