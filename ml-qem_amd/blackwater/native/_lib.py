"""Loads ``libmlqem_hip.so`` (the C ABI of include/mlqem_hip.h) through ctypes.

There is deliberately no fallback: if the library is missing or a call returns an error code this raises.
PyTorch appears only as the owner of device memory and streams -- every argument that crosses the boundary is
a raw device pointer, a size or a stream handle.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_size_t, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# MLQEM_LIB: another build of the same library (A/B measurements of one kernel on one box: scripts/ab_*.sh)
LIB_PATH = os.environ.get("MLQEM_LIB") or os.path.normpath(os.path.join(_HERE, "..", "..", "csrc", "libmlqem_hip.so"))


class NativeLibraryError(RuntimeError):
    pass


_P, _I, _L, _F, _S, _U, _D = c_void_p, c_int, c_int64, c_float, c_size_t, c_uint64, ctypes.c_double


MAX_COL_PARTS = 8   # MLQEM_MAX_COL_PARTS


MAX_HEAD_TERMS = 8   # MLQEM_HEAD_MAX_TERMS


class HeadDesc(ctypes.Structure):
    """``mlqem_head_desc``: the terms P_t . W_t of a pooled head and the output column each adds to."""

    _fields_ = [("n_terms", ctypes.c_int32), ("n_cols", ctypes.c_int32), ("P", c_void_p * MAX_HEAD_TERMS),
                ("ldp", c_int64 * MAX_HEAD_TERMS), ("W", c_void_p * MAX_HEAD_TERMS), ("col", ctypes.c_int32 * MAX_HEAD_TERMS),
                ("bias", c_void_p * MAX_HEAD_TERMS)]


class ColParts(ctypes.Structure):
    """``mlqem_col_parts``: a matrix given as up to four column blocks in separate buffers."""

    _fields_ = [("count", ctypes.c_int32), ("width", ctypes.c_int32), ("cols", ctypes.c_int32),
                ("reserved", ctypes.c_int32), ("ptr", c_void_p * MAX_COL_PARTS), ("ld", c_int64 * MAX_COL_PARTS)]


# name -> (restype, argtypes); kept in the order of include/mlqem_hip.h
SIGNATURES = {
    "mlqem_abi_version": (_I, []),
    "mlqem_error_string": (c_char_p, [_I]),
    "mlqem_csr_build_workspace_bytes": (_S, [_L, _L]),
    "mlqem_csr_build": (_I, [_P, _L, _L, _P, _P, _P, _P, _P, _P, _P, _S, _P]),
    "mlqem_graph_norms": (_I, [_P, _P, _P, _L, _P, _P, _P, _P]),
    "mlqem_batch_assemble": (_I, [_P, _L, _I, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _L, _L,
                                  _P, _L, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "mlqem_csr_aggregate_f32": (_I, [_P, _L, _P, _P, _P, _P, _P, _P, _F, _F, _P, _L, _P, _I, _F, _U, _P, _P, _L, _L, _I, _P]),
    "mlqem_csr_aggregate_pool_workspace_bytes": (_S, [_L, _L, _I]),
    "mlqem_csr_aggregate_pool_gate_bytes": (_S, [_L, _I]),
    "mlqem_csr_aggregate_pool_f32": (_I, [_P, _L, _P, _P, _P, _P, _P, _P, _F, _F, _P, _L, _P, _I, _F, _U, _P, _P, _L, _L, _I,
                                          _P, _P, _L, _P, _L, _P, _L, _P, _P, _S, _P]),
    "mlqem_csr_segment_max_f32": (_I, [_P, _L, _P, _P, _P, _P, _L, _L, _I, _P]),
    "mlqem_ell_from_csr": (_I, [_P, _P, _L, _P, _P]),
    "mlqem_mse_loss_workspace_bytes": (_S, []),
    "mlqem_seq2_forward_f32": (_I, [_P, _L, _L, _I, _P, _P, _I, _P, _P, _I, _F, _U, _P, _P, _P, _P, _L, _P]),
    "mlqem_seq2_backward_workspace_bytes": (_S, [_L, _I, _I, _I]),
    "mlqem_seq2_backward_f32": (_I, [_P, _L, _P, _L, _L, _I, _P, _I, _P, _I, _P, _P, _F, _P, _L, _P, _P, _P, _P, _P, _S, _P, _P]),
    "mlqem_mse_loss_grad_f32": (_I, [_P, _L, _P, _L, _P, _L, _L, _I, _L, _P, _P, _S, _P, _P]),
    "mlqem_adam_step_f32": (_I, [_P, _P, _P, _P, _L, _P, _P, _D, _D, _D, _P, _P, _P]),
    "mlqem_relu_dropout_bwd_f32": (_I, [_P, _L, _P, _L, _F, _P, _L, _L, _I, _P]),
    "mlqem_relu_dropout_f32": (_I, [_P, _L, _F, _U, _P, _P, _L, _P, _L, _P, _L, _L, _I, _P]),
    "mlqem_linear_f32": (_I, [_P, _L, _P, _I, _P, _P, _P, _L, _L, _I, _I, _I, _I, _F, _U, _I, _I, _P, _L, _F, _P, _P]),
    "mlqem_linear_bf16_f32": (_I, [_P, _L, _P, _I, _P, _P, _L, _L, _I, _I, _I, _P]),
    "mlqem_linear_wgrad_bf16_f32": (_I, [_P, _L, _P, _L, _P, _P, _L, _I, _I, _I, _P, _S, _P]),
    "mlqem_linear_parts_f32": (_I, [_P, _P, _P, _I, _P, _P, _P, _L, _P, _L, _F, _P, _P]),
    "mlqem_linear_wgrad_workspace_bytes": (_S, [_I, _I]),
    "mlqem_linear_wgrad_f32": (_I, [_P, _L, _P, _L, _P, _P, _L, _I, _I, _I, _P, _S, _P, _P]),
    "mlqem_linear_wgrad_parts_f32": (_I, [_P, _P, _L, _P, _P, _L, _I, _I, _P, _S, _P, _P]),
    "mlqem_mlp1_workspace_bytes": (_S, [_I, _I]),
    "mlqem_mlp1_forward": (_I, [_P, _L, _P, _P, _P, _P, _P, _P, _L, _L, _I, _I, _I, _I, _P, _L, _P, _L, _P, _S, _P]),
    "mlqem_mlp1_backward": (_I, [_P, _L, _P, _L, _P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _I, _P, _P, _S, _P]),
    "mlqem_layer_workspace_bytes": (_S, []),
    "mlqem_layer_gemm_bf16": (_I, [_P, _I, _L, _P, _I, _P, _P, _P, _I, _L, _I, _F, _U, _P, _L, _I, _I, _P, _S, _P]),
    "mlqem_layer_colstats_bf16": (_I, [_I, _P, _P, _P, _L, _P, _P, _P, _P, _P, _P, _F, _I, _F, _U, _P, _L, _I, _P, _P, _P, _P, _P,
                                       _P, _P, _F, _P, _P, _S, _P]),
    "mlqem_layer_pointwise_bf16": (_I, [_I, _P, _P, _P, _L, _P, _P, _P, _P, _P, _P, _P, _P, _I, _F, _U, _P, _P, _L, _I, _P]),
    "mlqem_layer_wgrad_bf16": (_I, [_P, _P, _I, _L, _P, _P, _L, _I, _I, _P, _S, _P]),
    "mlqem_layer_rowdot_bf16": (_I, [_P, _P, _P, _P, _L, _L, _I, _I, _P]),
    "mlqem_layer_rowdot_bwd_bf16": (_I, [_P, _L, _P, _P, _P, _F, _P, _P, _L, _I, _I, _P, _S, _P]),
    "mlqem_layer_gemm_f32": (_I, [_P, _L, _P, _I, _P, _P, _P, _I, _L, _I, _F, _U, _P, _L, _I, _I, _P, _S, _P]),
    "mlqem_layer_colstats_f32": (_I, [_I, _P, _P, _P, _L, _P, _P, _P, _P, _P, _P, _F, _I, _F, _U, _P, _L, _I, _P, _P, _P, _P, _P,
                                       _P, _P, _F, _P, _P, _S, _P]),
    "mlqem_layer_pointwise_f32": (_I, [_I, _P, _P, _P, _L, _P, _P, _P, _P, _P, _P, _P, _P, _I, _F, _U, _P, _P, _L, _I, _P]),
    "mlqem_layer_wgrad_f32": (_I, [_P, _P, _L, _P, _P, _L, _I, _I, _P, _S, _P]),
    "mlqem_layer_rowdot_f32": (_I, [_P, _P, _P, _P, _L, _L, _I, _I, _P]),
    "mlqem_layer_rowdot_bwd_f32": (_I, [_P, _L, _P, _P, _P, _F, _P, _P, _L, _I, _I, _P, _S, _P]),
    "mlqem_linear_bwd_fused_f32": (_I, [_P, _L, _P, _L, _P, _L, _P, _I, _F, _P, _L, _P, _P, _L, _I, _I, _P, _S, _P]),
    "mlqem_pooled_head_f32": (_I, [_P, _L, _I, _P, _L, _P]),
    "mlqem_pooled_head_bwd_f32": (_I, [_P, _P, _L, _L, _I, _P, _P, _P, _P, _P]),
    "mlqem_segment_pool_workspace_bytes": (_S, [_L, _L, _I]),
    "mlqem_segment_pool_f32": (_I, [_P, _L, _P, _P, _L, _L, _I, _P, _L, _P, _L, _P, _S, _P]),
    "mlqem_pooled_grad_aggregate_supported": (_I, [_I]),
    "mlqem_pooled_grad_colsum_groups": (_I, [_L]),
    "mlqem_pooled_grad_colsum_f32": (_I, [_P, _P, _P, _L, _P, _L, _P, _L, _F, _L, _I, _P, _P]),
    "mlqem_pooled_grad_aggregate_f32": (_I, [_P, _P, _P, _P, _L, _P, _L, _P, _L, _F, _P, _P, _P, _P, _P, _F, _P, _L, _P, _L, _L, _I, _P]),
    "mlqem_segment_pool_bwd_f32": (_I, [_P, _L, _P, _L, _P, _P, _L, _L, _I, _P, _L, _F, _P, _P, _L, _P]),
    "mlqem_transformer_attention_f32": (_I, [_P, _L, _P, _P, _P, _L, _I, _I, _P, _L, _P]),
    "mlqem_csr_softmax_aggregate_f32": (_I, [_P, _L, _P, _P, _P, _P, _F, _L, _I, _P, _L, _P]),
    "mlqem_leconv_fitness_f32": (_I, [_P, _P, _P, _L, _P, _I, _P]),
    "mlqem_gather_scale_rows_f32": (_I, [_P, _L, _P, _P, _L, _I, _P, _L, _P]),
    "mlqem_pool_keep_ptr": (_I, [_P, _L, _F, _P, _P]),
    "mlqem_rank_grad_workspace_bytes": (_S, [_I]),
    "mlqem_rank_grad_f32": (_I, [_I, _P, _P, _P, _P, _P, _L, _I, _P, _P, _P, _S, _P]),
    "mlqem_segment_topk_workspace_bytes": (_S, [_L, _L]),
    "mlqem_segment_topk": (_I, [_P, _P, _P, _L, _L, _L, _L, _P, _P, _P, _S, _P]),
    "mlqem_asap_coarsen_workspace_bytes": (_S, [_L]),
    "mlqem_asap_hop1_count": (_I, [_P, _P, _P, _P, _P, _L, _L, _P, _P, _P, _S, _P]),
    "mlqem_asap_hop1_fill": (_I, [_P, _P, _P, _P, _P, _P, _L, _P, _P]),
    "mlqem_asap_hop2_count": (_I, [_P, _L, _P, _P, _P, _P, _P, _S, _P]),
    "mlqem_asap_hop2_fill": (_I, [_P, _L, _P, _P, _P, _P, _P, _P]),
    "mlqem_sort_unique_u64_workspace_bytes": (_S, [_L]),
    "mlqem_sort_unique_u64": (_I, [_P, _L, _P, _P, _P, _S, _P]),
    "mlqem_keys_to_edge_index": (_I, [_P, _L, _P, _P]),
    "mlqem_asap_coarsen_rows_workspace_bytes": (_S, [_L, _I]),
    "mlqem_asap_coarsen_rows_max_bits": (_I, []),
    "mlqem_asap_slot_map": (_I, [_P, _L, _L, _P, _P]),
    "mlqem_batch_norm_workspace_bytes": (_S, [_L, _I]),
    "mlqem_batch_norm_train_f32": (_I, [_P, _L, _L, _I, _P, _P, _F, _P, _L, _P, _P, _P, _P, _S, _P]),
    "mlqem_batch_norm_train_bwd_f32": (_I, [_P, _L, _P, _L, _L, _I, _P, _P, _P, _P, _L, _P, _P, _P, _S, _P]),
    "mlqem_asap_coarsen_rows_count": (_I, [_P, _P, _P, _P, _P, _P, _P, _L, _L, _L, _I, _I, _P, _P, _P, _P, _S, _P]),
    "mlqem_asap_coarsen_rows_fill": (_I, [_P, _L, _L, _I, _P, _P, _P, _P, _P, _P, _S, _P]),
    "mlqem_asap_coarsen_lists_workspace_bytes": (_S, [_L, _L, _L, _L]),
    "mlqem_asap_coarsen_lists_max_k": (_I, []),
    "mlqem_asap_coarsen_lists_caps": (_I, [_P, _P, _P, _P, _P, _P, _L, _L, _L, _P, _P, _S, _P]),
    "mlqem_asap_coarsen_lists_count": (_I, [_P, _P, _P, _P, _P, _P, _P, _L, _L, _L, _L, _I, _L, _P, _I, _P, _P, _P, _P, _S, _P]),
    "mlqem_asap_coarsen_lists_fill": (_I, [_L, _L, _L, _L, _P, _P, _P, _P, _P, _L, _P, _P, _S, _P]),
    "mlqem_asap_coarsen_dense_max_k": (_I, []),
    "mlqem_asap_coarsen_dense_workspace_bytes": (_S, [_L, _L, _I]),
    "mlqem_asap_coarsen_dense": (_I, [_P, _P, _P, _P, _P, _P, _P, _L, _L, _L, _I, _P, _I, _P, _P, _P, _P, _P, _P, _P, _S, _P]),
    "mlqem_gather_rows_f32": (_I, [_I, _P, _P, _P, _L, _P, _P]),
    "mlqem_asap_slot_map_graphs": (_I, [_P, _P, _P, _L, _L, _L, _P, _P]),
    "mlqem_pad_head_rows_parts_f32": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _P, _P]),
    "mlqem_transformer_attention_train_f32": (_I, [_P, _L, _P, _P, _P, _L, _L, _I, _I, _F, _U, _P, _I, _P, _I, _P, _L, _P, _L, _P, _P, _P]),
    "mlqem_transformer_attention_bwd_f32": (_I, [_P, _L, _P, _L, _P, _L, _P, _P, _P, _P, _P, _P, _P, _P, _L, _L, _I, _I,
                                                 _F, _U, _P, _I, _I, _P, _L, _P, _P, _P]),
    "mlqem_csr_softmax_aggregate_bwd_f32": (_I, [_P, _L, _P, _L, _P, _L, _P, _P, _P, _P, _P, _P, _P, _F, _L, _L, _I, _I,
                                                 _P, _L, _P, _P, _P, _P, _P, _L, _P, _L, _P, _P, _P]),
    "mlqem_csr_segment_max_bwd_f32": (_I, [_P, _L, _P, _L, _P, _L, _P, _P, _P, _P, _L, _I, _P, _L, _P, _L, _P, _L, _P, _P, _P]),
    "mlqem_gather_scale_rows_bwd_f32": (_I, [_P, _L, _P, _L, _P, _P, _L, _I, _P, _L, _P, _P]),
    "mlqem_leconv_fitness_bwd_f32": (_I, [_P, _P, _P, _P, _P, _L, _P, _P]),
    "mlqem_tile_order_by_position": (_I, [_P, _P, _P, _L, _P, _P]),
    "mlqem_dense_plan_record_ints": (_I, []),
    "mlqem_dense_plan_min_degree": (_I, []),
    "mlqem_dense_plan_max_blocks": (_L, [_L, _L]),
    "mlqem_dense_plan_build": (_I, [_P, _P, _P, _P, _P, _L, _L, _L, _P, _P, _P, _P, _P]),
    "mlqem_dense_attention_supported": (_I, [_I, _I, _I]),
    "mlqem_dense_attention_train_f32": (_I, [_P, _L, _P, _P, _P, _L, _L, _I, _I, _F, _U, _P, _I, _P, _P, _P, _L, _I, _P, _L, _P, _L, _P,
                                             _P, _P]),
    "mlqem_dense_attention_bwd_f32": (_I, [_P, _L, _P, _L, _P, _L, _P, _P, _P, _P, _P, _P, _P, _L, _L, _I, _I, _F, _U, _P, _I,
                                           _P, _P, _P, _L, _P, _P, _P, _L, _I, _P, _L, _P, _P]),
    "mlqem_dense_pool_supported": (_I, [_I]),
    "mlqem_dense_softmax_aggregate_f32": (_I, [_P, _L, _P, _P, _P, _P, _F, _L, _I, _P, _P, _P, _L, _P, _L, _P, _P]),
    "mlqem_dense_leconv_fitness_bwd_f32": (_I, [_P, _P, _P, _P, _P, _L, _P, _P, _P, _L, _P, _P]),
    "mlqem_dense_segment_max_f32": (_I, [_P, _L, _P, _P, _L, _I, _P, _P, _P, _L, _P, _L, _P]),
    "mlqem_dense_softmax_aggregate_bwd_f32": (_I, [_P, _L, _P, _L, _P, _L, _P, _P, _P, _P, _P, _P, _F, _L, _L, _I, _P, _P, _P, _P, _L,
                                                   _P, _P, _P, _L, _P, _L, _P, _P, _P, _P, _L, _P, _L, _P, _P]),
    "mlqem_dense_segment_max_bwd_f32": (_I, [_P, _L, _P, _L, _P, _P, _P, _P, _L, _I, _P, _L, _P, _L, _P, _L, _P, _P, _P, _P, _P, _L, _P]),
    "mlqem_gather_rows_dot_f32": (_I, [_P, _L, _P, _L, _P, _L, _I, _P, _P]),
    "mlqem_scatter_scale_rank_f32": (_I, [_P, _L, _P, _P, _P, _L, _P, _I, _L, _I, _P, _L, _P]),
    "mlqem_asap_scores_fused_f32": (_I, [_P, _L, _P, _P, _P, _P, _P, _P, _P, _F, _L, _I, _P, _L, _P, _P, _P, _L, _P, _P]),
    "mlqem_pad_head_rows_f32": (_I, [_P, _P, _I, _I, _I, _I, _P, _P, _P]),
    "mlqem_unpad_head_rows_f32": (_I, [_P, _P, _I, _I, _I, _I, _P, _P, _P]),
    "mlqem_asap_compose_f32": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P]),
    "mlqem_asap_compose_bwd_f32": (_I, [_P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P]),
    "mlqem_encode_qasm": (_I, [c_char_p, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P]),
    "mlqem_encode_last_error": (c_char_p, []),
    "mlqem_qasm_batch_parse": (_I, [_P, _L, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P]),
    "mlqem_qasm_batch_fill": (_I, [_P, _I, _P, _P, _P, _P]),
    "mlqem_qasm_batch_free": (None, [_P]),
    "mlqem_qasm_batch_stream_sizes": (_I, [_P, _P, _P, _P]),
    "mlqem_qasm_batch_stream_fill": (_I, [_P, _I, _P, _P, _P, _P, _P]),
    "mlqem_props_gate_tables": (_I, [_P, _P, _P]),
    "mlqem_encode_expand_workspace_bytes": (_S, [_L, _L]),
    "mlqem_encode_expand": (_I, [_P, _P, _P, _L, _P, _L, _L, _L, _L, _I, _P, _P, _P, _I, _P, _P, _P, _P, _I, _I, _I, _P, _L, _P, _P, _P,
                                 _P, _S, _P]),
    "mlqem_circuit_features_qasm": (_I, [c_char_p, _P, _I, _P, _I, _P, _P]),
    "mlqem_circuit_features_qasm_batch": (_I, [_P, _L, _P, _I, _P, _I, _I, _P, _P, _P]),
}

_lib = None
ERR_UNSUPPORTED = -2   # MLQEM_ERR_UNSUPPORTED: a shape this kernel does not serve
ERR_WORKSPACE = -4   # MLQEM_ERR_WORKSPACE: a caller-provided buffer is too small (the encoder then says what it needs)
ABI_VERSION = 42   # MLQEM_ABI_VERSION of include/mlqem_hip.h; bumped whenever a signature changes


def load() -> ctypes.CDLL:
    """Returns the loaded library, loading it on first use."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryError(
            f"{LIB_PATH} not found: build it with `make -C {os.path.dirname(LIB_PATH)}` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback."
        )
    # torch ships its own libamdhip64 (soname libamdhip64.so.7).  It must be in the process BEFORE this library is
    # mapped so that both resolve to ONE HIP runtime; loaded the other way round, /opt/rocm's copy comes in as a
    # second runtime and every launch on torch-owned memory fails.
    import torch  # noqa: F401

    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as exc:
            raise NativeLibraryError(f"{LIB_PATH} does not export {name}") from exc
        fn.restype, fn.argtypes = restype, argtypes
    found = lib.mlqem_abi_version()
    if found != ABI_VERSION:   # a stale build would be called with the wrong argument lists
        raise NativeLibraryError(f"{LIB_PATH} has ABI version {found}, this binding expects {ABI_VERSION}: rebuild it")
    _lib = lib
    return lib


_SYNC_OPS = os.environ.get("MLQEM_SYNC_OPS", "0") == "1"     # diagnostics: name every native call, then wait for it (a fault then has a name)


def check(code: int, what: str) -> None:
    if _SYNC_OPS:
        import sys

        import torch

        print(f"[native] {what}", file=sys.stderr, flush=True)
        torch.cuda.synchronize()
    if code != 0:
        msg = load().mlqem_error_string(code).decode()
        raise NativeLibraryError(f"{what} failed with code {code}: {msg}")
