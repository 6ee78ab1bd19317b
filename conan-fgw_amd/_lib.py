"""ctypes binding of libconan_fgw_hip.so (include/conan_fgw_hip.h).  Fails loudly: there is no CPU fallback."""
from __future__ import annotations

import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libconan_fgw_hip.so")
_LIB = None
ABI_VERSION = 5            # == CONAN_FGW_ABI_VERSION of include/conan_fgw_hip.h (tests/test_abi.py compares the two)

c_int, c_float, c_void_p, c_ll = ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_longlong


class FgwParams(ctypes.Structure):
    """Mirror of `conan_fgw_params` (include/conan_fgw_hip.h)."""
    _fields_ = [("alpha", c_float), ("epsilon", c_float), ("max_iter", c_int), ("tol", c_float), ("inner_tol", c_float),
                ("num_iter_max", c_int), ("stop_thr", c_float), ("fixed_structure", c_int), ("fixed_features", c_int),
                ("warmstart", c_int), ("loss_fun", c_int), ("cs_small_int", c_int)]


class BatchLayout(ctypes.Structure):
    """Mirror of `conan_batch_layout` (include/conan_fgw_hip.h)."""
    _fields_ = [(n, c_int) for n in ("B", "K", "num_graphs", "num_atoms", "num_bond_edges", "max_nodes", "x_dim", "ea_dim")] + \
               [(n, c_ll) for n in ("off_atom_off", "off_bond_off", "off_z", "off_pos", "off_x", "off_bsrc", "off_bdst", "off_battr", "off_y", "bytes")]


class WgradJob(ctypes.Structure):
    """Mirror of `conan_wgrad_job` (include/conan_fgw_hip.h)."""
    _fields_ = [("ws", c_void_p), ("dW", c_void_p), ("dbias", c_void_p), ("M", c_int), ("K", c_int), ("N", c_int), ("slices", c_int)]


class WgradSlabJob(ctypes.Structure):
    """Mirror of `conan_wgrad_slab_job` (include/conan_fgw_hip.h)."""
    _fields_ = [("g", c_void_p), ("x", c_void_p), ("m_dev", c_void_p), ("ws", c_void_p), ("M", c_int), ("K", c_int), ("N", c_int), ("slices", c_int)]


# name -> (restype, argtypes); kept in the header's order.  tests/test_abi.py checks this table against the header.
_P = c_void_p
SIGNATURES = {
    "conan_abi_version": (c_int, []),
    "conan_collate_layout": (c_int, [c_int, c_int, _P, _P, c_int, c_int, ctypes.POINTER(BatchLayout)]),
    "conan_collate_pack": (c_int, [ctypes.POINTER(BatchLayout), _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "conan_collate_unpack": (c_int, [_P, ctypes.POINTER(BatchLayout), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "conan_graph_ptr_from_batch": (c_int, [_P, c_int, c_int, _P, _P]),
    "conan_radius_graph_csr": (c_int, [_P, _P, c_int, c_int, c_float, c_int, c_int, _P, _P, _P, _P, _P, _P]),
    "conan_csr_transpose": (c_int, [_P, c_int, c_int, _P, _P, _P, _P, _P, _P]),
    "conan_edge_pairs": (c_int, [_P, _P, _P, _P, _P, c_int, _P, _P, _P, _P, _P, _P, _P, _P]),
    "conan_edge_index_i64": (c_int, [_P, _P, c_int, _P, _P]),
    "conan_embedding_fwd": (c_int, [_P, _P, c_int, c_int, c_int, _P, _P]),
    "conan_onehot_rows": (c_int, [_P, c_int, c_int, c_int, _P, _P]),
    "conan_embedding_bwd_ws": (c_ll, [c_int, c_int, c_int]),
    "conan_embedding_bwd": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P, _P, _P]),
    "conan_linear_fwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P]),
    "conan_linear_multi_fwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P]),
    "conan_linear_sum_fwd": (c_int, [_P, _P, _P, c_int, c_int, _P, _P, c_int, c_int, _P, _P, _P]),
    "conan_ssp_bwd": (c_int, [_P, _P, c_int, c_int, _P, _P, _P]),
    "conan_linear_wgrad_ws": (c_ll, [c_int, c_int, c_int]),
    "conan_linear_wgrad": (c_int, [_P, _P, c_int, c_int, c_int, _P, _P, _P, _P, _P]),
    "conan_bond_graph_csr": (c_int, [_P, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P]),
    "conan_gat_edge_vec": (c_int, [_P, _P, c_int, c_int, _P, _P]),
    "conan_gat_edge_vec_bwd": (c_int, [_P, _P, _P, c_int, c_int, _P, _P, _P]),
    "conan_gat_node_alpha": (c_int, [_P, _P, _P, c_int, c_int, _P, _P, _P]),
    "conan_gat_aggregate_fwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, _P, _P, c_float, c_int, c_int, _P, _P, _P, _P]),
    "conan_gat_bwd_ws": (c_ll, [c_int, c_int, c_int, c_int]),
    "conan_gat_aggregate_bwd": (c_int, [_P] * 15 + [c_int, _P, c_float, c_int, c_int, c_int, _P, _P, _P, _P]),
    "conan_linear_act_fwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P]),
    "conan_unary_fwd": (c_int, [_P, c_ll, c_int, _P, _P]),
    "conan_unary_bwd": (c_int, [_P, _P, c_ll, c_int, _P, _P]),
    "conan_zero_tail": (c_int, [_P, _P, c_int, c_int, _P]),
    "conan_rbf_wgrad": (c_int, [_P, _P, c_int, _P, c_int, c_float, c_int, _P, _P, _P, _P, _P]),
    "conan_wgrad_batchable": (c_int, [c_int, c_int]),
    "conan_linear_wgrad_scaled": (c_int, [_P, _P, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P]),
    "conan_linear_wgrad_slabs": (c_int, [_P, _P, c_int, c_int, c_int, _P, _P, _P]),
    "conan_rbf_wgrad_slabs": (c_int, [_P, _P, c_int, _P, c_int, c_float, c_int, _P, _P, _P]),
    "conan_wgrad_reduce_batch": (c_int, [ctypes.POINTER(WgradJob), c_int, _P]),
    "conan_linear_wgrad_slabs_batch": (c_int, [ctypes.POINTER(WgradSlabJob), c_int, _P]),
    "conan_filter_bwd_supported": (c_int, [c_int, c_int]),
    "conan_filter_bwd_slices": (c_int, [c_int]),
    "conan_filter_bwd_ws": (c_ll, [c_int, c_int, c_int]),
    "conan_filter_bwd": (c_int, [_P, _P, _P, c_int, _P, c_int, c_float, _P, c_int, _P, _P, _P, _P, _P, _P]),
    "conan_filter_bwd2_supported": (c_int, [c_int, c_int]),
    "conan_filter_bwd2_slices": (c_int, [c_int]),
    "conan_filter_bwd2_ws": (c_ll, [c_int, c_int, c_int]),
    "conan_filter_bwd2": (c_int, [_P, _P, _P, c_int, _P, c_int, c_float, _P, c_int, _P, _P, _P, _P, _P, _P, _P, _P]),
    "conan_rbf_fwd": (c_int, [_P, _P, c_int, _P, c_int, c_float, _P, _P]),
    "conan_cutoff_scale": (c_int, [_P, _P, c_int, c_int, c_float, _P, _P, _P]),
    "conan_stage2_head_supported": (c_int, [c_int]),
    "conan_mse_loss_fwd": (c_int, [_P, _P, c_int, _P, _P, _P]),
    "conan_stage2_head_fwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, c_float, c_int, c_int, c_int, _P, _P, _P, _P, _P]),
    "conan_stage2_head_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_float, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "conan_adam_flat_step": (c_int, [_P, _P, _P, _P, _P, _P, c_ll] + [ctypes.c_double] * 5 + [_P, _P]),
    "conan_grad_clip_flat": (c_int, [_P, c_ll, ctypes.c_double, _P, _P, _P, _P]),
    "conan_mlp2_supported": (c_int, [c_int, c_int, c_int, c_int]),
    "conan_mlp2_fwd": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, _P, _P]),
    "conan_mlp2_bwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, _P, _P]),
    "conan_mlp2_outact_supported": (c_int, [c_int, c_int, c_int, c_int]),
    "conan_mlp2_outact_fwd": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, _P, _P]),
    "conan_mlp2_outact_bwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P]),
    "conan_filter_fused_supported": (c_int, [c_int, c_int]),
    "conan_filter_fwd": (c_int, [_P, _P, c_int, _P, c_int, c_float, c_float, c_int, _P, _P, _P, _P, _P, _P, _P]),
    "conan_filter_cfconv_fwd_supported": (c_int, [c_int, c_int]),
    "conan_filter_cfconv_fwd": (c_int, [_P, _P, _P, _P, _P, c_int, _P, c_int, c_float, c_float, c_int, _P, _P, _P, _P, c_int, _P, _P]),
    "conan_cfconv_fwd": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, _P, _P, _P]),
    "conan_cfconv_bwd_xw_pairs_supported": (c_int, [c_int]),
    "conan_cfconv_bwd_xw_pairs": (c_int, [_P] * 10 + [ctypes.c_float, c_int, c_int, _P, _P, _P, _P]),
    "conan_cfconv_bwd_x": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, _P, _P, _P]),
    "conan_cfconv_bwd_w_pairs": (c_int, [_P, _P, _P, c_int, _P, _P, _P, _P, c_int, _P, c_float, _P, _P, _P]),
    "conan_cfconv_bwd_w": (c_int, [_P, _P, _P, c_int, _P, _P, c_int, _P, c_float, _P, _P]),
    "conan_segment_sum_fwd": (c_int, [_P, _P, c_int, c_int, _P, _P]),
    "conan_segment_sum_bwd": (c_int, [_P, _P, c_int, c_int, _P, _P]),
    "conan_visnet_edge_unit": (c_int, [_P, _P, _P, _P, c_int, _P, _P]),
    "conan_visnet_expnormal": (c_int, [_P, _P, c_int, _P, _P, c_int, c_float, c_float, _P, _P]),
    "conan_visnet_neighbor_scale": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_float, _P]),
    "conan_visnet_neighbor_scale_to": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_float, _P, _P]),
    "conan_concat2": (c_int, [_P, c_int, _P, c_int, c_ll, _P, _P]),
    "conan_visnet_edge_embed": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, _P, _P]),
    "conan_layernorm_fwd": (c_int, [_P, _P, _P, c_int, c_int, c_float, _P, _P]),
    "conan_scale_channels": (c_int, [_P, _P, c_ll, c_int, _P, _P]),
    "conan_scale_channels_add": (c_int, [_P, _P, _P, c_ll, c_int, _P, _P]),
    "conan_visnet_vecdot": (c_int, [_P, c_int, c_int, _P, _P]),
    "conan_visnet_attn_message": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_float, c_int, c_int, c_int, c_int, _P, _P, _P]),
    "conan_visnet_vec_aggregate": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, _P, _P]),
    "conan_visnet_node_update": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, _P, _P, _P]),
    "conan_visnet_edge_update": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, _P, _P, _P]),
    "conan_visnet_spatial_norm": (c_int, [_P, c_int, c_int, _P, _P]),
    "conan_visnet_gate": (c_int, [_P, _P, c_int, c_int, c_int, _P, _P, _P]),
    "conan_visnet_prior": (c_int, [_P, _P, _P, _P, c_int, c_int, _P, _P]),
    "conan_silu_fwd": (c_int, [_P, c_int, c_int, _P, _P, _P]),
    "conan_silu_bwd": (c_int, [_P, _P, c_int, c_int, _P, _P, _P]),
    "conan_split2": (c_int, [_P, c_int, c_int, c_ll, _P, _P, _P]),
    "conan_rowsum": (c_int, [_P, c_int, c_int, _P, _P]),
    "conan_scale_scalar": (c_int, [_P, _P, c_ll, _P, _P]),
    "conan_visnet_edge_embed_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, _P, _P, _P]),
    "conan_layernorm_bwd_ws": (c_ll, [c_int, c_int]),
    "conan_layernorm_bwd": (c_int, [_P, _P, _P, c_int, c_int, c_float, _P, _P, _P, _P, _P]),
    "conan_layernorm_bwd_res": (c_int, [_P, _P, _P, _P, c_int, c_int, c_float, _P, _P, _P, _P, _P]),
    "conan_visnet_vecdot_bwd": (c_int, [_P, _P, c_int, c_int, _P, _P]),
    "conan_visnet_attn_message_bwd": (c_int, [_P] * 13 + [c_float, c_int, c_int, c_int, c_int] + [_P] * 6),
    "conan_visnet_vec_aggregate_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, _P, _P]),
    "conan_visnet_node_update_bwd": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, _P, _P, _P, _P]),
    "conan_visnet_edge_update_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, _P, _P, _P, _P]),
    "conan_visnet_spatial_norm_bwd": (c_int, [_P, _P, c_int, c_int, _P, _P]),
    "conan_visnet_gate_bwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _P, _P, _P]),
    "conan_fgw_densify": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_float, c_float, c_float, _P, _P, _P, _P]),
    "conan_fgw_densify_bwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_float, c_float, c_float, _P, _P]),
    "conan_fgw_workspace_bytes": (c_ll, [c_int, c_int, c_int, c_int]),
    "conan_fgw_barycenter_fwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, ctypes.POINTER(FgwParams),
                                         _P, _P, _P, _P, _P, _P, _P, _P]),
    "conan_fgw_workspace_bytes_ragged": (c_ll, [c_int, c_int, c_int, c_int]),
    "conan_fgw_barycenter_fwd_ragged": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, ctypes.POINTER(FgwParams),
                                                _P, _P, _P, _P, _P, _P, _P, _P]),
    "conan_fgw_barycenter_bwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, _P]),
    "conan_fgw_readout_fwd": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, _P, _P]),
    "conan_fgw_readout_bwd": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P]),
}

_ERR = {-1: "bad argument", -2: "HIP launch failure", -3: "unsupported configuration"}


def library_path() -> str:
    return _SO


def build(force: bool = False) -> str:
    """Compile the HIP sources in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    args = ["make", "-C", os.path.join(_HERE, "csrc"), "-s", "-j8"]
    if force:
        args.append("-B")
    subprocess.check_call(args)
    return _SO


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(_SO):
            raise RuntimeError(
                f"{_SO} is missing: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or make -C conan-fgw_amd/csrc). There is no CPU fallback for this path.")
        L = ctypes.CDLL(_SO)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        if L.conan_abi_version() != ABI_VERSION:
            raise RuntimeError(f"libconan_fgw_hip.so: ABI version {L.conan_abi_version()} != {ABI_VERSION} expected by _lib.py (stale build: "
                               "make -C conan-fgw_amd/csrc)")
        _LIB = L
    return _LIB


def stream_ptr() -> int:
    """Raw hipStream_t of the current stream of the current device.  Goes through the C entry point torch exposes for this
    (no Stream object, no device-index parsing): the call sits in front of every kernel launch — ~190 per training step."""
    return _raw_stream(_cur_device())


try:
    _raw_stream = torch._C._cuda_getCurrentRawStream
    _cur_device = torch._C._cuda_getDevice
except AttributeError:                                          # pragma: no cover - older/newer torch without the private hooks
    def _raw_stream(_dev):
        return torch.cuda.current_stream().cuda_stream

    def _cur_device():
        return torch.cuda.current_device()


def ptr(t, dtype=None):
    """Device pointer of a contiguous CUDA tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("conan_fgw_amd ops run on the GPU only (got a CPU tensor); there is no CPU fallback")
    if not t.is_contiguous():
        raise RuntimeError("conan_fgw_amd ops need contiguous tensors")
    if dtype is not None and t.dtype != dtype:
        raise RuntimeError(f"expected dtype {dtype}, got {t.dtype}")
    return t.data_ptr()


_TRACE = None      # optional callable(name, fn, args) -> rc wrapped around EVERY entry-point call (bench.py / tools: per-entry-point HIP-event brackets)


def set_call_trace(fn):
    """Install (or, with None, remove) a wrapper around every C-ABI call of this process: `fn(name, cfunc, args)` must call `cfunc(*args)` and
    return its result.  Every module binds `call` by name at import, so the hook lives inside the function, not in a module attribute."""
    global _TRACE
    prev, _TRACE = _TRACE, fn
    return prev


def call(name: str, *args):
    fn = getattr(lib(), name)
    rc = fn(*args) if _TRACE is None else _TRACE(name, fn, args)
    if rc != 0:
        raise RuntimeError(f"{name} failed: {_ERR.get(rc, rc)}")
