"""ctypes binding of libgroupnet_hip.so (the C ABI declared in include/groupnet_hip.h).

There is no CPU fallback: if the shared object is missing or a tensor is not on the
GPU the call raises.  `import torch` must precede loading the library so that it binds
to the HIP runtime torch has already loaded (same SONAME, libamdhip64.so.7).
"""
from __future__ import annotations

import ctypes
import os
import threading

import torch  # noqa: F401  (must be imported before the .so is loaded)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GROUPNET_HIP_LIB") or os.path.join(_HERE, "libgroupnet_hip.so")  # env: tuning builds
ABI_VERSION = 34

GN_OK = 0
GN_ERR_K_RANGE = -3

_P = ctypes.c_void_p
_I = ctypes.c_int
_F = ctypes.c_float
_SZ = ctypes.c_size_t
_U64 = ctypes.c_ulonglong

class NodeGroup(ctypes.Structure):      # gn_node_group_t
    _fields_ = [("x", _P), ("W", _P), ("bias", _P), ("xp", _P), ("pq", _P), ("hid_out", _P), ("Wx", _P),
                ("WAx", _P), ("bA", _P), ("A", _P), ("KA", _I), ("Wh", _P), ("WAh", _P)]


class N2EGroup(ctypes.Structure):       # gn_n2e_group_t
    _fields_ = [("xp", _P), ("pq", _P), ("H", _P), ("w2", _P), ("edges", _P), ("b2", _P), ("E", _I), ("sym", _I)]


class EdgeGroup(ctypes.Structure):      # gn_edge_group_t
    _fields_ = [("edges", _P), ("U", _P), ("W", _P), ("bias", _P), ("edge_feat", _P), ("dist", _P),
                ("philox_offset", _U64), ("rows", _I), ("K", _I), ("sym_N", _I), ("keep_z1", _P), ("keep_z", _P),
                ("keep_dh1", _P), ("keep_lgf", _P), ("Wx", _P), ("xp", _P), ("pq", _P), ("pool_H", _P), ("w2", _P),
                ("b2", _P), ("pool_N", _I), ("pool_E", _I), ("Wh", _P)]


class GatherGroup(ctypes.Structure):    # gn_gather_group_t
    _fields_ = [("ori", _P), ("H", _P), ("eo", _P), ("E", _I), ("sym", _I)]


class AggGroup(ctypes.Structure):       # gn_agg_group_t
    _fields_ = [("eo", _P), ("edge_feat", _P), ("W", _P), ("b1", _P), ("b2", _P), ("feat", _P), ("rows", _I),
                ("K", _I), ("ori", _P), ("H", _P), ("E", _I), ("N", _I), ("sym", _I), ("A", _P), ("W2x", _P), ("W12x", _P),
                ("W2h", _P), ("W12h", _P), ("node_form", _I), ("m2x", _P), ("m2h", _P), ("m2bias", _P), ("y", _P), ("ldy", _I),
                ("dout", _I), ("divisor", _F)]


class ScatterGroup(ctypes.Structure):   # gn_scatter_group_t
    _fields_ = [("feat", _P), ("H", _P), ("ori", _P), ("out", _P), ("E", _I), ("sym", _I)]


class Mlp2Group(ctypes.Structure):      # gn_mlp2_group_t
    _fields_ = [("x", _P), ("W", _P), ("bias", _P), ("y", _P), ("feat", _P), ("H", _P), ("ori", _P), ("E", _I),
                ("sym", _I), ("in_out", _P), ("hid_out", _P), ("Wx", _P), ("Wh", _P)]


class BlockExtras(ctypes.Structure):    # gn_block_extras_t
    _fields_ = [("f_out", _P), ("f_out_ld", _I), ("H_cat", _P), ("counter", _P), ("counter_add", _U64),
                ("x_raw", _P), ("x_dim", _I), ("M", _P), ("c", _P), ("f_contig", _P)]


class AffinityJob(ctypes.Structure):   # gn_affinity_job_t
    _fields_ = [("f", _P), ("corr", _P), ("H_list", ctypes.POINTER(_P)), ("k_list", ctypes.POINTER(_I)), ("n_scales", _I),
                ("B", _I), ("N", _I), ("D", _I), ("extras", ctypes.POINTER(BlockExtras))]


class PackSeg(ctypes.Structure):       # gn_pack_seg_t
    _fields_ = [("src", _P), ("dst", _P), ("ld", _I), ("rows", _I), ("cols", _I), ("place_r", _I), ("place_c", _I),
                ("IT", _I), ("scale", _F), ("dst_ld", _I)]


class SplitJob(ctypes.Structure):      # gn_split_job_t
    _fields_ = [("packed", _P), ("out", _P), ("n_tiles", _I), ("reserved", _I)]


class GumbelBwdGroup(ctypes.Structure):  # gn_gumbel_bwd_group_t
    _fields_ = [("dist", _P), ("lgf", _P), ("def_", _P), ("gdist", _P), ("dlgf", _P), ("rows", ctypes.c_longlong),
                ("K", _I), ("sym_N", _I)]


class N2EBwdGroup(ctypes.Structure):     # gn_n2e_bwd_group_t
    _fields_ = [("xp", _P), ("pq", _P), ("H", _P), ("w2", _P), ("b2", _P), ("dedges", _P), ("dxp", _P), ("dpq", _P),
                ("dw2", _P), ("db2", _P), ("E", _I), ("sym", _I)]


class GemmDesc(ctypes.Structure):      # gn_gemm_desc_t
    _fields_ = [("A", _P), ("B", _P), ("C", _P), ("bias", _P), ("mask", _P), ("rs", _P), ("colsum", _P),
                ("M", _I), ("N", _I), ("K", _I), ("lda", _I), ("ldb", _I), ("ldc", _I), ("ldmask", _I), ("rs_ld", _I),
                ("flags", _I), ("alpha", _F), ("beta", _F)]


GEMM_TRANS_A, GEMM_TRANS_B, GEMM_RELU, GEMM_ACCUM, GEMM_TRANS_C = 1, 2, 4, 8, 16
MAX_GROUPS = 10

# name -> (restype, argtypes); mirrors include/groupnet_hip.h one to one
SIGNATURES = {
    "gn_abi_version": (_I, []),
    "gn_strerror": (ctypes.c_char_p, [_I]),
    "gn_affinity_f32": (_I, [_P, _P, _I, _I, _I, _P]),
    "gn_topk_incidence_f32": (_I, [_P, ctypes.POINTER(_P), ctypes.POINTER(_I), _I, _I, _I, _P]),
    "gn_listall_incidence_f32": (_I, [_P, _P, _I, _I, _I, _P]),
    "gn_affinity_topk_f32": (_I, [_P, _P, ctypes.POINTER(_P), ctypes.POINTER(_I), _I, _I, _I, _I,
                                  ctypes.POINTER(BlockExtras), _P]),
    "gn_affinity_topk_bf16": (_I, [_P, _P, ctypes.POINTER(_P), ctypes.POINTER(_I), _I, _I, _I, _I,
                                   ctypes.POINTER(BlockExtras), _P]),
    "gn_packed_elems": (_SZ, [_I, _I]),
    "gn_pack_linear_f32": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "gn_split_bf16_f32": (_I, [_P, _P, _I, _I, _P]),
    "gn_split_bf16_batch_f32": (_I, [_P, _I, _I, _I, _P]),
    "gn_pack_segments_f32": (_I, [_P, _I, _I, _P]),
    "gn_node_mlp_f32": (_I, [ctypes.POINTER(NodeGroup), _I, _I, _P]),
    "gn_node_mlp_bf16": (_I, [ctypes.POINTER(NodeGroup), _I, _I, _P]),
    "gn_node_mlp_affinity_f32": (_I, [ctypes.POINTER(NodeGroup), _I, _I, ctypes.POINTER(AffinityJob), _P]),
    "gn_node_mlp_affinity_bf16": (_I, [ctypes.POINTER(NodeGroup), _I, _I, ctypes.POINTER(AffinityJob), _P]),
    "gn_affinity_tail_lds_limit": (_SZ, []),
    "gn_node2edge_f32": (_I, [ctypes.POINTER(N2EGroup), _I, _I, _I, _P]),
    "gn_node2edge_bf16": (_I, [ctypes.POINTER(N2EGroup), _I, _I, _I, _P]),
    "gn_edge_mlp_gumbel_f32": (_I, [ctypes.POINTER(EdgeGroup), _I, _F, _U64, _P, _P]),
    "gn_edge_mlp_gumbel_bf16": (_I, [ctypes.POINTER(EdgeGroup), _I, _F, _U64, _P, _P]),
    "gn_agg_gather_f32": (_I, [ctypes.POINTER(GatherGroup), _I, _I, _I, _P]),
    "gn_agg_gather_bf16": (_I, [ctypes.POINTER(GatherGroup), _I, _I, _I, _P]),
    "gn_agg_mlp_f32": (_I, [ctypes.POINTER(AggGroup), _I, _P]),
    "gn_agg_mlp_bf16": (_I, [ctypes.POINTER(AggGroup), _I, _P]),
    "gn_node_linear_f32": (_I, [_P, _P, _P, _P, _I, _I, _P]),
    "gn_agg_scatter_f32": (_I, [ctypes.POINTER(ScatterGroup), _I, _I, _I, _F, _P]),
    "gn_agg_scatter_bf16": (_I, [ctypes.POINTER(ScatterGroup), _I, _I, _I, _F, _P]),
    "gn_mlp2_f32": (_I, [ctypes.POINTER(Mlp2Group), _I, _I, _I, _I, _I, _I, _I, _F, _P]),
    "gn_mlp2_bf16": (_I, [ctypes.POINTER(Mlp2Group), _I, _I, _I, _I, _I, _I, _I, _F, _P]),
    "gn_gemm_f32": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _I, _I, _F, _F, _P]),
    "gn_gemm_grouped_f32": (_I, [ctypes.POINTER(GemmDesc), _I, _P]),
    "gn_typed_bwd_f32": (_I, [_P, _P, _P, _I, _P, _P, _P, ctypes.c_longlong, _I, _I, _P]),
    "gn_axpby2d_f32": (_I, [_P, _I, _P, _I, ctypes.c_longlong, _I, _F, _F, _P]),
    "gn_gumbel_ef_f32": (_I, [_P, _P, _P, ctypes.c_longlong, _I, _I, _I, _F, _I, _P]),
    "gn_gumbel_bwd_f32": (_I, [_P, _P, _P, _P, _P, ctypes.c_longlong, _I, _I, _F, _I, _P]),
    "gn_node2edge_bwd_f32": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "gn_gumbel_bwd_grouped_f32": (_I, [ctypes.POINTER(GumbelBwdGroup), _I, _I, _F, _P]),
    "gn_node2edge_bwd_grouped_f32": (_I, [ctypes.POINTER(N2EBwdGroup), _I, _I, _I, _P]),
    "gn_philox_uniform_f32": (_I, [_P, _SZ, _U64, _U64, _P, _P]),
    "gn_counter_add_u64": (_I, [_P, _U64, _P]),
    "gn_copy_2d": (_I, [_P, _SZ, _P, _SZ, _SZ, _I, _P]),
}

_lib = None
_lock = threading.Lock()


class GroupNetHipError(RuntimeError):
    """A launcher of libgroupnet_hip.so returned a GN_ERR_* code."""


def load() -> ctypes.CDLL:
    """Load the shared object once; raise loudly if it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C groupnet_amd/csrc`.  groupnet_amd has no CPU fallback.")
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError here = header/library mismatch
            fn.restype = res
            fn.argtypes = args
        got = lib.gn_abi_version()
        if got != ABI_VERSION:
            raise ImportError(f"libgroupnet_hip.so ABI {got} != expected {ABI_VERSION}; rebuild it")
        _lib = lib
    return _lib


def check(rc: int, what: str) -> None:
    if rc == GN_OK:
        return
    msg = load().gn_strerror(rc).decode()
    if rc == GN_ERR_K_RANGE:
        # torch.topk raises RuntimeError("selected index k out of range") — keep the type
        raise RuntimeError(f"{what}: {msg}")
    raise GroupNetHipError(f"{what}: {msg} (code {rc})")


def stream_handle() -> ctypes.c_void_p:
    """hipStream_t of torch's current stream (launches join the caller's stream / graph capture)."""
    return _P(torch.cuda.current_stream().cuda_stream)
