"""ctypes binding of libcgat_hip.so (C ABI declared in include/cgat_hip.h).

There is no fallback: if the shared library is missing or a symbol cannot be resolved the
import raises, and every compute entry point refuses non-GPU tensors.
"""
import ctypes as C
import os

# torch must initialise its (bundled) HIP runtime before libcgat_hip.so is mapped: loading our
# library first would bind a second copy of libamdhip64 and leave one of the two without devices.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcgat_hip.so")

MAX_FC = 8
MAX_HYPER = 8
ABI_VERSION = 3

ACT_NONE, ACT_TANH, ACT_LEAKY, ACT_RELU = 0, 1, 2, 3

c_float_p = C.POINTER(C.c_float)
c_i32_p = C.POINTER(C.c_int32)
c_i64_p = C.POINTER(C.c_int64)
vp = C.c_void_p


class Plan(C.Structure):
    _fields_ = [("N", C.c_int32), ("E", C.c_int32), ("dst_rowptr", vp), ("dst_perm", vp), ("dst_sorted", vp),
                ("src_sorted", vp), ("src_rowptr", vp), ("src_pos", vp)]


class PackedDatasetStruct(C.Structure):
    _fields_ = [("n_graphs", C.c_int32), ("fea", C.c_int32), ("max_nbr", C.c_int32), ("n_elem", C.c_int32),
                ("table", vp), ("atom_ptr", vp), ("atom_elem", vp), ("shell", vp), ("self_idx", vp), ("nbr_idx", vp),
                ("comp_ptr", vp), ("comp_elem", vp), ("comp_weight", vp), ("y_val", vp)]


class CollatedStruct(C.Structure):
    _fields_ = [("x", vp), ("edge_index", vp), ("edge_attr", vp), ("y", vp), ("batch", vp), ("comp_weight", vp),
                ("comp_fea", vp), ("comp_self", vp), ("comp_nbr", vp), ("comp_crystal", vp)]


class AttnParams(C.Structure):
    _fields_ = [("C", C.c_int32), ("Ce", C.c_int32), ("H", C.c_int32), ("Hd", C.c_int32),
                ("A_in_w", vp), ("A_in_b", vp), ("A_out_w", vp), ("A_out_b", vp),
                ("M_in_w", vp), ("M_in_b", vp), ("M_out_w", vp), ("M_out_b", vp)]


class AttnGrads(C.Structure):
    _fields_ = [("A_in_w", vp), ("A_in_b", vp), ("A_out_w", vp), ("A_out_b", vp),
                ("M_in_w", vp), ("M_in_b", vp), ("M_out_w", vp), ("M_out_b", vp)]


class HyperLinearParams(C.Structure):
    _fields_ = [("fc_w", vp * MAX_FC), ("fc_b", vp * MAX_FC), ("head_w", vp), ("head_b", vp)]


class HnetParams(C.Structure):
    _fields_ = [("W", C.c_int32), ("n_fc", C.c_int32), ("n_hyper", C.c_int32),
                ("layer", HyperLinearParams * MAX_HYPER), ("damping", vp)]


class HnetGrads(C.Structure):
    _fields_ = [("layer", HyperLinearParams * MAX_HYPER), ("damping", vp)]


class GemmDesc(C.Structure):
    _fields_ = [("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("A", vp), ("lda", C.c_int64),
                ("a_kmajor", C.c_int32), ("a_rgather", vp), ("B", vp), ("ldb", C.c_int64), ("b_kmajor", C.c_int32),
                ("b_kgather", vp), ("C", vp), ("ldc", C.c_int64), ("c_scatter", vp), ("alpha", C.c_float),
                ("beta", C.c_float), ("bias", vp), ("add1", vp), ("add1_idx", vp), ("add2", vp), ("add2_idx", vp),
                ("ld_add", C.c_int64), ("act", C.c_int32), ("splits", C.c_int32), ("a_block", C.c_int64)]


# name -> (restype, argtypes); this table is also what tests/test_abi.py checks against the header
class ChainLayer(C.Structure):
    """cgat_chain_layer"""
    _fields_ = [("W", vp), ("w_so", C.c_int64), ("w_sk", C.c_int64), ("bias", vp), ("dact", vp), ("ld_dact", C.c_int64),
                ("resid", vp), ("ld_resid", C.c_int64), ("out", vp), ("ld_out", C.c_int64), ("act", C.c_int32),
                ("dact_type", C.c_int32), ("accumulate", C.c_int32)]


class ChainDesc(C.Structure):
    """cgat_chain_desc"""
    _fields_ = [("n_layers", C.c_int32), ("rows", C.c_int32), ("x", vp), ("ldx", C.c_int64), ("in_dact", vp),
                ("ld_in_dact", C.c_int64), ("in_dact_type", C.c_int32), ("in_store", vp), ("ld_in_store", C.c_int64),
                ("layer", ChainLayer * 5)]


class RowProgOp(C.Structure):
    """cgat_rowprog_op"""
    _fields_ = [("phase", C.c_int32), ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
                ("A", vp), ("a_rs", C.c_int64), ("a_ks", C.c_int64),
                ("dact", vp), ("d_rs", C.c_int64), ("d_ks", C.c_int64), ("dact_type", C.c_int32),
                ("B0", vp), ("b0_rs", C.c_int64), ("b0_ks", C.c_int64),
                ("B1", vp), ("b1_rs", C.c_int64), ("b1_ks", C.c_int64),
                ("bias", vp), ("act", C.c_int32),
                ("resid", vp), ("ld_resid", C.c_int64),
                ("out", vp), ("ldo", C.c_int64), ("alpha", C.c_float), ("beta", C.c_float),
                ("h_out", vp), ("ld_h", C.c_int64),
                ("rowsum", vp)]


ROWPROG_MAX_OPS = 24
ROWPROG_SYNC_WORDS = (1024 + 1) * 16


class RowProg(C.Structure):
    """cgat_rowprog"""
    _fields_ = [("n_ops", C.c_int32), ("op", RowProgOp * ROWPROG_MAX_OPS)]


PROTOTYPES = {
    "cgat_abi_version": (C.c_int, []),
    "cgat_last_error": (C.c_char_p, []),
    "cgat_prof_enable": (None, [C.c_int]),
    "cgat_prof_reset": (None, []),
    "cgat_prof_get": (C.c_int, [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_float)]),
    "cgat_prof_launches": (C.c_uint64, []),
    "cgat_plan_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32]),
    "cgat_plan_build": (C.c_int, [vp, C.c_int32, C.c_int32, vp, vp, vp, vp, vp, vp, vp, C.c_size_t, vp]),
    "cgat_csr_workspace_bytes": (C.c_size_t, [C.c_int32]),
    "cgat_csr_from_keys": (C.c_int, [vp, C.c_int32, C.c_int32, vp, vp, vp, C.c_size_t, vp]),
    "cgat_nodes_attention_saved_floats": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "cgat_nodes_attention_forward_workspace_bytes": (C.c_size_t, [C.POINTER(Plan), C.POINTER(AttnParams)]),
    "cgat_nodes_attention_backward_workspace_bytes": (C.c_size_t, [C.POINTER(Plan), C.POINTER(AttnParams)]),
    "cgat_nodes_attention_forward": (C.c_int, [C.POINTER(Plan), C.POINTER(AttnParams), vp, vp, vp, vp, vp,
                                               C.c_size_t, vp]),
    "cgat_nodes_attention_backward": (C.c_int, [C.POINTER(Plan), C.POINTER(AttnParams), vp, vp, vp, vp, vp, vp,
                                                C.POINTER(AttnGrads), vp, C.c_size_t, vp]),
    "cgat_debug_nodes_attention_signs": (C.c_int, [C.POINTER(Plan), C.POINTER(AttnParams), vp, vp, vp]),
    "cgat_hnet_saved_floats": (C.c_size_t, [C.c_int32, C.POINTER(HnetParams)]),
    "cgat_hnet_forward_workspace_bytes": (C.c_size_t, [C.c_int32, C.POINTER(HnetParams)]),
    "cgat_hnet_backward_workspace_bytes": (C.c_size_t, [C.c_int32, C.POINTER(HnetParams)]),
    "cgat_hnet_forward": (C.c_int, [C.c_int32, C.POINTER(HnetParams), vp, vp, vp, vp, vp, C.c_size_t, vp]),
    "cgat_hnet_backward": (C.c_int, [C.c_int32, C.POINTER(HnetParams), vp, vp, vp, vp, vp, vp, C.POINTER(HnetGrads),
                                     vp, C.c_size_t, vp]),
    "cgat_hnet_backward_side_workspace_bytes": (C.c_size_t, [C.c_int32, C.POINTER(HnetParams)]),
    "cgat_hnet_backward_overlapped": (C.c_int, [C.c_int32, C.POINTER(HnetParams), vp, vp, vp, vp, vp, vp,
                                                C.POINTER(HnetGrads), vp, C.c_size_t, vp, vp, C.c_size_t, vp]),
    "cgat_linear_backward_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    "cgat_linear_forward_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    "cgat_linear_forward": (C.c_int, [vp, C.c_int64, vp, C.c_int64, vp, vp, C.c_int64, C.c_int32, C.c_int32,
                                      C.c_int32, C.c_int32, vp, vp, C.c_size_t, vp]),
    "cgat_linear_backward": (C.c_int, [vp, C.c_int64, vp, C.c_int64, vp, C.c_int64, vp, C.c_int64, vp, vp, C.c_int64,
                                       C.c_int32, vp, C.c_int64, vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, vp,
                                       C.c_size_t, vp]),
    "cgat_segment_softmax_forward": (C.c_int, [vp, vp, vp, C.c_int32, C.c_int32, C.c_float, vp, vp]),
    "cgat_segment_softmax_backward": (C.c_int, [vp, vp, vp, vp, C.c_int32, C.c_int32, vp, vp, vp]),
    "cgat_segment_sum": (C.c_int, [vp, C.c_int64, vp, vp, C.c_int32, C.c_int32, vp, C.c_int64, vp]),
    "cgat_segment_attention_pool_forward": (C.c_int, [vp, C.c_int32, vp, vp, C.c_int64, vp, vp, C.c_int32, C.c_int32, C.c_float,
                                                      vp, vp, vp, vp, vp]),
    "cgat_segment_attention_pool_backward": (C.c_int, [vp, C.c_int32, vp, vp, C.c_int64, vp, vp, C.c_int32, C.c_int32, vp, vp, vp,
                                                       vp, vp, vp, vp, C.c_int64, vp, vp]),
    "cgat_mlp_chain_workspace_bytes": (C.c_size_t, [C.c_int32]),
    "cgat_mlp_chain": (C.c_int, [C.POINTER(ChainDesc), vp, C.c_size_t, vp]),
    "cgat_rowprog_run": (C.c_int, [C.POINTER(RowProg), vp, vp]),
    "cgat_dense_wgrad_batch_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32]),
    "cgat_dense_wgrad_batch": (C.c_int, [C.c_int32, vp, C.c_int64, vp, C.c_int64, vp, C.c_int64, vp, C.c_int32, vp,
                                          C.c_size_t, vp]),
    "cgat_gemm_workspace_bytes": (C.c_size_t, [C.POINTER(GemmDesc)]),
    "cgat_gemm": (C.c_int, [C.POINTER(GemmDesc), vp, C.c_size_t, vp]),
    "cgat_set_edge_storage": (None, [C.c_int32]),
    "cgat_get_edge_storage": (C.c_int32, []),
    "cgat_set_bilinear_mode": (None, [C.c_int32]),
    "cgat_get_bilinear_mode": (C.c_int32, []),
    "cgat_bilinear_rows_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "cgat_bilinear_rows": (C.c_int, [vp, C.c_int64, vp, C.c_int64, vp, vp, C.c_int64, vp, C.c_int64, C.c_int32,
                                     C.c_int32, C.c_int32, C.c_int32, vp, C.c_size_t, vp]),
    "cgat_edge_hidden_forward_workspace_bytes": (C.c_size_t, [C.POINTER(Plan), C.c_int32, C.c_int32, C.c_int32]),
    "cgat_edge_hidden_backward_workspace_bytes": (C.c_size_t, [C.POINTER(Plan), C.c_int32, C.c_int32, C.c_int32]),
    "cgat_edge_hidden_forward": (C.c_int, [C.POINTER(Plan), C.c_int32, C.c_int32, C.c_int32, vp, vp, vp, vp, vp, vp, vp,
                                           C.c_size_t, vp]),
    "cgat_edge_hidden_backward": (C.c_int, [C.POINTER(Plan), C.c_int32, C.c_int32, C.c_int32, vp, vp, vp, vp, vp, C.c_int32,
                                            vp, vp, vp, vp, vp, vp, C.c_size_t, vp]),
    "cgat_linear_backward_dact": (C.c_int, [vp, C.c_int64, vp, C.c_int64, vp, C.c_int64, vp, C.c_int64, vp, C.c_int64, vp,
                                            vp, C.c_int64, vp, C.c_int32, C.c_int32, C.c_int32, vp, C.c_size_t, vp]),
    "cgat_heads_linear_forward_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "cgat_heads_linear_forward": (C.c_int, [vp, C.c_int64, C.c_int64, vp, C.c_int64, C.c_int64, vp, C.c_int64, vp, C.c_int64,
                                            C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, vp, vp, C.c_size_t, vp]),
    "cgat_heads_linear_backward_dact_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "cgat_heads_linear_backward_dact": (C.c_int, [vp, C.c_int64, C.c_int64, vp, C.c_int64, C.c_int64, vp, C.c_int64, C.c_int64,
                                                  vp, C.c_int64, C.c_int64, vp, C.c_int64, C.c_int64, vp, vp, C.c_int64,
                                                  C.c_int64, vp, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, vp,
                                                  C.c_size_t, vp]),
    "cgat_mt_chunk_elems": (C.c_int32, []),
    "cgat_adamw_step": (C.c_int, [vp, vp, vp, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                  C.c_int64, vp]),
    "cgat_lamb_step": (C.c_int, [vp, vp, vp, C.c_int32, vp, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float,
                                 C.c_float, vp, vp]),
    "cgat_embedding_backward_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32]),
    "cgat_embedding_backward": (C.c_int, [vp, C.c_int64, vp, C.c_int64, C.c_int32, C.c_int32, vp, vp, C.c_size_t, vp]),
    "cgat_robust_loss": (C.c_int, [vp, vp, vp, C.c_int32, C.c_int32, vp, vp, vp, vp]),
    "cgat_collate_batch": (C.c_int, [C.POINTER(PackedDatasetStruct), vp, vp, vp, vp, C.c_int32, C.c_int64, C.c_int64,
                                     C.POINTER(CollatedStruct), vp]),
    "cgat_bilinear_dual_workspace_bytes": (C.c_size_t, [C.c_int32]),
    "cgat_bilinear_dual": (C.c_int, [vp, C.c_int64, vp, C.c_int64, vp, C.c_int64, vp, vp, C.c_int64, vp, C.c_int64, vp,
                                     C.c_int64, vp, C.c_int64, C.c_int32, vp, C.c_size_t, vp]),
    "cgat_bilinear_wgrad_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "cgat_bilinear_wgrad": (C.c_int, [vp, C.c_int64, vp, C.c_int64, vp, C.c_int64, vp, C.c_int32, C.c_int32,
                                      C.c_int32, C.c_int32, vp, C.c_size_t, vp]),
    "cgat_layernorm_tanh_forward": (C.c_int, [vp, vp, C.c_int32, C.c_int32, C.c_float, vp]),
    "cgat_layernorm_tanh_backward": (C.c_int, [vp, vp, vp, vp, C.c_int32, C.c_int32, C.c_float, vp]),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"cgat_amd: {LIB_PATH} not found -- build it with `bash cgat_amd/build_lib.sh` "
            "(or __graft_entry__.build()).  There is no CPU or PyTorch fallback for the HIP path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing: fail loudly
        fn.restype = res
        fn.argtypes = args
    got = lib.cgat_abi_version()
    if got != ABI_VERSION:
        raise ImportError(f"cgat_amd: libcgat_hip ABI {got} != binding ABI {ABI_VERSION}; rebuild the library")
    return lib


lib = _load()


class CgatHipError(RuntimeError):
    pass


def check(rc, what):
    if rc != 0:
        raise CgatHipError(f"{what} failed (code {rc}): {lib.cgat_last_error().decode(errors='replace')}")
