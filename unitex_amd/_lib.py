"""ctypes binding of libunitex_hip.so (the C ABI declared in include/unitex_hip.h).

This is the only place the Python host touches native code.  There is NO fallback: if the shared
library is missing or the device is not gfx950, construction raises.  torch is used purely for device
memory / streams (tensor.data_ptr(), torch.cuda.current_stream()).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libunitex_hip.so")

c_void_p, c_long, c_int, c_float = C.c_void_p, C.c_long, C.c_int, C.c_float


class GemmDesc(C.Structure):
    _fields_ = [("A", c_void_p), ("lda", c_long), ("B", c_void_p), ("ldb", c_long),
                ("A2", c_void_p), ("lda2", c_long), ("B2", c_void_p), ("ldb2", c_long),
                ("M", c_int), ("N", c_int), ("K", c_int), ("K2", c_int),
                ("lora_n_limit", c_int), ("lora_seg_n", c_int), ("alpha", c_float),
                ("bias", c_void_p), ("gelu_from", c_int), ("gate", c_void_p),
                ("res", c_void_p), ("ldres", c_long), ("C", c_void_p), ("ldc", c_long),
                ("n_split", c_int), ("C1", c_void_p), ("ldc1", c_long), ("ntn", c_int),
                ("conv_Hi", c_int), ("conv_Wi", c_int), ("conv_Wo", c_int), ("conv_cin_log2", c_int),
                ("conv_stride", c_int), ("conv_pad", c_int), ("conv_up", c_int), ("zero_page", c_void_p),
                ("a_scale", c_void_p), ("lds_a", c_long), ("b_scale", c_void_p), ("lds_b", c_long), ("mx8", c_int),
                ("qk_cols", c_int), ("qk_tok_off", c_int), ("qk_eps", c_float), ("qk_q_scale", c_float),
                ("qk_wq", c_void_p), ("qk_wk", c_void_p), ("qk_cos", c_void_p), ("qk_sin", c_void_p),
                ("qk_Qh", c_void_p), ("qk_Kh", c_void_p), ("qk_hs", c_long),
                ("sk_work", c_void_p), ("sk_work_bytes", C.c_size_t),
                ("q_out", c_void_p), ("ldq_out", c_long), ("qs_out", c_void_p), ("qs_out_rb", c_long), ("q_out_kt0", c_int)]


class GemvDesc(C.Structure):
    _fields_ = [("x", c_void_p), ("ldx", c_long), ("W", c_void_p), ("ldw", c_long),
                ("bias", c_void_p), ("y", c_void_p), ("ldy", c_long),
                ("M", c_int), ("N", c_int), ("K", c_int), ("silu_in", c_int), ("silu_out", c_int)]


class QkvPostDesc(C.Structure):
    _fields_ = [("qkv", c_void_p), ("ld", c_long), ("q_col", c_int), ("k_col", c_int), ("v_col", c_int),
                ("wq", c_void_p), ("wk", c_void_p), ("cosb", c_void_p), ("sinb", c_void_p),
                ("Qh", c_void_p), ("Kh", c_void_p), ("Vt", c_void_p),
                ("hs_qk", c_long), ("hs_v", c_long), ("S_pad", c_long),
                ("n_tok", c_int), ("tok_off", c_int), ("H", c_int), ("eps", c_float), ("q_scale", c_float),
                ("heads_per_group", c_int), ("gs_qk", c_long), ("gs_v", c_long), ("skip_qk", c_int),
                ("sub_heads", c_int), ("gs2_qk", c_long), ("gs2_v", c_long)]


class LnModDesc(C.Structure):
    _fields_ = [("x", c_void_p), ("ldx", c_long), ("shift", c_void_p), ("scale", c_void_p),
                ("y", c_void_p), ("ldy", c_long), ("n_tok", c_int), ("D", c_int), ("eps", c_float),
                ("q", c_void_p), ("ldq", c_long), ("qs", c_void_p), ("qs_row_blocks", c_long)]


class SchedDesc(C.Structure):
    _fields_ = [("x", c_void_p), ("v", c_void_p), ("cond", c_void_p),
                ("n_noise_elems", c_long), ("n_total_elems", c_long), ("dsigma", c_float)]


class BackprojectDesc(C.Structure):
    _fields_ = [("rast2d", c_void_p), ("verts", c_void_p), ("faces", c_void_p), ("fnormal", c_void_p),
                ("vndc", c_void_p), ("dirs", c_void_p), ("images", c_void_p),
                ("color", c_void_p), ("rayvis", c_void_p), ("alphaok", c_void_p),
                ("T_h", c_int), ("T_w", c_int), ("V", c_int), ("n_views", c_int), ("H", c_int), ("W", c_int),
                ("view_begin", c_int), ("view_count", c_int), ("cos_thresh", c_float), ("two_sqrt3", c_float)]


class KnnDesc(C.Structure):
    _fields_ = [("src_pos", c_void_p), ("src_attr", c_void_p), ("src_nrm", c_void_p), ("src_mask", c_void_p), ("N", c_long),
                ("dst_pos", c_void_p), ("dst_nrm", c_void_p), ("dst_mask", c_void_p), ("M", c_long),
                ("k", c_int), ("C", c_int), ("mode", c_int),
                ("out_attr", c_void_p), ("out_idx", c_void_p), ("out_d2", c_void_p)]


class DitLinear(C.Structure):
    _fields_ = [("w", c_void_p), ("b", c_void_p), ("lora_A", c_void_p), ("lora_B", c_void_p), ("lora_alpha", c_float), ("lora_rp", c_int), ("lora_nseg", c_int),
                ("q", c_void_p), ("s", c_void_p), ("lds_s", c_long), ("sp", c_void_p), ("sp_row_blocks", c_int)]


class DitDoubleBlock(C.Structure):
    _fields_ = [(n, DitLinear) for n in ("qkv_x", "qkv_c", "out_x", "out_c", "ff1_x", "ff2_x", "ff1_c", "ff2_c")] + \
               [("nq", c_void_p), ("nk", c_void_p), ("naq", c_void_p), ("nak", c_void_p), ("mod_x", c_int), ("mod_c", c_int)]


class DitSingleBlock(C.Structure):
    _fields_ = [("qkvm", DitLinear), ("out", DitLinear), ("nq", c_void_p), ("nk", c_void_p), ("mod", c_int)]


class DitWeights(C.Structure):
    _fields_ = [(n, DitLinear) for n in ("x_embedder", "context_embedder", "proj_out", "t_lin1", "t_lin2", "g_lin1", "g_lin2", "p_lin1", "p_lin2", "mod")] + \
               [("dbl", C.POINTER(DitDoubleBlock)), ("sgl", C.POINTER(DitSingleBlock)), ("mod_out", c_int), ("n_mod", c_int)]


class DitConfig(C.Structure):
    _fields_ = [(n, c_int) for n in ("num_heads", "num_double", "num_single", "in_channels", "joint_dim", "pooled_dim", "mlp_ratio", "guidance_embeds",
                                     "S_txt", "S_img", "n_out")] + [("key_bias_log2", c_float), ("key_bias_period", c_int), ("two_streams", c_int), ("n_cus", c_int),
                                                                     ("lora_rank_padded", c_int), ("fp8", c_int), ("fp8_fuse_quant", c_int)]


class DitWorkspace(C.Structure):
    _fields_ = [(n, c_void_p) for n in ("lat", "enc", "pooled", "tproj", "gproj", "e1", "e_t", "e_g", "e_p", "temb", "mod", "h", "xn", "qkv", "cat", "attn", "out",
                                        "cos", "sin", "Qh", "Kh", "Vt", "T", "Tc")] + \
               [("sk_work", c_void_p), ("sk_work_bytes", C.c_size_t), ("attn_work", c_void_p), ("attn_work_bytes", C.c_size_t),
                ("aq", c_void_p), ("as_rm", c_void_p), ("asp", c_void_p), ("asp_row_blocks", c_int), ("aq2", c_void_p), ("asp2", c_void_p), ("asp2_row_blocks", c_int)]


ABI_STRUCTS = [GemmDesc, GemvDesc, QkvPostDesc, LnModDesc, SchedDesc, BackprojectDesc, KnnDesc, DitLinear, DitDoubleBlock, DitSingleBlock, DitWeights, DitConfig,
               DitWorkspace]

# every symbol include/unitex_hip.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "utx_version": (c_int, []),
    "utx_init": (c_int, [c_int, C.POINTER(c_void_p)]),
    "utx_free": (None, [c_void_p]),
    "utx_last_error": (C.c_char_p, [c_void_p]),
    "utx_abi_sizes": (c_int, [C.POINTER(c_int), c_int]),
    "utx_plan_create": (c_int, [c_void_p, C.POINTER(c_void_p)]),
    "utx_plan_free": (None, [c_void_p]),
    "utx_plan_size": (c_int, [c_void_p]),
    "utx_plan_add_gemm": (c_int, [c_void_p, C.POINTER(GemmDesc)]),
    "utx_plan_add_gemv": (c_int, [c_void_p, C.POINTER(GemvDesc)]),
    "utx_plan_add_ln_mod": (c_int, [c_void_p, C.POINTER(LnModDesc)]),
    "utx_plan_add_qkv_post": (c_int, [c_void_p, C.POINTER(QkvPostDesc)]),
    "utx_plan_add_attn": (c_int, [c_void_p] * 5 + [c_long] * 7 + [c_int, c_int, c_int, c_float, c_float, c_int, c_void_p, C.c_size_t]),
    "utx_plan_add_quant_mx8": (c_int, [c_void_p, c_void_p, c_long, c_void_p, c_long, c_void_p, c_long, c_int, c_int, c_int]),
    "utx_plan_add_add3": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int]),
    "utx_plan_add_quant_vt_mx8": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int]),
    "utx_plan_add_attn_fp8": (c_int, [c_void_p] * 8 + [c_long, c_int, c_int, c_int, c_int, c_float, c_int]),
    "utx_plan_add_attn_fp8_ws": (c_int, [c_void_p] * 8 + [c_long, c_int, c_int, c_int, c_int, c_float, c_int, c_void_p, C.c_size_t]),
    "utx_plan_fork": (c_int, [c_void_p]),
    "utx_plan_main": (c_int, [c_void_p]),
    "utx_plan_join": (c_int, [c_void_p]),
    "utx_plan_run": (c_int, [c_void_p, c_void_p, C.POINTER(c_int)]),
    "utx_plan_run_range": (c_int, [c_void_p, c_int, c_int, c_void_p, C.POINTER(c_int)]),
    "utx_plan_assign_sk": (c_int, [c_void_p, c_void_p, C.c_size_t, c_int]),
    "utx_plan_entry": (c_int, [c_void_p, c_int, C.POINTER(c_int), C.POINTER(c_int), c_void_p, C.c_size_t]),
    "utx_dit_load": (c_int, [c_void_p, C.POINTER(DitConfig), C.POINTER(DitWeights), C.POINTER(DitWorkspace), C.POINTER(c_void_p)]),
    "utx_dit_step": (c_int, [c_void_p, c_void_p, C.POINTER(c_int)]),
    "utx_mesh_decimate_qem": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, C.c_double, c_void_p, c_void_p, C.POINTER(c_int), C.POINTER(c_int)]),
    "utx_set_option": (c_int, [C.c_char_p, c_int]),
    "utx_get_option": (c_int, [C.c_char_p, C.POINTER(c_int)]),
    "utx_is_ablation_build": (c_int, []),
    "utx_attn_fwd_bf16": (c_int, [c_void_p] * 5 + [c_long] * 7 + [c_int, c_int, c_float, c_void_p]),
    "utx_attn_fwd_bf16_kb": (c_int, [c_void_p] * 5 + [c_long] * 7 + [c_int, c_int, c_float, c_float, c_int, c_void_p]),
    "utx_attn_fwd_bf16_kbq": (c_int, [c_void_p] * 5 + [c_long] * 7 + [c_int, c_int, c_int, c_float, c_float, c_int, c_void_p]),
    "utx_attn_fwd_bf16_ws": (c_int, [c_void_p] * 5 + [c_long] * 7 + [c_int, c_int, c_int, c_float, c_float, c_int, c_void_p, C.c_size_t, c_void_p]),
    "utx_attn_fwd_bf16_blk": (c_int, [c_void_p] * 5 + [c_long] * 7 + [c_int, c_int, c_int, c_float, c_float, c_int, c_void_p, C.c_size_t, c_int, c_long, c_long, c_long,
                                      c_void_p]),
    "utx_attn_workspace_bytes": (C.c_size_t, [c_void_p, c_int, c_int, c_int]),
    "utx_quant_vt_mx8": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "utx_attn_fwd_fp8": (c_int, [c_void_p] * 8 + [c_long, c_int, c_int, c_int, c_int, c_float, c_int, c_void_p]),
    "utx_attn_fwd_fp8_ws": (c_int, [c_void_p] * 8 + [c_long, c_int, c_int, c_int, c_int, c_float, c_int, c_void_p, C.c_size_t, c_void_p]),
    "utx_attn_plan": (c_int, [c_int, c_int, c_int, c_int, c_void_p]),
    "utx_gemm_bf16": (c_int, [c_void_p, C.POINTER(GemmDesc), c_void_p]),
    "utx_gemm_streamk_workspace_bytes": (C.c_size_t, [c_void_p]),
    "utx_gemm_plan": (c_int, [C.POINTER(GemmDesc), c_int, C.POINTER(c_int * 4)]),
    "utx_quant_mx8": (c_int, [c_void_p, c_void_p, c_long, c_void_p, c_long, c_void_p, c_long, c_int, c_int, c_void_p]),
    "utx_quant_mx8_packed": (c_int, [c_void_p, c_void_p, c_long, c_void_p, c_long, c_void_p, c_long, c_int, c_int, c_void_p]),
    "utx_gemv_bf16": (c_int, [c_void_p, C.POINTER(GemvDesc), c_void_p]),
    "utx_group_norm_workspace_bytes": (c_long, []),
    "utx_group_norm": (c_int, [c_void_p, c_void_p, c_long, c_int, c_void_p, c_void_p, c_float, c_int, c_void_p, c_void_p, c_void_p]),
    "utx_softmax_rows": (c_int, [c_void_p, c_void_p, c_long, c_long, c_int, c_void_p]),
    "utx_conv3x3_thin": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "utx_qkv_post": (c_int, [c_void_p, C.POINTER(QkvPostDesc), c_void_p]),
    "utx_sp_unpack_qkv": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "utx_sp_unpack_qkv_dedup": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "utx_sp_unpack_o": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_long, c_void_p]),
    "utx_sp_unpack_o_cols": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_long, c_long, c_void_p]),
    "utx_ln_mod": (c_int, [c_void_p, C.POINTER(LnModDesc), c_void_p]),
    "utx_sched_step": (c_int, [c_void_p, C.POINTER(SchedDesc), c_void_p]),
    # geometry
    "utx_transform_points": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "utx_rasterize_workspace_bytes": (c_long, [c_int, c_int, c_int]),
    "utx_rasterize": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "utx_interpolate": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_long, c_void_p, c_void_p]),
    "utx_condition_shade": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, C.POINTER(c_float), c_long, c_void_p, c_void_p, c_void_p, c_void_p]),
    "utx_face_normals": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "utx_texture_shade": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, C.POINTER(c_float), c_long, c_void_p, c_void_p]),
    "utx_bvh_build": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, C.POINTER(c_void_p), c_void_p]),
    "utx_bvh_workspace_bytes": (C.c_size_t, [c_int]),
    "utx_bvh_build_ws": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, C.c_size_t, C.POINTER(c_void_p), c_void_p]),
    "utx_bvh_free": (None, [c_void_p]),
    "utx_bvh_arrays": (c_int, [c_void_p, C.POINTER(c_void_p), C.POINTER(c_void_p), C.POINTER(c_void_p), C.POINTER(c_void_p)]),
    "utx_bvh_trace": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_void_p, c_void_p]),
    "utx_bvh_trace_count": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_void_p, c_void_p, c_void_p]),
    "utx_bvh_depth": (c_int, [c_void_p]),
    "utx_backproject": (c_int, [c_void_p, C.POINTER(BackprojectDesc), c_void_p, c_void_p]),
    "utx_dilate_visibility": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "utx_composite": (c_int, [c_void_p, c_void_p, c_void_p, C.POINTER(c_int), c_int, c_long, c_void_p, c_void_p, c_void_p]),
    "utx_seam_mask": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "utx_view_visibility": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_float, c_int,
                                    c_void_p, c_void_p, c_void_p, c_void_p]),
    "utx_knn_workspace_bytes": (c_long, [c_long]),
    "utx_knn": (c_int, [c_void_p, C.POINTER(KnnDesc), c_void_p, c_long, c_void_p]),
    "utx_nn_fill_workspace_bytes": (c_long, [c_long]),
    "utx_nn_fill": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_void_p, c_void_p, c_void_p, c_long, c_void_p]),
    "utx_lens_blur_seam": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, C.POINTER(c_float), c_void_p, c_void_p]),
    "utx_pull_push_workspace_bytes": (c_long, [c_int, c_int]),
    "utx_pull_push": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "utx_chart_flood": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "utx_to_u8": (c_int, [c_void_p, c_void_p, c_long, c_long, c_int, c_void_p, c_void_p]),
}

_lib = None


def use_ablation_library():
    """tools/ only: bind libunitex_hip_ablate.so (built by `python unitex_amd/csrc/build.py --ablate`), the build that still
    contains the wrong-result timing ablations.  Must be called before the first op; never called by the product."""
    global LIB_PATH
    if _lib is not None:
        raise RuntimeError("the library is already loaded")
    LIB_PATH = os.path.join(_HERE, "lib", "libunitex_hip_ablate.so")


def load_library():
    """dlopen the C-ABI library and bind prototypes.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libunitex_hip.so not found at %s -- run `python unitex_amd/csrc/build.py` "
            "(or __graft_entry__.build()). There is no CPU/PyTorch fallback by design." % LIB_PATH)
    # ONE HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so (SONAME libamdhip64.so.7, requested by libtorch_hip.so under the
    # unversioned name); this library asks for libamdhip64.so.7.  With torch loaded first the dynamic loader resolves our request to torch's copy (SONAME
    # match); the other way round it maps the system copy for us and torch's bundled one afterwards -- two runtimes, and the second one to touch the
    # device fails (seen as utx_init -> -5 when __graft_entry__.build() loaded this library before smoke() imported torch).  torch owns the device
    # memory and the streams here anyway: import it first.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


OPTION_NAMES = ["UTX_ATTN_GLDS", "UTX_ATTN_FAST", "UTX_ATTN_Q64", "UTX_ATTN_TPB", "UTX_ATTN_TAILSPLIT",
                "UTX_GEMM_GROUP_M", "UTX_GEMM_TILE", "UTX_GEMM_TAILSPLIT", "UTX_GEMM_PERS_GRID", "UTX_GEMM_PERS_SCHED",
                "UTX_GEMM_STREAMK", "UTX_BVH_STACK_WALK", "UTX_BVH_PACKET", "UTX_ATTN_PEEL", "UTX_ATTN8_PEEL", "UTX_NN_GRID", "UTX_GEMM_FASTK"]


def set_option(name, value):
    """result-preserving launch option (include/unitex_hip.h `utx_set_option`); raises on unknown / ablation-only names."""
    rc = load_library().utx_set_option(name.encode(), int(value))
    if rc != 0:
        raise ValueError("utx_set_option(%s) -> %d (%s)" % (name, rc, "ablation-only switch: not in this library" if rc == -7 else "unknown option"))


def get_options():
    """{name: value} of every launch option as the library currently holds them (what bench.py reports)."""
    lib, out = load_library(), {}
    for n in OPTION_NAMES:
        v = c_int()
        if lib.utx_get_option(n.encode(), C.byref(v)) == 0:
            out[n] = v.value
    return out


def check_abi():
    """Compare ctypes struct sizes with the library's sizeof() -- runs without a GPU."""
    lib = load_library()
    n = len(ABI_STRUCTS)
    out = (c_int * n)()
    m = lib.utx_abi_sizes(out, n)
    if m != n:
        raise RuntimeError("ABI mismatch: library reports %d descriptor structs, binding has %d" % (m, n))
    for i, st in enumerate(ABI_STRUCTS):
        if C.sizeof(st) != out[i]:
            raise RuntimeError("ABI mismatch for %s: ctypes %d vs C %d" % (st.__name__, C.sizeof(st), out[i]))
    return True


class Context:
    """One utx_ctx per device.  All ops are stream-ordered on torch's current stream."""

    def __init__(self, device=0):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("unitex_amd requires an MI355X (gfx950) GPU; no HIP device is visible")
        self.lib = load_library()
        check_abi()
        self.device = int(device)
        h = c_void_p()
        rc = self.lib.utx_init(self.device, C.byref(h))
        if rc != 0:
            raise RuntimeError("utx_init(device=%d) failed with code %d (need a gfx950 device)" % (device, rc))
        self.handle = h
        self._torch = torch

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.utx_free(self.handle)
                self.handle = None
        except Exception:
            pass

    def stream(self):
        return c_void_p(self._torch.cuda.current_stream(self.device).cuda_stream)

    def check(self, rc):
        if rc != 0:
            raise RuntimeError(self.lib.utx_last_error(self.handle).decode())


def ptr(t):
    return c_void_p(t.data_ptr()) if t is not None else c_void_p(0)
