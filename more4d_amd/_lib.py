"""ctypes binding of libmore4d_hip.so (include/more4d_hip.h).  There is NO fallback: if the library
is missing or a call fails the caller gets an exception — the HIP kernels are the product path."""
import ctypes
import os
from ctypes import (POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_int64, c_void_p)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libmore4d_hip.so")
if os.environ.get("M4D_LIB"):      # tools only: "abl" = the -DM4D_ABLATIONS build (wrong results by design), else a side build to A/B
    LIB_PATH = os.path.join(_HERE, "lib", "libmore4d_hip_%s.so" % os.environ["M4D_LIB"])
ABLATION_BUILD = bool(os.environ.get("M4D_LIB"))      # anything but the shipping library: bench.py refuses it

M4D_F32, M4D_BF16 = 0, 1
EPI_STORE, EPI_GELU_TANH, EPI_GELU_ERF, EPI_SILU, EPI_RESID_GATE, EPI_STORE_F32 = range(6)
MAX_KV_SEGS = 8


class KvSegs(Structure):
    _fields_ = [("k", c_void_p * MAX_KV_SEGS), ("vt", c_void_p * MAX_KV_SEGS),
                ("k_bs", c_int64 * MAX_KV_SEGS), ("k_ls", c_int64 * MAX_KV_SEGS),
                ("vt_bs", c_int64 * MAX_KV_SEGS), ("vt_ls", c_int64 * MAX_KV_SEGS),
                ("len", c_int64 * MAX_KV_SEGS), ("nseg", c_int32), ("new_softmax", c_int32)]


class AttnBwdArgs(Structure):
    _fields_ = ([(n, c_void_p) for n in ("q", "k", "v", "o", "d_o", "qt", "kt", "dot", "lse", "delta", "dq", "dk", "dv")] +
                [(n, c_int64) for n in ("q_bs", "q_ls", "k_bs", "k_ls", "v_bs", "v_ls", "o_bs", "o_ls", "do_bs", "do_ls",
                                        "qt_bs", "qt_ls", "kt_bs", "kt_ls", "dot_bs", "dot_ls",
                                        "dq_bs", "dq_ls", "dk_bs", "dk_ls", "dv_bs", "dv_ls", "Lq", "Lk", "Lk_rows")] +
                [(n, c_int32) for n in ("B", "heads", "head_dim", "accumulate_dq", "accumulate_dkv")] +
                [("scale", c_float), ("ws", c_void_p), ("ws_elems", c_int64)])


# name -> (restype, argtypes); mirrors include/more4d_hip.h one to one
SIGNATURES = {
    "m4d_version": (c_int, []),
    "m4d_source_hash": (c_char_p, []),
    "m4d_launch_count": (c_int64, [c_int, c_int]),
    "m4d_last_error": (c_char_p, []),
    "m4d_gemm_bt": (c_int, [c_int, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int, c_void_p, c_int64,
                            c_int64, c_int64, c_int64, c_int, c_void_p, c_int64, c_int64, c_void_p]),
    "m4d_gemm_bt_workspace_bytes": (c_int64, [c_int, c_int64, c_int64, c_int64]),
    "m4d_gemm_bt_ws": (c_int, [c_int, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int, c_void_p, c_int64,
                               c_int64, c_int64, c_int64, c_int, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p]),
    "m4d_pack_frag_elems": (c_int64, [c_int64, c_int64]),
    "m4d_pack_frag": (c_int, [c_int, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_void_p]),
    "m4d_gemm_bt_packed": (c_int, [c_int, c_void_p, c_int64, c_void_p, c_int64, c_int, c_void_p, c_int, c_void_p, c_int64,
                                   c_int64, c_int64, c_int64, c_int, c_void_p, c_int64, c_int64, c_void_p]),
    "m4d_ln_modulate": (c_int, [c_int, c_void_p, c_int, c_void_p, c_int64, c_int, c_int64, c_void_p, c_void_p,
                                c_int64, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_int64, c_int64,
                                c_void_p]),
    "m4d_ln_modulate_g": (c_int, [c_int, c_void_p, c_int, c_void_p, c_int64, c_int, c_int64, c_void_p, c_void_p,
                                  c_int64, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_int64, c_int64, c_int64,
                                  c_void_p]),
    "m4d_rmsnorm_rope": (c_int, [c_int, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int, c_int,
                                 c_float, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p]),
    "m4d_attention": (c_int, [c_int, c_void_p, c_int64, c_int64, POINTER(KvSegs), c_void_p, c_int64, c_int64,
                              c_int, c_int64, c_int, c_int, c_float, c_int, c_void_p]),
    "m4d_attention_lse": (c_int, [c_int, c_void_p, c_int64, c_int64, POINTER(KvSegs), c_void_p, c_int64, c_int64,
                                  c_int, c_int64, c_int, c_int, c_float, c_int, c_void_p, c_void_p]),
    "m4d_attention_bwd": (c_int, [c_int, POINTER(AttnBwdArgs), c_void_p]),
    "m4d_transpose": (c_int, [c_int, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int64, c_void_p]),
    "m4d_colsum": (c_int, [c_int, c_void_p, c_int64, c_int, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int64,
                           c_void_p]),
    "m4d_scale_cast": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int, c_void_p, c_int64, c_int64, c_void_p]),
    "m4d_resid_gate": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p]),
    "m4d_add": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "m4d_act_bwd": (c_int, [c_int, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "m4d_ln_modulate_bwd": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int64, c_int, c_void_p, c_int64,
                                    c_void_p, c_float, c_void_p, c_void_p, c_int64, c_void_p]),
    "m4d_attn_merge": (c_int, [c_int, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int, c_int64,
                               c_int, c_int, c_void_p]),
    "m4d_guidance_bwd": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int64, c_int, c_void_p, c_void_p, c_int64, c_float,
                                 c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p]),
    "m4d_guidance_bwd_m": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int64, c_int, c_void_p, c_void_p, c_int64, c_int64, c_float,
                                   c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p]),
    "m4d_rmsnorm_rope_bwd": (c_int, [c_int, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_void_p,
                                     c_void_p, c_void_p, c_int64, c_int, c_int, c_float, c_void_p, c_void_p, c_int64,
                                     c_int64, c_int64, c_void_p]),
    "m4d_sumsq": (c_int, [c_int, c_void_p, c_int64, c_void_p, c_void_p]),
    "m4d_adamw": (c_int, [c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int64, c_float, c_float, c_float,
                          c_float, c_float, c_int64, c_void_p, c_void_p]),
    "m4d_patchify": (c_int, [c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int,
                             c_int, c_int, c_int, c_int, c_void_p]),
    "m4d_unpatchify": (c_int, [c_void_p, c_int64, c_int64, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int,
                               c_int, c_int, c_int, c_void_p]),
    "m4d_cfg_euler": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_float, c_float, c_int, c_void_p]),
    "m4d_unary": (c_int, [c_int, c_void_p, c_int, c_void_p, c_int64, c_int, c_void_p]),
    "m4d_add_bcast": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    "m4d_axpby": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_void_p]),
    "m4d_lincomb": (c_int, [c_void_p, c_float, c_void_p, c_float, c_void_p, c_float, c_void_p, c_float, c_void_p, c_int64,
                            c_void_p]),
    "m4d_rel_l1": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "m4d_pad_transpose": (c_int, [c_int, c_void_p, c_int64] + [c_int] * 9 + [c_void_p, c_int64, c_void_p]),
    "m4d_gemm_bt_batched": (c_int, [c_int, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64,
                                    c_int64, c_int64, c_int, c_int, c_void_p]),
    "m4d_wgrad_reduce": (c_int, [c_void_p, c_void_p] + [c_int] * 8 + [c_void_p]),
    "m4d_gemm_bt_taps": (c_int, [c_int, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int64, c_int,
                                 c_int, c_int, c_int64, c_int64, c_void_p]),
    "m4d_wgrad_reduce_taps": (c_int, [c_void_p, c_void_p] + [c_int] * 6 + [c_void_p]),
    "m4d_rmsnorm_silu_cl_bwd": (c_int, [c_int, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int,
                                        c_int, c_void_p]),
    "m4d_softmax_rows_bwd": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_float, c_void_p]),
    "m4d_upsample2x_cl": (c_int, [c_int, c_void_p, c_void_p] + [c_int] * 6 + [c_void_p]),
    "m4d_groupnorm_cl_bwd_workspace": (c_int64, [c_int, c_int64, c_int]),
    "m4d_groupnorm_cl_bwd": (c_int, [c_int] + [c_void_p] * 8 + [c_int64, c_int, c_int64, c_int, c_int, c_float, c_int, c_void_p]),
    "m4d_minmax": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p]),
    "m4d_backproject": (c_int, [c_void_p, c_int, c_int, c_float, c_float, c_void_p, c_void_p, c_void_p]),
    "m4d_depth_control": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "m4d_flow_recover": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_int, c_void_p]),
    "m4d_bilinear_cl": (c_int, [c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "m4d_conv_cl": (c_int, [c_int, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64] +
                    [c_int] * 19 + [c_void_p]),
    "m4d_conv_tiled_weight_bytes": (c_int64, [c_int, c_int, c_int]),
    "m4d_conv_pack_weights": (c_int, [c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "m4d_conv_cl_tw": (c_int, [c_int, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64] +
                       [c_int] * 19 + [c_void_p]),
    "m4d_conv_cl_planar_tw": (c_int, [c_int, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64] +
                              [c_int] * 7 + [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p]),
    "m4d_rmsnorm_silu_cl": (c_int, [c_int, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int, c_void_p]),
    "m4d_rmsnorm_silu_cl_planar": (c_int, [c_int, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int, c_void_p]),
    "m4d_conv_cl_planar": (c_int, [c_int, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64] +
                           [c_int] * 7 + [c_void_p]),
    "m4d_conv_cl_planar_gnstats_blocks": (c_int, [c_int, c_int]),
    "m4d_conv_cl_planar_gnstats": (c_int, [c_int, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64] +
                                   [c_int] * 7 + [c_void_p, c_void_p]),
    "m4d_groupnorm_cl_planar_apply": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int,
                                              c_int, c_float, c_int, c_int, c_int64, c_int64, c_void_p]),
    "m4d_conv_cl_planar_norm": (c_int, [c_int, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64] +
                                [c_int] * 7 + [c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "m4d_groupnorm_cl_workspace": (c_int64, [c_int, c_int64, c_int]),
    "m4d_groupnorm_cl": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int, c_int64, c_int,
                                 c_int, c_float, c_int, c_void_p]),
    "m4d_groupnorm_cl_planar": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int, c_int64, c_int,
                                        c_int, c_float, c_int, c_int, c_int64, c_int64, c_void_p]),
    "m4d_softmax_rows": (c_int, [c_int, c_void_p, c_int64, c_int, c_void_p, c_int64, c_int64, c_int, c_int, c_float,
                                 c_void_p]),
    "m4d_ncthw_to_cl": (c_int, [c_int, c_void_p, c_int, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_int, c_float,
                                c_float, c_void_p, c_void_p, c_void_p]),
    "m4d_cl_to_ncthw": (c_int, [c_int, c_void_p, c_int64, c_int, c_void_p, c_int, c_int, c_int, c_int, c_float, c_float,
                                c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
}

_lib = None


class More4DHipError(RuntimeError):
    pass


def load():
    """Load the library once; raise loudly if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise More4DHipError(
            f"{LIB_PATH} not found: build it with `python -m more4d_amd.build` "
            "(the HIP extension is the only compute path; there is no CPU fallback)")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        if ABLATION_BUILD and not hasattr(lib, name):
            continue             # side builds for same-box A/B (M4D_LIB=<tag>) may predate an entry point; the shipping library may not
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    if not ABLATION_BUILD:
        # the shipped binary must be what the tracked sources compile to: the hash of csrc/ + include/ is compiled in (api.cpp)
        from .build import source_hash
        got, want = lib.m4d_source_hash().decode(), source_hash()
        if got != want:
            raise More4DHipError(
                f"{LIB_PATH} was built from other sources (library {got[:16]}, tree {want[:16]}): "
                "rebuild with `python -m more4d_amd.build`")
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().m4d_last_error()
        raise More4DHipError(f"{what} failed (rc={rc}): {msg.decode() if msg else '?'}")
