"""ctypes binding of libhsimae_hip.so (include/hsimae_hip.h).

The shared library is the product's only compute path: if it is missing this module raises at
import of the symbol table, and every wrapper raises RuntimeError on a non-zero return code.
There is no CPU or PyTorch fallback.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# HSIMAE_LIB: another build of the same library (A/B runs of compile-time variants: hsimae_amd/variants/*.so)
LIB_PATH = os.environ.get("HSIMAE_LIB") or os.path.join(HERE, "libhsimae_hip.so")

i32, i64, f32, vp = C.c_int32, C.c_int64, C.c_float, C.c_void_p


class Config(C.Structure):
    _fields_ = [("bands", i32), ("embed_dim", i32), ("depth", i32), ("s_depth", i32), ("num_heads", i32),
                ("dec_dim", i32), ("dec_depth", i32), ("dec_heads", i32), ("hidden", i32), ("dec_hidden", i32),
                ("norm_pix_loss", i32), ("precision", i32)]


class IO(C.Structure):
    _fields_ = [("x", vp), ("sn", i64), ("sb", i64), ("sh", i64), ("sw", i64),
                ("N", i32), ("len_t", i32), ("len_l", i32),
                ("noise1", vp), ("noise2", vp), ("params", vp), ("wpk", vp),
                ("workspace", vp), ("workspace_bytes", i64),
                ("grad_scale", f32), ("want_recons", i32),
                ("loss", vp), ("pred_img", vp), ("mask_img", vp), ("mask", vp),
                ("ids_keep", vp), ("ids_restore", vp), ("latent", vp), ("pred", vp), ("drop_scale", vp),
                ("bucket_stream", vp), ("det_acc", vp)]


class MaskParams(C.Structure):
    _fields_ = [("noise1", vp), ("noise2", vp), ("N", i32), ("T", i32), ("L", i32), ("len_t", i32), ("len_l", i32),
                ("ids_keep", vp), ("ids_restore", vp), ("mask", vp)]


class PatchParams(C.Structure):
    _fields_ = [("x", vp), ("sn", i64), ("sb", i64), ("sh", i64), ("sw", i64), ("N", i32), ("T", i32), ("K", i32),
                ("ids_keep", vp), ("out", vp), ("pos_ids", vp)]


class GemmParams(C.Structure):
    _fields_ = [("A", vp), ("lda", i32), ("M", i32), ("N", i32), ("K", i32), ("n_valid", i32),
                ("W", vp), ("W2", vp), ("bias", vp), ("bias2", vp), ("gamma", vp), ("beta", vp),
                ("stats", vp), ("u_out", vp), ("ldu", i32), ("out", vp), ("ldo", i32),
                ("res", vp), ("res2", vp), ("ldr", i32), ("pos", vp), ("ids", vp), ("ldpos", i32),
                ("h13", vp), ("ldh", i32), ("hoff", i32),
                ("lnx", vp), ("dgamma", vp), ("dbeta", vp), ("accumulate", i32),
                ("a_rowscale", vp), ("out_rowscale", vp),
                ("prec", i32), ("W8", vp), ("S8", vp), ("W8b", vp), ("S8b", vp), ("ln_width", i32),
                ("det_base", vp), ("det_acc", vp)]


class PackDesc(C.Structure):
    _fields_ = [("src", vp), ("rows", i32), ("cols", i32), ("transpose", i32), ("n_off", i32), ("k_off", i32),
                ("KS", i32), ("dst", vp), ("fp8", i32), ("scales", vp)]


class AttnParams(C.Structure):
    _fields_ = [("qkv", vp), ("ld", i32), ("d", i32), ("heads", i32), ("hd", i32), ("Ts", i32), ("nsamples", i32),
                ("mode", i32), ("len_l", i32), ("o", vp), ("ldo", i32), ("lse", vp), ("dout", vp), ("lddo", i32),
                ("dqkv", vp), ("kv_off", i32)]


class MlpWeights(C.Structure):
    _fields_ = [("n2w", vp), ("n2b", vp), ("w1b", vp), ("w3b", vp), ("w2b", vp),
                ("w1", vp), ("w3", vp), ("w2", vp), ("w2T", vp), ("w13T", vp), ("hidden", i32)]


class DecBlockWeights(C.Structure):
    _fields_ = [(n, vp) for n in ("n1w", "n1b", "bqkv", "pb", "n2w", "n2b", "w1b", "w3b", "w2b", "qkv", "p", "w1", "w3", "w2", "w2T",
                                  "qf", "kf", "vf", "pf", "w1f", "w3f")] + [("hidden", i32)]


class DecBlockGrads(C.Structure):
    _fields_ = [(n, vp) for n in ("n1w", "n1b", "qw", "qb", "kw", "kb", "vw", "vb", "pw", "pb", "n2w", "n2b", "w1w", "w1b", "w2w",
                                  "w2b", "w3w", "w3b")]


class WgradTask(C.Structure):
    _fields_ = [("dO", vp), ("dO_f32", i32), ("ldo", i32), ("A", vp), ("lda", i32), ("N", i32), ("K", i32),
                ("dW", vp), ("ldw", i32), ("db", vp), ("dO_rowscale", vp), ("dO_plane_rows", i32), ("A_plane_rows", i32)]


class WgradParams(C.Structure):
    _fields_ = [("t", WgradTask * 16), ("ntasks", i32), ("M", i32), ("msplit", i32), ("det_base", vp), ("det_acc", vp), ("slab", vp)]


class LnBwdParams(C.Structure):
    _fields_ = [("du", vp), ("x", vp), ("stats", vp), ("gamma", vp), ("dres", vp), ("dx", vp), ("accumulate", i32),
                ("dgamma", vp), ("dbeta", vp), ("M", i32), ("d", i32), ("ld", i32),
                ("det_base", vp), ("det_acc", vp)]


class AssembleParams(C.Structure):
    _fields_ = [("y", vp), ("N", i32), ("K", i32), ("TL", i32), ("Dd", i32), ("ids_restore", vp), ("pos", vp),
                ("yfull", vp), ("dyfull", vp), ("dy", vp), ("ld", i32)]


class LossParams(C.Structure):
    _fields_ = [("x", vp), ("sn", i64), ("sb", i64), ("sh", i64), ("sw", i64), ("N", i32), ("T", i32),
                ("pred", vp), ("mask", vp), ("norm_pix", i32), ("inv_scale", f32), ("partial", vp), ("loss", vp),
                ("sum_mask", f32), ("dpred", vp), ("pred_img", vp), ("mask_img", vp)]


class CubeParams(C.Structure):
    _fields_ = [("scenes", vp), ("scene_f64", i32), ("scene_off", vp), ("scene_w", vp), ("bands", i32),
                ("cut", vp), ("index", vp), ("flips", vp), ("N", i32),
                ("out", vp), ("sn", i64), ("sb", i64), ("sh", i64), ("sw", i64)]


class BuildInfo(C.Structure):
    _fields_ = [("abi_version", i32), ("variant_bits", C.c_uint32), ("kernel_source_hash", C.c_uint64), ("flags_hash", C.c_uint64),
                ("default_flags", i32), ("reserved", i32)]


BUCKET_CB = C.CFUNCTYPE(None, i32, i64, i64, vp)

# name -> (restype, argtypes); also the list the CPU test checks against include/hsimae_hip.h
SYMBOLS = {
    "hsimae_version": (C.c_int, []),
    "hsimae_build_info": (C.c_int, [C.POINTER(BuildInfo)]),
    "hsimae_variant_name": (C.c_char_p, [C.c_int]),
    "hsimae_strerror": (C.c_char_p, [C.c_int]),
    "hsimae_two_streams_active": (C.c_int, []),
    "hsimae_effective_precision": (C.c_int, [C.POINTER(Config)]),
    "hsimae_dec_block_slab_floats": (i64, []),
    "hsimae_wgrad_slab_bytes": (i64, [C.POINTER(Config), i32]),
    "hsimae_param_layout": (C.c_int, [C.POINTER(Config), C.POINTER(i64), C.POINTER(i64), C.c_int]),
    "hsimae_wpk_elems": (i64, [C.POINTER(Config)]),
    "hsimae_pack_table_bytes": (i64, [C.POINTER(Config)]),
    "hsimae_build_pack_table": (C.c_int, [C.POINTER(Config), vp, vp, vp]),
    "hsimae_pack_params": (C.c_int, [C.POINTER(Config), vp, vp]),
    "hsimae_workspace_bytes": (i64, [C.POINTER(Config), i32, i32, i32]),
    "hsimae_forward": (C.c_int, [C.POINTER(Config), C.POINTER(IO), vp]),
    "hsimae_backward": (C.c_int, [C.POINTER(Config), C.POINTER(IO), vp, BUCKET_CB, vp, vp]),
    "hsimae_mask_from_noise": (C.c_int, [C.POINTER(MaskParams), vp]),
    "hsimae_patch_gather": (C.c_int, [C.POINTER(PatchParams), vp]),
    "hsimae_gemm": (C.c_int, [C.POINTER(GemmParams), i32, i32, vp]),
    "hsimae_gemm_tiled": (C.c_int, [C.POINTER(GemmParams), i32, i32, i32, i32, vp]),
    "hsimae_pack_matrix": (C.c_int, [vp, i32, i32, vp]),
    "hsimae_enc_mlp_fwd": (C.c_int, [vp, vp, vp, i32, i32, C.POINTER(MlpWeights), vp, vp]),
    "hsimae_enc_mlp_bwd": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, C.POINTER(MlpWeights), vp, vp, vp, vp, i32, vp]),
    "hsimae_dec_block_fwd": (C.c_int, [C.POINTER(DecBlockWeights), vp, vp, vp, vp, vp, i32, i32, i32, vp]),
    "hsimae_dec_block_bwd": (C.c_int, [C.POINTER(DecBlockWeights), C.POINTER(DecBlockGrads), vp, vp, vp, vp, vp, vp, vp, i32, i32, vp, vp]),
    "hsimae_attn_fwd": (C.c_int, [C.POINTER(AttnParams), vp]),
    "hsimae_attn_bwd": (C.c_int, [C.POINTER(AttnParams), vp]),
    "hsimae_wgrad": (C.c_int, [C.POINTER(WgradParams), vp]),
    "hsimae_wgrad_msplit": (i32, [i32, i64]),
    "hsimae_ln_bwd": (C.c_int, [C.POINTER(LnBwdParams), vp]),
    "hsimae_ln_fwd": (C.c_int, [vp, vp, vp, vp, i32, i32, vp]),
    "hsimae_assemble_fwd": (C.c_int, [C.POINTER(AssembleParams), vp]),
    "hsimae_assemble_bwd": (C.c_int, [C.POINTER(AssembleParams), vp]),
    "hsimae_loss_partials": (C.c_int, [i32, i32]),
    "hsimae_loss": (C.c_int, [C.POINTER(LossParams), vp]),
    "hsimae_adamw_step": (C.c_int, [vp, vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, i32, vp]),
    "hsimae_cube_gather": (C.c_int, [C.POINTER(CubeParams), vp]),
    "hsimae_encode": (C.c_int, [C.POINTER(Config), C.POINTER(IO), vp]),
    "hsimae_encode_backward": (C.c_int, [C.POINTER(Config), C.POINTER(IO), vp, vp, BUCKET_CB, vp, vp]),
    "hsimae_agg_pool": (C.c_int, [vp, vp, i32, i32, i32, i32, vp]),
    "hsimae_head_bwd": (C.c_int, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "hsimae_decode": (C.c_int, [C.POINTER(Config), C.POINTER(IO), vp, vp, vp]),
    "hsimae_decode_backward": (C.c_int, [C.POINTER(Config), C.POINTER(IO), vp, vp, vp, BUCKET_CB, vp, vp]),
}

ABI_VERSION = 105       # HSIMAE_VERSION of include/hsimae_hip.h these ctypes structs mirror (a CPU test compares the two)
PREC_BF16, PREC_FP8 = 0, 1
A_BF16, A_F32, A_F32_LN = 0, 1, 2
E_BF16, E_F32, E_RES_F32, E_POS_F32, E_SWIGLU, E_SWIGLU_BWD, E_LN_BWD = 0, 1, 2, 3, 4, 5, 6

_lib = None


def load() -> C.CDLL:
    """Load the in-tree shared library (raises if it has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `python -m hsimae_amd.build` (hipcc --offload-arch=gfx950). "
                "hsimae_amd has no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        got = lib.hsimae_version()
        if got != ABI_VERSION:          # a stale .so next to newer bindings: struct layouts would silently disagree
            raise RuntimeError(f"{LIB_PATH} answers ABI version {got}, these bindings were written for {ABI_VERSION}: "
                               "rebuild it with `python -m hsimae_amd.build --force`")
        info = _query_build_info(lib)
        if info["variant"] and os.environ.get("HSIMAE_ALLOW_VARIANT") != "1":
            # a timing-ablation / instrumented build computes wrong results on purpose: it must never pass for the product
            raise RuntimeError(f"{LIB_PATH} is a variant build ({', '.join(info['variant'])}): its kernels are instrumented or compute "
                               "wrong results on purpose (timing ablations).  Set HSIMAE_ALLOW_VARIANT=1 to load it for an experiment.")
        _lib = lib
    return _lib


def _query_build_info(lib) -> dict:
    bi = BuildInfo()
    rc = lib.hsimae_build_info(C.byref(bi))
    if rc != 0:
        raise RuntimeError(f"hsimae_build_info failed with code {rc}")
    names = []
    for b in range(32):
        if bi.variant_bits >> b & 1:
            n = lib.hsimae_variant_name(b)
            names.append(n.decode() if n else f"bit{b}")
    return {"path": LIB_PATH, "abi_version": bi.abi_version, "variant_bits": bi.variant_bits, "variant": names,
            "kernel_source_hash": f"{bi.kernel_source_hash:016x}", "flags_hash": f"{bi.flags_hash:016x}",
            "default_flags": bool(bi.default_flags)}


def build_info() -> dict:
    """What the loaded library was built from (hsimae_build_info): ABI version, variant bits and their names, the hash of the
    kernel sources and of the hipcc flag list, whether that list is build.py's default (no tuning knob overridden)."""
    return _query_build_info(load())


def check(code: int, what: str = "hsimae") -> None:
    if code != 0:
        msg = load().hsimae_strerror(code)
        raise RuntimeError(f"{what} failed with code {code}: {msg.decode() if msg else '?'}")


def ptr(t) -> int | None:
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()
