"""Per-phase cycle shares of the fused decoder kernels (run on the GPU box: python3 scripts/phase_timing.py).

Builds a second copy of the library with -DHS_PHASE_TIMING (wave 0 of every workgroup accumulates s_memtime deltas
between the kernels' barriers), runs one C2-sized forward+backward and prints where each kernel's time goes.
The shipped libhsimae_hip.so is not touched and never contains the instrumentation.
"""
import ctypes
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build_variant() -> str:
    from hsimae_amd import build as B
    tmp = tempfile.mkdtemp(prefix="hs_phase_")
    objs = []
    for u in B.UNITS:
        obj = os.path.join(tmp, u + ".o")
        timed = u in ("fused_dec", "fused_enc", "gemm")
        flags = B.unit_flags(u, B.FLAGS + (["-DHS_PHASE_TIMING"] + os.environ.get("HS_EXTRA_FLAGS", "").split() if timed else []))
        src_obj = os.path.join(B.HERE, "build", B.flags_hash(B.FLAGS), u + ".o")
        if not timed and os.path.exists(src_obj):
            shutil.copy(src_obj, obj)
        else:
            subprocess.run([B.HIPCC] + flags + ["-c", os.path.join(B.CSRC, u + ".hip"), "-o", obj], check=True)
        objs.append(obj)
    lib = os.path.join(tmp, "libhsimae_hip.so")
    subprocess.run([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs, check=True)
    return lib


def main():
    # PHASE_LIB: a library built beforehand with -DHS_PHASE_TIMING (python scripts/build_variant.py <name> -DHS_PHASE_TIMING ...): no
    # compile on the GPU box.  The loader refuses instrumented builds unless asked (hsimae_build_info, round 6).
    os.environ["HSIMAE_ALLOW_VARIANT"] = "1"
    lib_path = os.environ.get("PHASE_LIB") or build_variant()
    import hsimae_amd._lib as L
    L.LIB_PATH = lib_path
    import torch
    from hsimae_amd import HSIMAE
    # MODEL=base|large|huge (huge runs the fp8 encoder linears), BATCH overrides the per-model default
    bands, dim, heads, dbatch = {"base": (96, 128, 8, 4096), "large": (96, 256, 16, 4096), "huge": (192, 512, 32, 1024)}[os.environ.get("MODEL", "base")]
    batch = int(os.environ.get("BATCH", str(dbatch)))
    torch.manual_seed(0)
    m = HSIMAE(img_size=9, patch_size=3, in_chans=1, bands=bands, b_patch_size=8, embed_dim=dim, depth=12, num_heads=heads,
               s_depth=9, decoder_embed_dim=64, decoder_depth=8, decoder_num_heads=8, norm_pix_loss=True,
               trunc_init=True).cuda()
    if dim == 512:
        m.set_precision("fp8")
    x = torch.rand(batch, 1, bands, 9, 9, device="cuda")
    lib = L.load()
    lib.hsimae_debug_phases.restype = ctypes.c_int
    lib.hsimae_debug_phases.argtypes = [ctypes.POINTER(ctypes.c_uint64), ctypes.c_int]
    buf = (ctypes.c_uint64 * 32)()
    buf2 = (ctypes.c_uint64 * 32)()
    lib.hsimae_debug_phases_enc.restype = ctypes.c_int
    lib.hsimae_debug_phases_enc.argtypes = [ctypes.POINTER(ctypes.c_uint64), ctypes.c_int]
    for it in range(3):
        loss, _, _ = m(x, 0.75)
        loss.backward()
        torch.cuda.synchronize()
        lib.hsimae_debug_phases(buf, 1)
        lib.hsimae_debug_phases_enc(buf2, 1)
    buf3 = (ctypes.c_uint64 * 64)()
    lib.hsimae_debug_phases_gemm.restype = ctypes.c_int
    lib.hsimae_debug_phases_gemm.argtypes = [ctypes.POINTER(ctypes.c_uint64), ctypes.c_int]
    lib.hsimae_debug_phases_gemm(buf3, 1)          # accumulated over the 3 iterations
    v = list(buf) + list(buf2)
    g = list(buf3)
    for name, base in (("gemm A_BF16 -> bf16 (dO / misc)", 0), ("gemm A_BF16 -> f32 family (proj+res, w2+res, du, du2, pos)", 4),
                       ("gemm A_BF16 E_LN_BWD (du + LN1 bwd)", 8), ("gemm A_F32 -> bf16", 12), ("gemm A_F32 -> other (gate backward)", 16),
                       ("gemm A_F32_LN -> bf16 (LN1 + qkv)", 24), ("gemm A_F32_LN -> other (LN2 + w1|w3 + gate)", 28)):
        tot = sum(g[base:base + 3]) or 1
        print(f"{name}: wave-0 cycles {tot}: stage {100.0 * g[base] / tot:.1f} %  k-loop {100.0 * g[base + 1] / tot:.1f} %  epilogue {100.0 * g[base + 2] / tot:.1f} %")
    names = {
        "dec_bwd_attn": (0, ["prologue (x dx1 O lse) + LN", "qkv mm", "dO mm + dWp", "delta (+ barrier)", "dO store", "attention",
                             "du mm + dWqkv", "epilogue LN bwd"]),
        "dec_bwd_mlp": (8, ["prologue LN", "gate mm + silu", "wgrad", "du2 stage", "epilogue LN bwd", "du2 mm", "-", "-"]),
        "dec_fwd": (16, ["LN1 + residual", "qkv", "attention", "o store + proj", "LN2", "gate chunks", "w2 mm", "store"]),
        "dec_bwd_attn, inside 'du mm + dWqkv'": (24, ["next sample's loads issued", "du = dq Wq + dk Wk + dv Wv", "dWq|dWk|dWv + biases",
                                                      "du -> fp32 tile, re-reads issued", "barrier", "-", "-", "-"]),
        "enc_mlp_bwd": (32, ["prologue LN", "gate mm + silu", "operand stores", "du2 mm", "epilogue LN bwd", "LN grads", "-", "-"]),
        "enc_mlp_fwd": (40, ["prologue LN", "bias init", "gate + W2 chunks", "(barrier)", "store", "-", "-", "-"]),
    }
    for k, (base, ph) in names.items():
        tot = sum(v[base:base + 8]) or 1
        print(f"{k}: total wave-0 cycles {tot}")
        for i, n in enumerate(ph):
            if v[base + i]:
                print(f"    {n:24s} {100.0 * v[base + i] / tot:5.1f} %")


if __name__ == "__main__":
    main()
