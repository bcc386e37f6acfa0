"""Per-phase cycle shares of blk256_fwd_kernel (run on the GPU box: python3 scripts/phase_wide.py).
Builds a copy of the library whose attn_wide.hip carries -DHS_PHASE_TIMING (wave 5 of every workgroup accumulates cycle-counter
deltas between the kernel's phases), runs Large-sized forward passes and prints where the kernel's time goes."""
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
    B.build()
    tmp = tempfile.mkdtemp(prefix="hs_phase_")
    objs = []
    for u in B.UNITS:
        obj = os.path.join(tmp, u + ".o")
        if u == "attn_wide":
            subprocess.run([B.HIPCC] + B.FLAGS + ["-DHS_PHASE_TIMING"] + os.environ.get("HS_EXTRA_FLAGS", "").split() +
                           ["-c", os.path.join(B.CSRC, u + ".hip"), "-o", obj], check=True)
        else:
            shutil.copy(os.path.join(B.HERE, "build", u + ".o"), obj)
        objs.append(obj)
    lib = os.path.join(tmp, "libhsimae_hip.so")
    subprocess.run([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs, check=True)
    return lib


def main():
    lib_path = build_variant()
    import hsimae_amd._lib as L
    L.LIB_PATH = lib_path
    import torch
    from hsimae_amd import HSIMAE
    torch.manual_seed(0)
    m = HSIMAE(img_size=9, patch_size=3, in_chans=1, bands=96, b_patch_size=8, embed_dim=256, depth=12, num_heads=16, s_depth=9,
               decoder_embed_dim=64, decoder_depth=8, decoder_num_heads=8, norm_pix_loss=True, trunc_init=True).cuda()
    x = torch.rand(int(os.environ.get("BATCH", "4096")), 1, 96, 9, 9, device="cuda")
    lib = L.load()
    lib.hsimae_debug_phases_wide.restype = ctypes.c_int
    lib.hsimae_debug_phases_wide.argtypes = [ctypes.POINTER(ctypes.c_uint64), ctypes.c_int]
    buf = (ctypes.c_uint64 * 16)()
    with torch.no_grad():
        for it in range(3):
            m(x, 0.75)
            torch.cuda.synchronize()
            lib.hsimae_debug_phases_wide(buf, 1 if it < 2 else 0)
    v = list(buf)
    names = ["LN1 + u store", "wait B1", "q|k|v products", "staging writes + prefetch issue", "wait B2", "q|k|v row stores",
             "attention", "wait B3", "o / lse stores (+ early loads)", "projection + x1 store", "wait B4", "loop overhead"]
    tot = sum(v[:12]) or 1
    print(f"blk256_fwd: wave-5 cycles {tot} per step (21 launches)")
    for n, c in zip(names, v):
        print(f"    {n:34s} {100.0 * c / tot:5.1f} %")


if __name__ == "__main__":
    main()
