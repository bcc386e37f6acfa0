"""Per-phase cycle shares of blk128_bwd_kernel (run on the GPU box: python3 scripts/phase_blk128_bwd.py).
Builds a copy of the library whose attn.hip carries -DHS_PHASE_TIMING (wave 5 of every workgroup accumulates cycle-counter
deltas between the kernel's phases), runs Base-sized forward + backward passes and prints where the kernel's time goes."""
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
        if u == "attn":
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
    m = HSIMAE(img_size=9, patch_size=3, in_chans=1, bands=96, b_patch_size=8, embed_dim=128, depth=12, num_heads=8, s_depth=9,
               decoder_embed_dim=64, decoder_depth=8, decoder_num_heads=8, norm_pix_loss=True, trunc_init=True).cuda()
    x = torch.rand(int(os.environ.get("BATCH", "4096")), 1, 96, 9, 9, device="cuda")
    lib = L.load()
    lib.hsimae_debug_phases_b128.restype = ctypes.c_int
    lib.hsimae_debug_phases_b128.argtypes = [ctypes.POINTER(ctypes.c_uint64), ctypes.c_int]
    buf = (ctypes.c_uint64 * 32)()
    for it in range(3):
        m.zero_grad()
        loss, _, _ = m(x, 0.75)
        loss.backward()
        torch.cuda.synchronize()
        lib.hsimae_debug_phases_b128(buf, 1 if it < 2 else 0)
    v = list(buf)
    names = ["wait: group's images complete", "issue loads (epilogue rows, next group)", "dO + delta", "mask + attention core",
             "wait: dq|dk|dv complete", "dq|dk|dv row stores", "du product", "wait: du tile complete", "LayerNorm epilogue + dx store",
             "commit next group's images"]
    tot = sum(v[:10]) or 1
    print(f"blk128_bwd: wave-5 cycles {tot} per step (21 launches)")
    for n, c in zip(names, v):
        print(f"    {n:42s} {100.0 * c / tot:5.1f} %")
    f = v[16:24]
    fnames = ["LN1 + u store", "issue next group's x loads", "wait: U image complete", "q|k|v products -> images", "attention",
              "wait: O image complete", "residual loads + q|k|v / o / lse row stores", "projection + x1 store"]
    tot = sum(f) or 1
    print(f"blk128_fwd: wave-5 cycles {tot} per step (21 launches)")
    for n, c in zip(fnames, f):
        print(f"    {n:42s} {100.0 * c / tot:5.1f} %")


if __name__ == "__main__":
    main()
