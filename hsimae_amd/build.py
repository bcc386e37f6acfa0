"""Build libhsimae_hip.so in-tree with hipcc for gfx950 (one object per translation unit, in parallel)."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libhsimae_hip.so")
UNITS = ["gemm", "gemm_dma", "attn", "wgrad", "elem", "pack", "fused_dec", "fused_enc", "loader", "api"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# HSIMAE_HIPCC_EXTRA: extra hipcc flags for experiments (e.g. "-Xclang -target-feature -Xclang -packed-fp32-ops",
# which removes the v_pk_*_f32 forms: measured neutral on the step, so not the default)
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-value"] + os.environ.get("HSIMAE_HIPCC_EXTRA", "").split()


def _stale(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    hdrs = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "kernels.h"),
            os.path.join(HERE, "..", "include", "hsimae_hip.h")]
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)

    def compile_one(u: str) -> str:
        src, obj = os.path.join(CSRC, u + ".hip"), os.path.join(objdir, u + ".o")
        if force or _stale(obj, [src] + hdrs):
            cmd = [HIPCC] + FLAGS + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode:
                raise RuntimeError(f"hipcc failed for {u}.hip:\n{r.stderr[-4000:]}")
        return obj

    with ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, UNITS))
    if force or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
