"""Build libhsimae_hip.so in-tree with hipcc for gfx950 (one object per translation unit, in parallel)."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libhsimae_hip.so")
UNITS = ["gemm", "attn", "attn_wide", "wgrad", "elem", "pack", "fused_dec", "fused_enc", "loader", "api"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# HSIMAE_HIPCC_EXTRA: extra hipcc flags for experiments (e.g. "-Xclang -target-feature -Xclang -packed-fp32-ops",
# which removes the v_pk_*_f32 forms: measured neutral on the step, so not the default)
BASE_FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-value", "-fvisibility=hidden"]
FLAGS = BASE_FLAGS + os.environ.get("HSIMAE_HIPCC_EXTRA", "").split()


def flags_hash(flags: list[str]) -> str:
    """16 hex digits over the hipcc flag list: names the object directory (a flag change recompiles: the mtime check alone did
    not notice one) and is compiled into the library (hsimae_build_info().flags_hash)."""
    import hashlib
    return hashlib.sha256(" ".join(flags).encode()).hexdigest()[:16]


def unit_flags(unit: str, flags: list[str]) -> list[str]:
    """Flags of one translation unit: `flags` + what hsimae_build_info() reports (api.hip only: the flag-list hash, whether the
    list is the default one, the kernel-source hash)."""
    if unit != "api":
        return list(flags)
    return list(flags) + [f"-DHS_BUILD_FLAGS_HASH=0x{flags_hash(flags)}ULL", f"-DHS_BUILD_DEFAULT_FLAGS={int(flags == BASE_FLAGS)}",
                          f"-DHS_KERNEL_SOURCE_HASH=0x{kernel_source_hash()}ULL"]


def kernel_source_hash() -> str:
    """sha256 (16 hex digits) over every kernel source and header the library is built from.  Measurement files under
    profiles/ (step traffic) record it, and bench.py flags them as stale when the kernels have changed since."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h", ".cpp")))
    files.append(os.path.join(HERE, "..", "include", "hsimae_hip.h"))
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def _stale(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    hdrs = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "kernels.h"), os.path.join(CSRC, "plan.h"),
            os.path.join(HERE, "..", "include", "hsimae_hip.h")]
    # one object directory per flag list, so that a changed HSIMAE_HIPCC_EXTRA never links objects of another build
    objdir = os.path.join(HERE, "build", flags_hash(FLAGS))
    os.makedirs(objdir, exist_ok=True)
    srchash = kernel_source_hash()

    def compile_one(u: str) -> str:
        src = os.path.join(CSRC, u + ".hip")
        # api.o carries the kernel-source hash: its name changes with ANY source, so it is rebuilt whenever the hash it reports would be stale
        obj = os.path.join(objdir, (f"api_{srchash}.o" if u == "api" else u + ".o"))
        if u == "api":
            for old in os.listdir(objdir):
                if old.startswith("api_") and old != os.path.basename(obj):
                    os.remove(os.path.join(objdir, old))
        if force or _stale(obj, [src] + hdrs):
            cmd = [HIPCC] + unit_flags(u, FLAGS) + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode:
                raise RuntimeError(f"hipcc failed for {u}.hip:\n{r.stderr[-4000:]}")
        return obj

    with ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, UNITS))
    stamp = os.path.join(HERE, "build", "linked_from")
    linked = open(stamp).read() if os.path.exists(stamp) else ""
    if force or _stale(LIB, objs) or linked != objdir:          # (objects of another flag list may be OLDER than the library)
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
        with open(stamp, "w") as fh:
            fh.write(objdir)
    return LIB


ASAN_LIB = os.path.join(HERE, "libhsimae_plan_asan.so")


def build_asan(force: bool = False) -> str:
    """Host-only sanitizer build of the planning code (csrc/plan.h: geometry, layouts, pack table, workspace carve) behind
    the library's own C entry points: g++ -fsanitize=address,undefined.  CPU only; never loaded by the product."""
    src = os.path.join(CSRC, "plan_host.cpp")
    deps = [src, os.path.join(CSRC, "plan.h"), os.path.join(HERE, "..", "include", "hsimae_hip.h")]
    if force or _stale(ASAN_LIB, deps):
        cmd = [os.environ.get("CXX", "g++"), "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
               "-fno-sanitize-recover=undefined", "-shared", "-fPIC", src, "-o", ASAN_LIB]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            raise RuntimeError(f"sanitizer build failed:\n{r.stderr[-4000:]}")
    return ASAN_LIB


def asan_runtime() -> str:
    """Path of libasan for LD_PRELOAD (the interpreter itself is not instrumented)."""
    r = subprocess.run([os.environ.get("CXX", "g++"), "-print-file-name=libasan.so"], capture_output=True, text=True)
    return os.path.realpath(r.stdout.strip())


if __name__ == "__main__":
    if "--asan" in sys.argv:
        print(build_asan(force="--force" in sys.argv))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
