#!/usr/bin/env python3
"""A/B builds: python scripts/build_variant.py <name> [-DFLAG ...] -> variants/<name>/libhsimae_hip.so (git-ignored; run with
HSIMAE_LIB=variants/<name>/libhsimae_hip.so).  The shipped library is not touched."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hsimae_amd import build as B  # noqa: E402

name, flags = sys.argv[1], sys.argv[2:]
out = os.path.join(ROOT, "variants", name)
os.makedirs(out, exist_ok=True)
objs = []
procs = []
for u in B.UNITS:
    obj = os.path.join(out, u + ".o")
    objs.append(obj)
    procs.append(subprocess.Popen([B.HIPCC] + B.unit_flags(u, B.FLAGS + flags) + ["-c", os.path.join(B.CSRC, u + ".hip"), "-o", obj],
                                  stderr=subprocess.DEVNULL))
for p in procs:
    assert p.wait() == 0
lib = os.path.join(out, "libhsimae_hip.so")
subprocess.run([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs, check=True)
for o in objs:
    os.remove(o)
print(lib)
