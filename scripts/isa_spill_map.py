#!/usr/bin/env python3
"""Where a kernel's scratch spills / reloads sit relative to its barriers, global loads and stores (round 6).
A scratch reload is a vector-memory load on gfx950: its s_waitcnt vmcnt(n) also waits for every OLDER global load of the wave, so a
reload between a prefetch burst and the code the burst was meant to hide behind exposes the whole HBM round trip.
    python scripts/isa_spill_map.py <unit.hip> <kernel-name substring> [-DFLAG ...]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hsimae_amd import build as B  # noqa: E402

unit, kern, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
with tempfile.TemporaryDirectory() as d:
    out = os.path.join(d, "k.s")
    subprocess.run([B.HIPCC] + B.BASE_FLAGS + flags + ["-S", "--cuda-device-only", os.path.join(B.CSRC, unit), "-o", out],
                   check=True, stderr=subprocess.DEVNULL)
    s = open(out).read()
for m in re.finditer(r"^(_Z\w*%s\w*):" % re.escape(kern), s, re.M):
    name = m.group(1)
    body = s[m.end():s.index(".Lfunc_end", m.end())].split("\n")
    ev = []
    for n, l in enumerate(body):
        t = l.strip()
        k = ("RELOAD" if "scratch_load" in t else "SPILL" if "scratch_store" in t else "--barrier" if t.startswith("s_barrier") else
             "gload" if t.startswith(("global_load", "buffer_load")) else "gstore" if t.startswith(("global_store", "buffer_store")) else
             "[loop" if "Loop Header" in t else "exp" if t.startswith("v_exp_f32") else "vmcnt0" if re.match(r"s_waitcnt vmcnt\(0\)", t) else None)
        if k:
            ev.append((n, k))
    out, last, cnt = [], None, 0
    for n, k in ev:
        if k == last:
            cnt += 1
        else:
            if last:
                out.append("%s x%d" % (last, cnt))
            last, cnt = k, 1
            out.append("@%d" % n)
    out.append("%s x%d" % (last, cnt))
    print(name, len(body), "lines")
    print("  " + " ".join(out))
