#!/usr/bin/env python3
"""GPU idle time between kernels from a rocprofv3 --kernel-trace CSV: union of the kernel intervals against the span of the
last full step, the idle gaps grouped by the kernel that ran before them.  usage: trace_gaps.py <kernel_trace.csv> [steps]"""
import csv, sys, collections, re

def nm(x):
    m = re.search(r"(\w+_kernel(?:<[^>]*>)?|__amd\w+)", x)
    return m.group(1) if m else x[:40]


rows = list(csv.DictReader(open(sys.argv[1])))
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), nm(r["Kernel_Name"])) for r in rows))
# steps end with the optimizer / loss kernels; take the window between the last nsteps+1 occurrences of the step's first kernel
idx = [i for i, e in enumerate(ev) if e[2].startswith("patch_gather_kernel")]
lo, hi = idx[-nsteps - 1], idx[-1]
win = ev[lo:hi]
t0, t1 = win[0][0], max(e[1] for e in win)
busy, gaps, cur_end, prev = 0, collections.defaultdict(lambda: [0, 0]), win[0][0], None
for s, e, n in win:
    if s > cur_end:
        g = gaps[prev]; g[0] += s - cur_end; g[1] += 1
        busy += 0
        cur_start = s
    if e > cur_end:
        busy += e - max(s, cur_end); cur_end = e; prev = n
span = t1 - t0
print(f"window: {nsteps} steps, {len(win)} kernels, span {span/nsteps/1e6:.3f} ms/step, busy {busy/nsteps/1e6:.3f} ms/step, idle {(span-busy)/nsteps/1e6:.3f} ms/step")
print(f"sum of kernel durations {sum(e-s for s,e,_ in win)/nsteps/1e6:.3f} ms/step")
for n, (t, c) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:14]:
    print(f"  after {n:60s} {c/nsteps:6.1f} gaps/step  {t/c/1e3:6.2f} us each  {t/nsteps/1e6:.3f} ms/step")
