#!/usr/bin/env python3
"""HBM bytes per bench step from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected separately, as
/opt/skills/guides/MI355X_MICROARCH.md prescribes):   python scripts/step_traffic.py <dir_fetch> <dir_write> <steps> <out.json> [label]
bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 summed over every dispatch, divided by the number of steps the profiled command ran
(warm-up included).  FETCH_SIZE is doubled per the guide's gfx950 correction (it tallies 128-B requests at 64 B)."""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hsimae_amd.build import kernel_source_hash  # noqa: E402

fetch_dir, write_dir, steps, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
label = sys.argv[5] if len(sys.argv) > 5 else ""


def collect(d, counter):
    tot, per = 0.0, collections.defaultdict(float)
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                v = float(r["Counter_Value"])
                tot += v
                per[r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:60]] += v
    return tot, per


fs, fper = collect(fetch_dir, "FETCH_SIZE")
ws, wper = collect(write_dir, "WRITE_SIZE")
total = (2 * fs + ws) * 1024 / steps
kern = {k: round((2 * fper.get(k, 0) + wper.get(k, 0)) * 1024 / steps) for k in set(fper) | set(wper)}
top = dict(sorted(kern.items(), key=lambda kv: -kv[1])[:12])
allk = {k: v for k, v in sorted(kern.items(), key=lambda kv: -kv[1]) if v > 0 and not k.startswith(("void at::", "__amd"))}
json.dump({"hbm_bytes_per_step": round(total), "fetch_kb_per_step": round(fs / steps), "write_kb_per_step": round(ws / steps),
           "formula": "(2*FETCH_SIZE + WRITE_SIZE)*1024 over all dispatches / steps", "steps_profiled": steps, "source": label,
           "kernel_source_sha": kernel_source_hash(),
           "top_kernels_bytes_per_step": top, "kernels_bytes_per_step": allk}, open(out, "w"), indent=1)
print(json.dumps({"hbm_GB_per_step": round(total / 1e9, 3), "top": {k: round(v / 1e9, 3) for k, v in list(top.items())[:6]}}))
