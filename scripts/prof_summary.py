#!/usr/bin/env python3
"""Summarise a rocprofv3 --stats kernel CSV: python scripts/prof_summary.py gpurun_out/<tag> [steps]"""
import csv, glob, sys
d = sys.argv[1]; steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
f = glob.glob(d + '/*/*_kernel_stats.csv')[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("kernel ms per step %.2f" % (tot / steps / 1e6))
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 10]:
    print(f"{r['Name'][:88]:88s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:8.1f} pct={float(r['TotalDurationNs'])/tot*100:5.1f}")
