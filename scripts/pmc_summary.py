"""Per-kernel averages of rocprofv3 --pmc counter_collection CSVs: python scripts/pmc_summary.py <dir> [<dir> ...]"""
import collections
import csv
import glob
import sys

rows = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = (r["Kernel_Name"].replace("(anonymous namespace)::", "")[:48], r["Grid_Size"])
            rows[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            key = (f, r["Dispatch_Id"])
            if key not in seen:
                seen.add(key)
                dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
names = sorted({c for v in rows.values() for c in v})
tot = {k: sum(v) / max(1, len(names) if False else 1) for k, v in dur.items()}
print(f"{'kernel':50s} {'grid':>8s} {'n':>4s} {'us':>8s} " + " ".join(f"{n[:14]:>14s}" for n in names))
for k in sorted(rows, key=lambda k: -sum(dur[k]))[:18]:
    n = len(next(iter(rows[k].values())))
    print(f"{k[0]:50s} {k[1]:>8s} {n:4d} {sum(dur[k]) / len(dur[k]):8.1f} " +
          " ".join(f"{(sum(rows[k][c]) / len(rows[k][c]) if rows[k][c] else 0):14.1f}" for c in names))
