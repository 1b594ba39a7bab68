"""Per-dispatch durations (ms) of the kernels whose name contains one of the given substrings, from a rocprofv3
--kernel-trace CSV, in dispatch order, grouped in runs of GROUP dispatches (median per group)."""
import csv
import glob
import sys
import numpy as np

pat, group = sys.argv[1], int(sys.argv[2])
names = sys.argv[3:]
rows = []
for f in glob.glob(pat, recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
rows.sort()
for nm in names:
    d = [t for _, k, t in rows if nm in k]
    meds = [round(float(np.median(d[i:i + group])), 4) for i in range(0, len(d), group)]
    print(nm, len(d), "dispatches; median ms per group of", group, ":", meds)
