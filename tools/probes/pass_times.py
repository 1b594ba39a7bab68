"""Per-pass durations of the circuit kernels from a rocprofv3 --kernel-trace CSV: the dispatches of one kernel family are
cut into steps of PASSES launches; prints, per position in the step, the median duration in microseconds."""
import csv
import glob
import sys
import numpy as np

pat, family, passes = sys.argv[1], sys.argv[2], int(sys.argv[3])
rows = []
for f in glob.glob(pat, recursive=True):
    for r in csv.DictReader(open(f)):
        if family in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, int(r.get("Grid_Size", 0) or 0)))
rows.sort()
d = np.array([t for _, t, _ in rows])
g = np.array([gs for _, _, gs in rows])
n = len(d) // passes * passes
d, g = d[-n:].reshape(-1, passes), g[-n:].reshape(-1, passes)
print(family, "dispatches", len(rows), "steps", d.shape[0])
print("median us per position:", [round(float(x), 1) for x in np.median(d, axis=0)], "sum", round(float(np.median(d, axis=0).sum()), 1))
print("grid (threads) per position:", [int(x) for x in np.median(g, axis=0)])
