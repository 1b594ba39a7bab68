"""Prints per-dispatch PMC counter values of one kernel from a rocprofv3 --pmc counter_collection CSV, in dispatch order."""
import csv
import glob
import sys
from collections import OrderedDict

pat, kernel = sys.argv[1], sys.argv[2]
rows = OrderedDict()
for f in glob.glob(pat, recursive=True):
    for r in csv.DictReader(open(f)):
        if kernel in r.get("Kernel_Name", ""):
            d = rows.setdefault(int(r["Dispatch_Id"]), {})
            d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
names = sorted({k for d in rows.values() for k in d})
print("dispatch", *names)
for i, (did, d) in enumerate(sorted(rows.items())):
    print(did, *[f"{d.get(k, 0):.4g}" for k in names])
