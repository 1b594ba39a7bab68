#!/usr/bin/env python3
"""Per-kernel sums and per-dispatch means of rocprofv3 --pmc counters:  tools/pmc_rows.py '<glob of counter_collection.csv>' <name filter>"""
import collections
import csv
import glob
import json
import re
import sys

pat, flt = sys.argv[1], sys.argv[2]
out = sys.argv[3] if len(sys.argv) > 3 else None
tot = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(pat, recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"^void ", "", r["Kernel_Name"]).replace("(anonymous namespace)::", "").split("(")[0].replace("bornvi::", "")
        if flt not in k:
            continue
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"]] += 1
res = {}
for k in sorted(tot):
    res[k] = {c: {"per_dispatch": tot[k][c] / max(cnt[k][c], 1), "dispatches": cnt[k][c]} for c in sorted(tot[k])}
    print(k)
    for c in sorted(tot[k]):
        print(f"   {c:28s} {tot[k][c] / max(cnt[k][c], 1):16.1f} per dispatch  ({cnt[k][c]} dispatches)")
if out:
    json.dump(res, open(out, "w"), indent=1)
