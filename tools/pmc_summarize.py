#!/usr/bin/env python3
"""gpurun_out/pmc_traffic_{FETCH_SIZE,WRITE_SIZE}/ (tools/pmc_traffic.sh) -> profiles/<out>.json: HBM bytes per launch of
every kernel.  gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts half of a wide (16 B/lane)
coalesced read, so fetch bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE * 1024 is exact."""
import collections
import csv
import glob
import json
import os
import re
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = sys.argv[1] if len(sys.argv) > 1 else "r02_pmc_traffic_n16_L6_dense.json"
note = sys.argv[2] if len(sys.argv) > 2 else ""
commit = sys.argv[3] if len(sys.argv) > 3 else "?"


def short(name):
    name = re.sub(r"^void ", "", name)
    name = name.split("(")[0]
    return name.replace("bornvi::", "")


tot = {"FETCH_SIZE": collections.defaultdict(float), "WRITE_SIZE": collections.defaultdict(float)}
cnt = {"FETCH_SIZE": collections.Counter(), "WRITE_SIZE": collections.Counter()}
for c in tot:
    files = glob.glob(os.path.join(REPO, "gpurun_out", f"pmc_traffic_{c}", "*", "*counter_collection.csv"))
    # (gpurun merges new files into gpurun_out without deleting older runs': take the newest run only)
    for f in sorted(files, key=os.path.getmtime)[-1:]:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != c:
                continue
            k = short(r["Kernel_Name"])
            tot[c][k] += float(r["Counter_Value"])
            cnt[c][k] += 1
kern = {}
for k in sorted(set(tot["FETCH_SIZE"]) | set(tot["WRITE_SIZE"])):
    if "at::native" in k or k.startswith("__amd"):
        continue
    nf, nw = cnt["FETCH_SIZE"][k], cnt["WRITE_SIZE"][k]
    fb = 2.0 * tot["FETCH_SIZE"][k] * 1024 / nf if nf else 0.0
    wb = tot["WRITE_SIZE"][k] * 1024 / nw if nw else 0.0
    kern[k] = {"fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb, "hbm_bytes_per_launch": fb + wb,
               "launches_sampled": int(max(nf, nw))}
# the plan these bytes belong to: bench.py shows the traffic only while the library still plans the same passes
workload = sys.argv[4] if len(sys.argv) > 4 else "n16_L6_dense"
variant = sys.argv[5] if len(sys.argv) > 5 else "r3"          # the pass kernel of the profiled run (r3: 8 amplitudes per thread, read map)
sys.path.insert(0, REPO)
from tensornetworks_amd import _ext  # noqa: E402  (host-only planner entry point: no GPU needed)
n_w, l_w = int(workload.split("_")[0][1:]), int(workload.split("_")[1][1:])
plan = _ext.plan_words(_ext.ANSATZ_IDS["hardware_efficient"], n_w, l_w, (_ext.R3 | 0x100) if variant == "r3" else 0)
signature = {"workload": workload, "passes": int(plan[3]), "tile_bits": int(plan[2]), "kernel": variant, "commit": commit}
rec = {"signature": signature, "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/pmc_traffic.sh), bench.py --steps 2 "
                 f"--warmup 1, workload {workload}" + (" -- " + note if note else ""),
       "correction": "gfx950: FETCH_SIZE counts half of a wide (16 B/lane) coalesced read -> fetch bytes = 2 * FETCH_SIZE * 1024; "
                     "WRITE_SIZE * 1024 exact (MI355X_MICROARCH.md, HBM)",
       "kernels": kern}
json.dump(rec, open(os.path.join(REPO, "profiles", out), "w"), indent=1)
for k, v in kern.items():
    print(f"{k:45s} fetch {v['fetch_bytes_per_launch'] / 1e9:9.4f} GB  write {v['write_bytes_per_launch'] / 1e9:9.4f} GB  x{v['launches_sampled']}")
