#!/usr/bin/env python3
"""sum the counters of the fast circuit kernel over its launches: pmc_sum.py gpurun_out/pmc_<tag>_*"""
import csv, glob, sys, collections
for d in sys.argv[1:]:
    tot = collections.Counter(); n = 0
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "circuit_pass_fast" in r["Kernel_Name"]:
                tot[r["Counter_Name"]] += float(r["Counter_Value"]); n += 1
    print(d, {k: f"{v:.4g}" for k, v in sorted(tot.items())})
