"""Per-kernel averages over the TRAINING STEPS of a rocprofv3 --kernel-trace CSV.  rocprofv3 --stats averages over every
dispatch of the process, i.e. also the contractions _place_gram times on the K_p copies it does not keep; this lists,
per kernel, the dispatches that sit between the first circuit pass of a step and its optimiser update:
  python tools/trace_step_stats.py 'gpurun_out/prof_x/**/*kernel_trace.csv' > profiles/..._kernel_stats_in_step.csv"""
import csv
import glob
import sys
from collections import OrderedDict

import os
rows = []
files = sorted(glob.glob(sys.argv[1], recursive=True), key=os.path.getmtime)
for f in files[-1:]:                   # the newest trace only (gpurun merges the runs of a round into one directory)
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]))
rows.sort()
# a step = from a build_gates_kernel (the head of the circuit batch) up to the dispatch before the next one
heads = [i for i, r in enumerate(rows) if "build_gates_kernel" in r[2]]
stats = OrderedDict()
steps = 0
for a, b in zip(heads, heads[1:]):
    names = [rows[i][2] for i in range(a, b)]
    if not any(("shift_dot_kernel" in x) or ("dot_finish_kernel" in x) for x in names) or any("gram" in x for x in names):
        continue                       # not a training step (a probability evaluation, a set-up phase)
    steps += 1
    for i in range(a, b):
        s, e, k = rows[i]
        d = stats.setdefault(k, [0, 0])
        d[0] += 1
        d[1] += e - s
print("Name,Calls,AverageNs,CallsPerStep,Steps")
for k, (c, t) in sorted(stats.items(), key=lambda kv: -kv[1][1]):
    print(f'"{k}",{c},{t / c:.1f},{c / max(steps, 1):.2f},{steps}')
