"""Device timeline of the last few training steps of a rocprofv3 --kernel-trace CSV: per dispatch its start relative to
the step's first kernel, its duration and the gap to the previous dispatch (us).
  python tools/probes/step_timeline.py 'gpurun_out/prof_x/**/*kernel_trace.csv' [steps]"""
import csv, glob, os, sys
files = sorted(glob.glob(sys.argv[1], recursive=True), key=os.path.getmtime)
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows = []
for r in csv.DictReader(open(files[-1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:90]))
rows.sort()
heads = [i for i, r in enumerate(rows) if "build_gates_kernel" in r[2]]
spans = list(zip(heads, heads[1:]))
mid = len(spans) * 3 // 4
for a, b in spans[mid:mid + nsteps]:
    t0 = rows[a][0]
    print(f"--- step of {b - a} dispatches, {(rows[b][0] - t0) / 1e3:.1f} us to the next step's head")
    prev_end = t0
    for i in range(a, b):
        s, e, k = rows[i]
        print(f"  +{(s - t0) / 1e3:7.1f} us  dur {(e - s) / 1e3:6.1f}  gap {(s - prev_end) / 1e3:5.1f}  {k}")
        prev_end = e
