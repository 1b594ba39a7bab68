#!/bin/bash
# rocprofv3 kernel stats of a bench command -> gpurun_out/prof_<tag>/   (tools/prof_stats.sh <tag> [bench args...])
cd /tmp && export TMPDIR=/tmp
tag=${1:-r02}; shift
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_$tag -- \
  python3 /root/repo/bench.py --steps ${STEPS:-10} --warmup 3 --repeats 3 --no-cpu-baseline --no-gate-bench --no-extras --series none "$@" > /root/repo/gpurun_out/prof_$tag.log 2>&1
echo rc=$?
f=$(ls /root/repo/gpurun_out/prof_$tag/*/*kernel_stats.csv | head -1)
python3 - "$f" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:12]:
    print('%-60s calls %5s avg_us %10.1f  pct %s' % (r['Name'].split('(')[0][-60:], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
PY
