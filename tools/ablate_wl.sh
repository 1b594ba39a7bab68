#!/bin/bash
for df in ${FLAGS:-0 3 12 4 8}; do
  timeout -k 10 300 python bench.py --steps ${STEPS:-3} --warmup 1 --no-cpu-baseline --no-gate-bench --workload ${WL:-n20_L8_kron} --debug-flags $df ${EXTRA} 2>/dev/null | tail -1 \
    | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('${WL:-n20_L8_kron} dbg', $df, 'circuits_ms', r['phase_ms']['circuits'], 'passes', r['config']['passes'])"
done
