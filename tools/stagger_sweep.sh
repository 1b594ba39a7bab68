#!/bin/bash
for wl in ${WLS:-n16_L6_kron n20_L8_kron}; do for st in ${TICKS:-0 300 600 900 1200 1800}; do
  timeout -k 10 300 python bench.py --steps ${STEPS:-6} --warmup 2 --no-cpu-baseline --no-gate-bench --workload $wl --stagger $st 2>/dev/null | tail -1 \
   | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$wl', 'stagger', $st, 'steps/s', r['value'], r['phase_ms'], r['loss_first_last'][1])"
done; done
