#!/bin/bash
for wl in ${WLS:-n16_L6_kron n16_L6_dense n20_L8_kron}; do for tb in ${TBS:-13 12 11 10}; do
  timeout -k 10 300 python bench.py --steps ${STEPS:-8} --warmup 2 --no-cpu-baseline --no-gate-bench --workload $wl --tile-bits $tb 2>/dev/null | tail -1 \
   | python -c "import json,sys; r=json.loads(sys.stdin.read()); p=r['phase_ms']; print('$wl', 'tile', $tb, 'passes', r['config']['passes'], 'steps/s', r['value'], 'circuits', p['circuits'], 'stein', p['stein'])"
done; done
