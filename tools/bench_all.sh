#!/bin/bash
# all single-GPU workloads, one JSON line each (no CPU baseline / gate microbench)
for wl in n8_L4_dense n12_L4_dense n16_L6_dense n16_L6_kron n20_L8_kron; do
  timeout -k 10 300 python bench.py --steps ${STEPS:-5} --warmup 2 --no-cpu-baseline --no-gate-bench --workload $wl ${EXTRA} 2>/dev/null \
    | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$wl', 'steps/s', r['value'], 'ms', r['ms_per_step'], r['phase_ms'], 'passes', r['config']['passes'], 'precompute_s', r['precompute_seconds'])"
done
