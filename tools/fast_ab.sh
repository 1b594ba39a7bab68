#!/bin/bash
# A/B of the fast circuit kernel against the generic one; loss must agree
for fp in ${FP:-0 1}; do
  timeout -k 10 120 python bench.py --steps 6 --warmup 2 --workload ${WL:-n16_L6_kron} --no-cpu-baseline --no-gate-bench --fast-path $fp ${EXTRA} 2>&1 \
    | python -c "import json,sys; t=sys.stdin.read(); r=json.loads(t.strip().splitlines()[-1]); print('fast', $fp, 'ms/step', r['ms_per_step'], 'circuits_ms', r['phase_ms']['circuits'], 'loss', r['loss_first_last'])" || echo "fast $fp FAILED"
done
