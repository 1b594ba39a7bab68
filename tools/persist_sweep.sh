#!/bin/bash
# persistent circuit kernel: workgroups per CU sweep (0 = one workgroup per tile); loss must not change
for w in ${WGS:-0 1 2 3 4}; do
  timeout -k 10 120 python bench.py --steps 6 --warmup 2 --workload ${WL:-n16_L6_kron} --no-cpu-baseline --no-gate-bench --wgs-per-cu $w ${EXTRA} 2>/dev/null \
    | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('wgs/cu', $w, 'ms/step', r['ms_per_step'], 'circuits_ms', r['phase_ms']['circuits'], 'loss', r['loss_first_last'])"
done
