#!/bin/bash
run() {
  timeout -k 10 200 python bench.py --steps $2 --warmup 2 --workload $3 --no-cpu-baseline --no-gate-bench --no-extras $4 2>/dev/null | grep '^{' \
    | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'passes', r['config']['passes'], 'ms/step', round(r['ms_per_step'],4), 'circuits_ms', round(r['phase_ms']['circuits'],4), 'loss', r['loss_first_last'][-1])" || echo "$1 FAILED"
}
for lo in 4 5 6 7; do run "n16 lo$lo" 8 n16_L6_kron "--opt low_bits=$lo"; done
for lo in 4 5 6 7; do run "n20 lo$lo" 2 n20_L8_kron "--opt low_bits=$lo"; done
