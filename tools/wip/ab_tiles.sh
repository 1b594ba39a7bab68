#!/bin/bash
# tile size sweep with prefix sharing on: n = 16 kron (circuits_ms) and n = 20
run() {
  timeout -k 10 200 python bench.py --steps $2 --warmup 2 --workload $3 --no-cpu-baseline --no-gate-bench $4 2>/dev/null | grep '^{' \
    | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'ms/step', round(r['ms_per_step'],4), 'circuits_ms', round(r['phase_ms']['circuits'],4), 'loss', r['loss_first_last'][-1])" || echo "$1 FAILED"
}
for k in 10 11 12; do run "n16 k$k" 8 n16_L6_kron "--opt tile_bits_multi=$k"; done
for k in 11 12 13; do run "n20 k$k" 2 n20_L8_kron "--opt tile_bits_multi=$k"; done
run "n16 dense" 8 n16_L6_dense ""
