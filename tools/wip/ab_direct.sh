#!/bin/bash
# same-box A/B: direct-stage masks x tile sizes, n=16 kron and n=20, and the committed-HEAD build if present
run() { # label, extra args, env
  timeout -k 10 200 python bench.py --steps $2 --warmup 2 --workload $3 --no-cpu-baseline --no-gate-bench $4 2>/dev/null | grep '^{' \
    | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'ms/step', round(r['ms_per_step'],4), 'circuits_ms', round(r['phase_ms']['circuits'],4), 'loss', r['loss_first_last'][-1])" || echo "$1 FAILED"
}
for m in 0 3; do for k in 11 12; do run "n16 mask$m k$k" 8 n16_L6_kron "--opt direct_stages=$m --opt tile_bits_multi=$k" || exit 1; done; done
for m in 0 3; do run "n20 mask$m" 2 n20_L8_kron "--opt direct_stages=$m" || exit 1; done
if [ -f tools/_variants/libhead.so ]; then BORNVI_LIB=tools/_variants/libhead.so run "n16 head" 8 n16_L6_kron "" ; BORNVI_LIB=tools/_variants/libhead.so run "n20 head" 2 n20_L8_kron ""; fi
