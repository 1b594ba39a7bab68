#!/bin/bash
# timing-only ablation sweep of circuit_pass_kernel (results are invalid when flags != 0)
# flags: 1 skip U math, 2 skip stage LDS traffic, 4 skip global load, 8 skip global store, 16 skip stage barrier
for df in ${FLAGS:-0 1 2 3 4 8 12 15 16 31}; do
  timeout -k 10 120 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-gate-bench --debug-flags $df ${EXTRA} 2>/dev/null \
    | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('dbg', $df, 'tile', r['config']['tile_bits'], 'circuits_ms', r['phase_ms']['circuits'])"
done
