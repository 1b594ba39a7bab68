#!/bin/bash
# does running the passes over sub-batches that fit the 256 MB Infinity Cache pay?  (chunking via the workspace cap)
for c in 33 65 97 129 193 289 577; do
  cap=$(( c * (2*16*65536 + 112*64) + 4*1024*1024 ))
  BORNVI_WORKSPACE_CAP=$cap timeout -k 10 200 python bench.py --steps 8 --warmup 2 --workload n16_L6_kron --no-cpu-baseline --no-gate-bench --no-extras 2>/dev/null | grep '^{' \
    | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('circuits per chunk ~', $c, 'circuits_ms', r['phase_ms']['circuits'], 'loss', r['loss_first_last'][-1])"
done
