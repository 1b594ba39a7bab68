#!/bin/bash
# timing-only ablations of the fast circuit kernel (results invalid): 1 no gate arithmetic, 2 no LDS traffic in stages,
# 4 no tile loads, 8 no tile stores, 32 no matrix loads; DEBUG instantiation waits with vmcnt(0) everywhere
for d in 0 256 1 2 3 4 8 12 15; do
  timeout -k 10 200 python bench.py --steps 6 --warmup 2 --workload n16_L6_kron --no-cpu-baseline --no-gate-bench --no-extras --debug-flags $d 2>/dev/null | grep '^{' \
    | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('dbg', $d, 'circuits_ms', r['phase_ms']['circuits'])"
done
