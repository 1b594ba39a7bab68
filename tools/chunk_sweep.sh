#!/bin/bash
# circuits per chunk (workspace cap) sweep: do the ping-pong states of a chunk stay in the 256 MiB Infinity Cache?
for wl in ${WLS:-n16_L6_kron n16_L6_dense}; do for bc in ${CHUNKS:-577 288 192 144 96 72}; do
  cap=$(( bc * (2 * 1048576 + 6144) + 4096 ))
  BORNVI_WORKSPACE_CAP=$cap timeout -k 10 300 python bench.py --steps ${STEPS:-10} --warmup 2 --no-cpu-baseline --no-gate-bench --workload $wl 2>/dev/null | tail -1 \
   | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$wl', 'chunk', $bc, 'steps/s', r['value'], r['phase_ms'])"
done; done
