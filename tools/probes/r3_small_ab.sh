#!/bin/bash
# latency-bound sizes: default library against the timing-only variants (r3_probe: ms per parameter-shift batch)
cd "$(dirname "$0")/../.."
for lib in default tools/_variants/libbornvi_r3_*.so; do
  unset BORNVI_LIB; [ $lib = default ] || export BORNVI_LIB=$PWD/$lib
  echo "== $lib"; timeout -k 10 200 python tools/probes/r3_probe.py 8,4 10,4 12,4 13,4 2>&1 | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r['n'], r['L'], 'r3', r['r3_ms'], 'r4', r['r4_ms'])"
done
