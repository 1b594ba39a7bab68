#!/bin/bash
# HBM traffic of the hot kernels: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes
# (MI355X_MICROARCH.md: TCC slots; gfx950: FETCH_SIZE counts half of a wide coalesced read).
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /root/repo/gpurun_out/pmc_traffic_$c
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d /root/repo/gpurun_out/pmc_traffic_$c -- \
    python3 /root/repo/bench.py --steps 2 --warmup 1 --repeats 1 --no-cpu-baseline --no-gate-bench --no-extras --series none "$@" > /root/repo/gpurun_out/pmc_traffic_$c.log 2>&1
  echo "$c rc=$?"
done
