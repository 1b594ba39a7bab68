#!/bin/bash
# option sweep of the default library through circuit_ab.py: one line of BORNVI_OPTS per argument
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
i=0
for o in "$@"; do
  i=$((i+1))
  BORNVI_OPTS=$o timeout -k 10 ${LIMIT:-200} python tools/probes/circuit_ab.py > gpurun_out/r3_opts_$i.log 2>&1
  rc=$?; grep "n=" gpurun_out/r3_opts_$i.log || tail -3 gpurun_out/r3_opts_$i.log
  if [ $rc -ge 124 ]; then echo "!!! timed out: stopping"; exit 1; fi
done
