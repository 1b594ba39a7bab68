#!/bin/bash
# Same-box A/B on the GPU box: the default library and every tools/_variants/libbornvi_r3_*.so through circuit_ab.py
# with the 8-amplitude kernel selected (BORNVI_OPTS=reg_wires=3[,...]).
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
OPTS=${BORNVI_OPTS:-reg_wires=3}
for lib in ${LIBS:-default tools/_variants/libbornvi_r3_*.so}; do
  v=$(basename $lib .so); unset BORNVI_LIB; [ $lib = default ] || export BORNVI_LIB=$PWD/$lib
  BORNVI_OPTS=$OPTS timeout -k 10 ${LIMIT:-200} python tools/probes/circuit_ab.py > gpurun_out/r3_ab_$v.log 2>&1
  rc=$?; grep "n=" gpurun_out/r3_ab_$v.log
  if [ $rc -ge 124 ]; then echo "!!! $v timed out: stopping"; exit 1; fi
done
