#!/bin/bash
# whole-step A/B of the two circuit kernels through bench.py (same box): default, then BORNVI_REG_WIRES=3 BORNVI_READ_MAP=1
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
for cfg in ${CFGS:-"4 0" "3 1"}; do
  set -- $cfg
  for wl in ${WLS:-n16_L6_dense n12_L4_dense n8_L4_dense}; do
    BORNVI_REG_WIRES=$1 BORNVI_READ_MAP=$2 timeout -k 10 300 python bench.py --steps ${STEPS:-20} --warmup 5 --workload $wl --no-cpu-baseline --no-gate-bench --series none --no-extras > gpurun_out/r3_bench_$1_$2_$wl.log 2>&1
    rc=$?
    tail -1 gpurun_out/r3_bench_$1_$2_$wl.log | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('reg_wires $1 read_map $2 $wl', 'steps/s', round(r['value'],2), 'ms', r['ms_per_step'], r.get('phase_ms'))" || tail -3 gpurun_out/r3_bench_$1_$2_$wl.log
    if [ $rc -ge 124 ]; then echo "timed out"; exit 1; fi
  done
done
