#!/bin/bash
for ov in 0 1; do
  timeout -k 10 200 python bench.py --steps ${STEPS:-20} --warmup 3 --no-cpu-baseline --no-gate-bench --overlap $ov ${EXTRA} 2> gpurun_out/overlap_$ov.err | tail -1 \
   | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('overlap', $ov, r['value'], r['ms_per_step'], r['phase_ms'], r['loss_first_last'])" || tail -5 gpurun_out/overlap_$ov.err
done
