#!/bin/bash
cd /tmp && export TMPDIR=/tmp
i=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INSTS_SALU" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH SQ_INSTS_SENDMSG"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d /root/repo/gpurun_out/pmc_sq_$i -- python3 /root/repo/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-gate-bench --workload n16_L6_kron ${EXTRA} > /root/repo/gpurun_out/pmc_sq_$i.log 2>&1
  echo rc=$?
done
