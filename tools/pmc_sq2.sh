#!/bin/bash
# SQ / instruction-cache counters of the circuit kernel for a given library build (BORNVI_LIB)
cd /tmp && export TMPDIR=/tmp
tag=$1; shift
i=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_MISSES"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d /root/repo/gpurun_out/pmc_${tag}_$i -- python3 /root/repo/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-gate-bench --no-extras --workload n16_L6_kron "$@" > /root/repo/gpurun_out/pmc_${tag}_$i.log 2>&1
  echo rc=$?
done
