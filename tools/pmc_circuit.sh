#!/bin/bash
# SQ / LDS counters of the circuit kernels, one rocprofv3 --pmc pass per counter group (no trace flags beside --pmc):
#   tools/pmc_circuit.sh <tag> [bench args...]   -> gpurun_out/pmc_<tag>_<i>/
cd /tmp && export TMPDIR=/tmp
tag=${1:-r03}; shift
i=0
for c in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
         "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA" \
         "GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_INST_CYCLES_SMEM SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d /root/repo/gpurun_out/pmc_${tag}_$i -- python3 /root/repo/bench.py --steps 2 --warmup 1 --repeats 1 --no-cpu-baseline --no-gate-bench --no-extras --series none "$@" > /root/repo/gpurun_out/pmc_${tag}_$i.log 2>&1
  rc=$?; echo "group $i rc=$rc"
  if [ $rc -ge 124 ]; then exit 1; fi
done
python3 /root/repo/tools/pmc_rows.py "/root/repo/gpurun_out/pmc_${tag}_*/**/*counter_collection.csv" circuit_pass
