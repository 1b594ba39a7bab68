#!/bin/bash
# Round-3 late additions on the GPU box: the driver-style bench line, the kernel trace of the n = 16 step and of the
# graph-replayed n = 8 step, matrix-core counters of the Gram build (gram_tables_kernel).  Everything lands in gpurun_out/.
cd "$(dirname "$0")/.."
R=$PWD
mkdir -p gpurun_out
step() { echo "=== $(date +%T) $*"; }
step "bench (driver invocation)"
timeout -k 10 500 python3 bench.py > gpurun_out/r03b_bench.log 2>&1 || exit 1
tail -1 gpurun_out/r03b_bench.log > gpurun_out/r03b_bench_n16_L6_dense.json
step "kernel trace n16"
rm -rf gpurun_out/prof_r03b_n16_L6_dense
STEPS=10 tools/prof_stats.sh r03b_n16_L6_dense --workload n16_L6_dense || exit 1
step "kernel trace n8 (graph replay)"
rm -rf gpurun_out/prof_r03b_n8_L4_dense
STEPS=50 tools/prof_stats.sh r03b_n8_L4_dense --workload n8_L4_dense || exit 1
step "MFMA counters"
rm -rf gpurun_out/pmc_r03b_mfma
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_r03b_mfma -- python3 $R/tools/mfma_time.py > $R/gpurun_out/pmc_r03b_mfma.log 2>&1; echo rc=$? )
grep "rep" gpurun_out/pmc_r03b_mfma.log
step done
