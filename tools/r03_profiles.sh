#!/bin/bash
# Round-3 profile set on the GPU box (every rocprofv3 run is its own process; --pmc runs carry no trace flags):
#   kernel-trace + stats of the n = 16 and n = 20 bench workloads, HBM traffic (FETCH_SIZE / WRITE_SIZE) of both,
#   SQ / LDS counters of the circuit kernel at both sizes, matrix-core counters of the Gram build and the batched contraction.
cd "$(dirname "$0")/.."
R=$PWD
mkdir -p gpurun_out profiles
commit=${COMMIT:-worktree}
step() { echo "=== $(date +%T) $*"; }
for wl in n16_L6_dense n20_L8_kron; do
  step "kernel trace $wl"
  rm -rf gpurun_out/prof_r03_$wl
  STEPS=${STEPS:-10} tools/prof_stats.sh r03_$wl --workload $wl || exit 1
  f=$(ls gpurun_out/prof_r03_$wl/*/*kernel_stats.csv | head -1)
  cp "$f" profiles/r03_${wl}_kernel_stats.csv
  python3 tools/trace_step_stats.py "gpurun_out/prof_r03_$wl/**/*kernel_trace.csv" > profiles/r03_${wl}_kernel_stats_in_step.csv
  tail -1 gpurun_out/prof_r03_$wl.log > profiles/r03_${wl}_bench_under_rocprof.json
  head -8 profiles/r03_${wl}_kernel_stats_in_step.csv
done
python3 tools/probes/pass_times.py "gpurun_out/prof_r03_n20_L8_kron/**/*kernel_trace.csv" "circuit_pass_r3_kernel<false>" 12 > profiles/r03_n20_L8_circuit_pass_times.txt 2>&1
python3 tools/probes/pass_times.py "gpurun_out/prof_r03_n16_L6_dense/**/*kernel_trace.csv" "circuit_pass_r3_kernel<false>" 7 > profiles/r03_n16_L6_circuit_pass_times.txt 2>&1
cat profiles/r03_n20_L8_circuit_pass_times.txt
for wl in n16_L6_dense n20_L8_kron; do
  step "HBM traffic $wl"
  tools/pmc_traffic.sh --workload $wl || exit 1
  python3 tools/pmc_summarize.py r03_pmc_traffic_$wl.json "" $commit $wl r3
done
for wl in n16_L6_dense n20_L8_kron; do
  step "SQ counters $wl"
  rm -rf gpurun_out/pmc_r03sq_${wl}_*
  tools/pmc_circuit.sh r03sq_$wl --workload $wl > gpurun_out/pmc_r03sq_$wl.txt 2>&1 || exit 1
  python3 tools/pmc_rows.py "$R/gpurun_out/pmc_r03sq_${wl}_*/**/*counter_collection.csv" circuit_pass profiles/r03_pmc_sq_circuit_$wl.json | head -30
done
step "MFMA counters"
rm -rf gpurun_out/pmc_r03_mfma
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_r03_mfma -- python3 $R/tools/mfma_time.py > $R/gpurun_out/pmc_r03_mfma.log 2>&1; echo rc=$? )
python3 tools/pmc_rows.py "$R/gpurun_out/pmc_r03_mfma/**/*counter_collection.csv" kernel profiles/r03_pmc_mfma_n16.json | grep -A8 "gram_mfma\|quadform_batched"
grep "rep" gpurun_out/pmc_r03_mfma.log
