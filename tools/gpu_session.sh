#!/bin/bash
# One GPU-box session: runs the named steps in order; a step that TIMES OUT or is killed ends the session (no further GPU
# step after a hang), an ordinary failure (assertion, non-zero exit) is logged and the session goes on.
#   tools/gpu_session.sh step1 step2 ...   (steps = functions below)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
run() {   # run <seconds> <logfile> <cmd...>
  local secs=$1 log=$2; shift 2
  echo "=== $(date +%T) $* (limit ${secs}s) -> $log"
  timeout -k 10 "$secs" "$@" > "gpurun_out/$log" 2>&1
  local rc=$?
  echo "    rc=$rc"
  if [ $rc -ge 124 ]; then echo "!!! step timed out or was killed: ending the session"; exit 1; fi
  return 0
}
mall() { run 120 r2_mall_probe.log tools/_variants/mall_probe; tail -n 25 gpurun_out/r2_mall_probe.log; }
sym() {
  for lib in tools/_variants/libbornvi_sym_*.so; do
    v=$(basename $lib .so)
    BORNVI_LIB=$PWD/$lib PADS=${PADS:-0} ALLOCS=${ALLOCS:-4} run 240 r2_sym_probe_$v.log python tools/probes/sym_probe.py
    grep -v amdgpu.ids gpurun_out/r2_sym_probe_$v.log
  done
}
sym_default() { PADS=${PADS:-0,32,0,32} ALLOCS=${ALLOCS:-3} run 400 r2_sym_probe_default.log python tools/probes/sym_probe.py; grep -v amdgpu.ids gpurun_out/r2_sym_probe_default.log; }
step_place() { run 600 r2_step_placement.log python tools/probes/step_placement_probe.py; grep "^copy" gpurun_out/r2_step_placement.log; }
sym_place() {   # contraction alone on ALLOCS padded copies of K_p: default build, then every tools/_variants/libbornvi_sym_*.so
  for lib in default tools/_variants/libbornvi_sym_*.so; do
    v=$(basename $lib .so); unset BORNVI_LIB; [ $lib = default ] || export BORNVI_LIB=$PWD/$lib
    PADS=${PADS:-32} ALLOCS=${ALLOCS:-7} run 400 r2_sym_place_$v.log python tools/probes/sym_probe.py; grep "pad=" gpurun_out/r2_sym_place_$v.log
  done; unset BORNVI_LIB
}
ws_place() { run 400 r2_ws_place.log python tools/probes/ws_place_probe.py; grep "^K_p" gpurun_out/r2_ws_place.log; }
sym_trace() {   # which kernel carries the allocation-dependent time: main kernel or reduce?
  cd /tmp; rm -rf /root/repo/gpurun_out/sym_trace
  PADS=32 ALLOCS=${ALLOCS:-6} timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d /root/repo/gpurun_out/sym_trace -- python3 /root/repo/tools/probes/sym_probe.py > /root/repo/gpurun_out/r2_sym_trace.log 2>&1
  rc=$?; echo "rc=$rc"; [ $rc -ge 124 ] && exit 1
  cd /root/repo; grep "pad=" gpurun_out/r2_sym_trace.log
  python tools/probes/kernel_durations.py 'gpurun_out/sym_trace/**/*kernel_trace.csv' 18 quadform_sym_kernel quadform_sym_reduce_kernel
}
ws_in_kp() { run 400 r2_ws_in_kp.log python tools/probes/ws_in_kp_probe.py; grep "^copy" gpurun_out/r2_ws_in_kp.log; FRONT=1 run 400 r2_ws_in_kp_front.log python tools/probes/ws_in_kp_probe.py; grep "^copy" gpurun_out/r2_ws_in_kp_front.log; }
ws_far() { run 500 r2_ws_far.log python tools/probes/ws_far_probe.py; grep -A1 "^K_p" gpurun_out/r2_ws_far.log; }
copy_far() { run 500 r2_copy_far.log python tools/probes/copy_far_probe.py; grep "^trial" gpurun_out/r2_copy_far.log; }
sym_small() { run 300 r2_sym_small.log python tools/probes/sym_vs_full_probe.py; grep "^n=" gpurun_out/r2_sym_small.log; }
uncached() { run 500 r2_uncached.log python tools/probes/uncached_kp_probe.py; grep "ms per\|failed" gpurun_out/r2_uncached.log; }
chunks() { run 600 r2_chunk_probe.log python tools/probes/chunk_stream_probe.py; grep -v amdgpu.ids gpurun_out/r2_chunk_probe.log | tail -14; }
circ_ab() {
  run 300 r2_circ_ab_default.log python tools/probes/circuit_ab.py; grep "n=" gpurun_out/r2_circ_ab_default.log
  for lib in tools/_variants/libbornvi_circ_*.so; do
    v=$(basename $lib .so)
    BORNVI_LIB=$PWD/$lib run 300 r2_circ_ab_$v.log python tools/probes/circuit_ab.py; grep "n=" gpurun_out/r2_circ_ab_$v.log
  done
}
circ_opts() { for o in ${CIRC_OPTS}; do BORNVI_OPTS=$o run 300 r2_circ_ab_opt_$o.log python tools/probes/circuit_ab.py; grep "n=" gpurun_out/r2_circ_ab_opt_$o.log; done; }
stamps() { for v in ${STAMP_LIBS:-stamps}; do BORNVI_LIB=$PWD/tools/_variants/libbornvi_circ_$v.so run 300 r2_stamp_probe_$v.log python tools/probes/stamp_probe.py; echo "--- $v"; grep -v amdgpu.ids gpurun_out/r2_stamp_probe_$v.log; done; }
tests_circ_variant() { BORNVI_LIB=$PWD/tools/_variants/libbornvi_circ_${CIRC_VARIANT}.so run 600 r2_gpu_tests_circ_${CIRC_VARIANT}.log python -m pytest tests/test_gpu_circuit.py -m gpu -x -q; tail -n 4 gpurun_out/r2_gpu_tests_circ_${CIRC_VARIANT}.log; }
tilesweep() { run 600 r2_tile_sweep.log python tools/tile_sweep_n.py; grep "^n=" gpurun_out/r2_tile_sweep.log; }
rehearse() { for N in 2 4; do N=$N tools/rehearse_multi.sh; done; }
emulate() { run 900 r2_emulate_ranks.log python tools/probes/emulate_ranks.py; grep "^n=" gpurun_out/r2_emulate_ranks.log; }
pmc_sq() {
  cd /tmp
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" \
             "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
             "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES SQ_INSTS_SMEM GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64"; do
    tag=$(echo $set | cut -d' ' -f1)
    rm -rf /root/repo/gpurun_out/pmc_sq_$tag
    timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d /root/repo/gpurun_out/pmc_sq_$tag -- \
      python3 /root/repo/bench.py --steps 2 --warmup 1 --repeats 1 --no-cpu-baseline --no-gate-bench --no-extras --series none > /root/repo/gpurun_out/pmc_sq_$tag.log 2>&1
    echo "$tag rc=$?"
    python3 /root/repo/tools/probes/pmc_rows.py "/root/repo/gpurun_out/pmc_sq_$tag/**/*counter_collection.csv" circuit_pass_fast | tail -3
  done
  cd /root/repo
}
pmc_circ() {   # instruction counts of the circuit engine, per pass: default build and every tools/_variants/libbornvi_circ_*.so
  cd /tmp
  for lib in default /root/repo/tools/_variants/libbornvi_circ_*.so; do
    v=$(basename $lib .so); [ $lib = default ] || export BORNVI_LIB=$lib
    for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES SQ_ACTIVE_INST_VALU" \
               "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS"; do
      tag=${v}_$(echo $set | cut -d' ' -f1)
      rm -rf /root/repo/gpurun_out/pmc_circ_$tag
      timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d /root/repo/gpurun_out/pmc_circ_$tag -- \
        python3 /root/repo/tools/probes/circuit_once.py > /root/repo/gpurun_out/pmc_circ_$tag.log 2>&1
      rc=$?; echo "$tag rc=$rc"; [ $rc -ge 124 ] && exit 1
      python3 /root/repo/tools/probes/pmc_rows.py "/root/repo/gpurun_out/pmc_circ_$tag/**/*counter_collection.csv" circuit_pass_fast | sed -n '1p;16,22p'
    done
  done
  unset BORNVI_LIB
  cd /root/repo
}
profile() {
  tools/prof_stats.sh r02_n16_L6_dense 2>&1 | tail -14
  tools/prof_stats.sh r02_n20_L8_kron --workload n20_L8_kron 2>&1 | tail -8
  tools/pmc_traffic.sh 2>&1 | tail -3
}
tests_advq() { run 600 r2_gpu_tests_adv.log python -m pytest tests/test_gpu_adversarial.py -m gpu -x -q; tail -n 25 gpurun_out/r2_gpu_tests_adv.log; }
bench_adv() { run 300 r2_bench_adv.log python bench.py --steps ${ADV_STEPS:-200} --warmup 3 --workload adv_n12_b65536; grep '^{' gpurun_out/r2_bench_adv.log > gpurun_out/r2_bench_adv.json; tail -c 1500 gpurun_out/r2_bench_adv.json; }
tests_graph() { run 600 r2_gpu_tests_graph.log python -m pytest tests/test_gpu_trainer.py -m gpu -x -q -k "graphed or async"; tail -n 25 gpurun_out/r2_gpu_tests_graph.log; }
bench_small() { for w in n8_L4_dense n12_L4_dense; do run 300 r2_bench_$w.log python bench.py --steps 200 --warmup 5 --workload $w --no-cpu-baseline --no-gate-bench --series none; grep '^{' gpurun_out/r2_bench_$w.log > gpurun_out/r2_bench_$w.json; python -c "
import json; r=json.load(open('gpurun_out/r2_bench_$w.json')); print('$w', r['value'], 'steps/s', r['ms_per_step'], 'ms', r['launch'], r['extras']['adjoint_engine'])"; done; }
tests_adj() { run 600 r2_gpu_tests_adj.log python -m pytest tests/test_gpu_adjoint.py -m gpu -x -q; tail -n 25 gpurun_out/r2_gpu_tests_adj.log; }
tests_stein() { run 600 r2_gpu_tests_stein.log python -m pytest tests/test_gpu_stein.py tests/test_gpu_shard.py tests/test_gpu_trainer.py -m gpu -x -q; tail -n 15 gpurun_out/r2_gpu_tests_stein.log; }
tests_shard() { run 600 r2_gpu_tests_shard.log python -m pytest tests/test_gpu_shard.py -m gpu -x -q; tail -n 15 gpurun_out/r2_gpu_tests_shard.log; }
tests() { run 900 r2_gpu_tests.log python -m pytest tests -m gpu -x -q; tail -n 30 gpurun_out/r2_gpu_tests.log; }
bench() { run 600 r2_bench.log python bench.py --steps 20 --warmup 5; grep '^{' gpurun_out/r2_bench.log > gpurun_out/r2_bench.json; tail -c 3000 gpurun_out/r2_bench.log; }
bench_ab() {   # the headline step with backend options switched (BENCH_AB_OPTS="name=value ..."), same box: A B A B
  for rep in 1 2; do for o in default ${BENCH_AB_OPTS}; do
    arg=""; [ $o = default ] || arg="--opt $o"
    run 300 r2_bench_ab_${o}_$rep.log python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-gate-bench --no-extras --series none $arg
    grep '^{' gpurun_out/r2_bench_ab_${o}_$rep.log | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$o', r['value'], r['ms_per_step'], r['phase_ms']['circuits'], r['phase_ms']['stein'])"
  done; done
}
bench_libs() {   # the headline step with other builds of the library (tools/_variants/libbornvi_circ_*.so), same box: A B A B
  for rep in 1 2; do for lib in default tools/_variants/libbornvi_circ_*.so; do
    v=$(basename $lib .so); unset BORNVI_LIB; [ $lib = default ] || export BORNVI_LIB=$PWD/$lib
    run 300 r2_bench_lib_${v}_$rep.log python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-gate-bench --no-extras --series none
    grep '^{' gpurun_out/r2_bench_lib_${v}_$rep.log | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$v', r['value'], r['ms_per_step'], r['phase_ms']['circuits'], r['phase_ms']['stein'], r['phase_ms']['finish'])"
  done; done; unset BORNVI_LIB
}
bench_place() {   # the headline step with 1 and with the default number of K_p placement tries, same box: A B A B A B
  for rep in 1 2 3; do for tries in 1 -1; do
    BORNVI_PLACEMENT_TRIES=$tries run 300 r2_bench_place_${tries}_$rep.log python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-gate-bench --no-extras --series none
    grep '^{' gpurun_out/r2_bench_place_${tries}_$rep.log | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('tries $tries', r['value'], r['ms_per_step'], r['phase_ms']['circuits'], r['phase_ms']['stein'], r['precompute_seconds'], (r['gram_placement'] or {}).get('kept'))"
  done; done
}
small_libs() {   # the small graph-replayed workloads under every circuit-engine variant, same box
  for rep in 1 2; do for lib in default tools/_variants/libbornvi_circ_*.so; do
    v=$(basename $lib .so); unset BORNVI_LIB; [ $lib = default ] || export BORNVI_LIB=$PWD/$lib
    for w in n8_L4_dense n12_L4_dense; do
      run 300 r2_small_${v}_$w.log python bench.py --steps 200 --warmup 5 --workload $w --no-cpu-baseline --no-gate-bench --no-extras --series none
      grep '^{' gpurun_out/r2_small_${v}_$w.log | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$v $w', r['value'], r['ms_per_step'])"
    done
  done; done; unset BORNVI_LIB
}
bench_overlap() { for o in -1 1 2; do run 300 r2_bench_overlap_$o.log python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-gate-bench --no-extras --series none --overlap $o
    grep '^{' gpurun_out/r2_bench_overlap_$o.log | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('overlap $o', r['value'], r['ms_per_step'], r['phase_ms']['circuits'], r['phase_ms']['stein'], r['overlap'])"; done; }
prof_adv() {
  cd /tmp; rm -rf /root/repo/gpurun_out/prof_adv
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_adv -- python3 /root/repo/bench.py --steps 200 --warmup 3 --workload adv_n12_b65536 > /root/repo/gpurun_out/prof_adv.log 2>&1
  rc=$?; echo "rc=$rc"; [ $rc -ge 124 ] && exit 1
  cd /root/repo
  f=$(ls -t gpurun_out/prof_adv/*/*kernel_stats.csv | head -1)
  python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:22]:
    print('%-70s calls %6s avg_us %9.1f  pct %5.1f' % (r['Name'].split('(')[0][-70:], r['Calls'], float(r['AverageNs'])/1e3, 100*float(r['TotalDurationNs'])/tot))
PY
}
prof_small() {
  cd /tmp; rm -rf /root/repo/gpurun_out/prof_small
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_small -- python3 /root/repo/bench.py --steps 200 --warmup 5 --workload ${SMALL_WL:-n8_L4_dense} --no-cpu-baseline --no-gate-bench --no-extras --series none > /root/repo/gpurun_out/prof_small.log 2>&1
  rc=$?; echo "rc=$rc"; [ $rc -ge 124 ] && exit 1
  cd /root/repo
  f=$(ls -t gpurun_out/prof_small/*/*kernel_stats.csv | head -1)
  python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:16]:
    print('%-70s calls %6s avg_us %9.1f  pct %5.1f' % (r['Name'].split('(')[0][-70:], r['Calls'], float(r['AverageNs'])/1e3, 100*float(r['TotalDurationNs'])/tot))
PY
}
smoke() { run 300 r2_smoke.log python __graft_entry__.py smoke; tail -n 5 gpurun_out/r2_smoke.log; }
stream() { run 300 r2_stream_probe_zero.log tools/_variants/stream_probe 3 0 0; cat gpurun_out/r2_stream_probe_zero.log
           run 300 r2_stream_probe_rand.log tools/_variants/stream_probe 3 0 1; cat gpurun_out/r2_stream_probe_rand.log; }
alloc() { run 300 r2_alloc_probe.log python tools/probes/alloc_probe.py; grep -v amdgpu.ids gpurun_out/r2_alloc_probe.log; }
batched() { run 600 r2_batched_probe.log python tools/probes/batched_probe.py; grep -v amdgpu.ids gpurun_out/r2_batched_probe.log | tail -6; }
alloc_parts() { for w in 512 1024 2048; do BORNVI_SYM_MIN_WGS=$w ALLOCS=4 run 300 r2_alloc_probe_wgs$w.log python tools/probes/alloc_probe.py; echo "min_wgs $w"; grep "^alloc" gpurun_out/r2_alloc_probe_wgs$w.log | cut -c1-60; done; }
alloc_noz() { BORNVI_SYM_ABLATE=1 run 300 r2_alloc_probe_noz.log python tools/probes/alloc_probe.py; grep -v amdgpu.ids gpurun_out/r2_alloc_probe_noz.log; }
pmc_tlb() {
  rm -rf gpurun_out/pmc_tlb
  run 600 r2_pmc_tlb.log rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE \
      --kernel-trace -d gpurun_out/pmc_tlb --output-format csv -- python tools/probes/alloc_probe.py
  grep -v amdgpu.ids gpurun_out/r2_pmc_tlb.log | tail -8
  python tools/probes/pmc_rows.py 'gpurun_out/pmc_tlb/**/*counter_collection.csv' quadform_sym_kernel > gpurun_out/r2_pmc_tlb_rows.txt; head -80 gpurun_out/r2_pmc_tlb_rows.txt
}
counters() { run 120 r2_counters.txt rocprofv3 -L; grep -c . gpurun_out/r2_counters.txt; }
for step in "$@"; do $step; done
echo "=== session done"
