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
  for v in 8_4_1 4_4_2 8_2_2 2_8_2 4_4_1 4_2_2; do
    BORNVI_LIB=$PWD/tools/_variants/libbornvi_sym_$v.so run 240 r2_sym_probe_$v.log python tools/probes/sym_probe.py
    cat gpurun_out/r2_sym_probe_$v.log | grep -v amdgpu.ids
  done
}
tests() { run 900 r2_gpu_tests.log python -m pytest tests -m gpu -x -q; tail -n 30 gpurun_out/r2_gpu_tests.log; }
bench() { run 600 r2_bench.log python bench.py --steps 20 --warmup 5; grep '^{' gpurun_out/r2_bench.log > gpurun_out/r2_bench.json; tail -c 3000 gpurun_out/r2_bench.log; }
smoke() { run 300 r2_smoke.log python __graft_entry__.py smoke; tail -n 5 gpurun_out/r2_smoke.log; }
for step in "$@"; do $step; done
echo "=== session done"
