# same-box A/B of the inter-pass buffer layout (BORNVI_CONTIG_OUT = 0: 1-KiB runs on both sides; 1: contiguous tile stores)
set -e
timeout -k 10 700 python -m pytest tests/test_gpu_r3.py tests/test_gpu_circuit.py -m gpu -x -q > gpurun_out/contig_tests.log 2>&1 || { tail -30 gpurun_out/contig_tests.log; exit 1; }
tail -2 gpurun_out/contig_tests.log
for w in n16_L6_dense n20_L8_kron; do
for c in 0 1 0 1; do
  BORNVI_CONTIG_OUT=$c timeout -k 10 300 python bench.py --workload $w --series none --no-cpu-baseline --no-gate-bench --no-extras --steps 20 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w contig_out=$c', d['ms_per_step'], 'circuits', d['phase_ms']['circuits'], 'finish', d['phase_ms']['finish'], 'stein', d['phase_ms']['stein'])"
done
done
