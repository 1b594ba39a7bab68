set -e
export BORNVI_LIB=$PWD/tools/_variants/libbornvi_r3_nostages.so
for c in 0 1 0 1; do
  BORNVI_CONTIG_OUT=$c timeout -k 10 300 python bench.py --workload n20_L8_kron --series none --no-cpu-baseline --no-gate-bench --no-extras --steps 10 --warmup 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('nostages n20 contig_out=$c', d['ms_per_step'], 'circuits', d['phase_ms']['circuits'], 'finish', d['phase_ms']['finish'])"
done
