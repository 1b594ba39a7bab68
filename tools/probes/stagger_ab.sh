set -e
for v in shipped stag224 stag448 shipped; do
  if [ $v = shipped ]; then unset BORNVI_LIB; else export BORNVI_LIB=$PWD/tools/_variants/libbornvi_r3_$v.so; fi
  timeout -k 10 300 python bench.py --workload n20_L8_kron --series none --no-cpu-baseline --no-gate-bench --no-extras --steps 10 --warmup 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v n20', d['ms_per_step'], 'circuits', d['phase_ms']['circuits'], 'finish', d['phase_ms']['finish'])"
done
