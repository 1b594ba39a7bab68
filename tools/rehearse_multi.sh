#!/bin/bash
# N ranks sharing ONE GPU (gloo for the two small all-gathers): rehearsal of the sharded step; checks that the
# sharded losses equal the single-rank ones.  Timings are meaningless here (the ranks time-share the card).
N=${N:-2}
export BORNVI_DIST_BACKEND=gloo
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29517 \
  bench.py --gpus $N --steps 3 --warmup 1 --no-cpu-baseline --no-gate-bench ${EXTRA} 2> gpurun_out/rehearse_${N}.err | tee gpurun_out/rehearse_${N}.json \
  | grep '^{' | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('ranks', r['n_gpus'], 'loss', r['loss_first_last'], 'ms', r['ms_per_step'], r['phase_ms'])"
