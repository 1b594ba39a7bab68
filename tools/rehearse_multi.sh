#!/bin/bash
# N ranks sharing ONE GPU (gloo for the collectives), started by torchrun exactly as the driver starts the N-GPU bench:
# rehearsal of the sharded step incl. --dist-selftest and the n = 20 series leg.  Timings are meaningless here (the ranks
# time-share the card); what is checked: it runs, the self-test passes, the losses equal the single-rank ones.
N=${N:-2}
export BORNVI_DIST_BACKEND=gloo
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29517 \
  bench.py --gpus $N --steps 3 --warmup 1 --repeats 2 --dist-selftest --no-cpu-baseline --no-gate-bench ${EXTRA} 2> gpurun_out/rehearse_${N}.err > gpurun_out/rehearse_${N}.json
echo "rc=$?"
grep '^{' gpurun_out/rehearse_${N}.json | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('ranks', r['n_gpus'], 'loss', r['loss_first_last'], 'ms', r['ms_per_step'], r['phase_ms_per_rank'], 'selftest', r['dist_selftest']['ok'], r['dist_selftest']['max_rel_err_grad'], 'series', r['series'][0]['value'], r['series'][0]['loss_first_last'])"
tail -n 3 gpurun_out/rehearse_${N}.err
