"""Per-rank cost of the W-way sharded step on ONE GPU, no communication (rank 0's share of the shifted circuits and of the
band pairs; the collectives replaced by local no-ops) -- EMULATION, not a scaling measurement.  Sweeps the persistent
grid of the circuit engine (workgroups per CU): a rank's batch is W times smaller than the single-GPU one."""
import os
import sys
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tensornetworks_amd import backend, paramshift_shard as shard
from tensornetworks_amd.bayesian_network import synthetic_network
from tensornetworks_amd.ksd_vi_quantum import KSDVariationalInference

dev = torch.device("cuda", 0)
for (n, L, mode) in ((16, 6, "dense"), (20, 8, "kron")):
    for W in (1, 2, 4, 8):
        shard.world = lambda group=None, W=W: (0, W)

        def fake_flat(out, msg, group=None, W=W):
            out.zero_(); out.view(W, -1)[0].copy_(msg); return out
        shard.all_gather_flat = fake_flat
        shard.all_reduce_sum = lambda msg, group=None: msg
        bn, lat, obs, x = synthetic_network(n, 0)
        torch.manual_seed(0)
        vi = KSDVariationalInference(bn, lat, obs, n, L, pytorch_device="cuda:0", gram_mode=mode)
        vi._prepare_stein(x)
        params, opt, sched = vi.make_optimizer(0.005, 1000)
        for wgs in ((0, 3, 2, 1) if n == 16 else (0,)):
            backend.set_option(dev, "fast_workgroups_per_cu", wgs)
            for _ in range(3):
                vi.training_step_async(params, opt, sched, 10.0)
            torch.cuda.synchronize()
            vi.timers = {}
            K = 20 if n == 16 else 5
            t0 = time.perf_counter()
            for _ in range(K):
                vi.training_step_async(params, opt, sched, 10.0)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / K * 1e3
            ph = {k: round(sum(a.elapsed_time(b) for a, b in v) / len(v), 3) for k, v in vi.timers.items()}
            vi.timers = None
            print(f"n={n} emulated rank 0 of {W}, workgroups/CU {wgs or 'auto'}: {ms:.3f} ms/step  {ph}", flush=True)
        backend.set_option(dev, "fast_workgroups_per_cu", 0)
        del vi
        backend.release_workspaces()
        torch.cuda.empty_cache()
