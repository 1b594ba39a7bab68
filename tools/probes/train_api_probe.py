"""Probe: epochs/s of the drop-in KSDVariationalInference.train() itself, host_sync=True (the reference's epoch: loss.item()
per epoch) against host_sync=False (deferred read-backs; HIP-graph replay for n <= 13), BASELINE configs 2 and 3."""
import contextlib
import gc
import io
import os
import sys
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench

dev = torch.device("cuda", 0)
for workload, epochs in (("n8_L4_dense", 400), ("n12_L4_dense", 400), ("n16_L6_dense", 60)):
    for host_sync in (False, True, False):
        vi, x = bench.make_vi(workload, dev)
        with contextlib.redirect_stdout(io.StringIO()):
            vi.train(x, 6, 0.005, verbose=False, host_sync=host_sync)          # plans, K_p, placement
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            h = vi.train(x, epochs, 0.005, verbose=False, host_sync=host_sync)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        print(f"{workload} train({epochs} epochs, host_sync={host_sync}): {epochs / dt:.1f} epochs/s ({1e3 * dt / epochs:.4f} ms/epoch), "
              f"KSD {h['loss_ksd'][0]:.4f} -> {h['loss_ksd'][-1]:.4f}", flush=True)
        del vi, h
        gc.collect()                      # (the trainer's closures hold K_p in reference cycles)
        from tensornetworks_amd import backend
        backend.release_workspaces()
        torch.cuda.empty_cache()
        print(f'   free {torch.cuda.mem_get_info(dev)[0] / 2**30:.1f} GiB', flush=True)
