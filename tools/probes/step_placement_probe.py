"""Probe: does the contraction time INSIDE the training step follow the allocation K_p lives in?  One process, the
headline workload (n = 16, L = 6, dense): for several copies of K_p (each built in fresh memory while the earlier ones
are still held) -- the contraction alone (20 calls back to back) and the `stein` / `circuits` phases of 20 training steps
that use this copy.  Prints the device address of each copy."""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from tensornetworks_amd import backend

dev = torch.device("cuda", 0)
vi, x = bench.make_vi("n16_L6_dense", dev, overlap=0)
n = vi.num_latent_vars
vi._prepare_stein(x)
opt_state = vi.make_optimizer(0.005, 100000, True, "adam", (0.9, 0.999))
q = torch.full((1 << n,), 1.0 / (1 << n), dtype=torch.float64, device=dev)


def alone(K):
    for _ in range(3):
        backend.stein_quadform_sym(K, q, n)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for a, b in ev:
        a.record(); backend.stein_quadform_sym(K, q, n); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))


def in_step():
    for _ in range(3):
        vi.training_step_async(*opt_state, 10.0)
    vi.timers = {}
    for _ in range(20):
        vi.training_step_async(*opt_state, 10.0)
    torch.cuda.synchronize()
    t = vi.timers
    vi.timers = None
    return bench.mean_ms(t.get("stein")), bench.mean_ms(t.get("circuits"))


def tune_windows(fn, nbytes, tries, stride_bytes, early_stop):
    """`tries` 2 MiB-aligned windows of ONE buffer as the contraction's workspace: ms per window, and the best."""
    key = backend._ws_key(dev, "qfsym")
    stride = -(-max(int(nbytes), int(stride_bytes)) // (2 << 20)) * (2 << 20)
    backend.release_workspaces()
    backend._workspaces[key] = torch.empty(stride * tries, dtype=torch.uint8, device=dev)
    times = []
    for i in range(tries):
        backend._ws_windows[key] = (i * stride, int(nbytes))
        fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(3):
            fn()
        b.record()
        torch.cuda.synchronize()
        times.append(round(a.elapsed_time(b) / 3, 4))
        if early_stop and len(times) >= 2 and min(times) < 0.95 * max(times):
            break
    best = min(range(len(times)), key=lambda i: times[i])
    backend._ws_windows[key] = (best * stride, int(nbytes))
    return times, best


held = []
for trial in range(int(os.environ.get("COPIES", "5"))):
    if trial:
        held.append(vi._K)                       # keep the old copy alive: the new one lands somewhere else
        vi._K = backend.stein_gram(vi._S, n, vi.base_kernel_length_scale, ld=int(os.environ.get("LD", backend.gram_ld(n))))
    backend.release_workspaces()
    a0 = alone(vi._K)
    s, c = in_step()
    times, kept = tune_windows(lambda: backend.stein_quadform_sym(vi._K, q, n), backend.stein_sym_workspace_bytes(dev, n),
                               int(os.environ.get("TRIES", "6")), int(os.environ.get("STRIDE_MIB", "0")) << 20,
                               not os.environ.get("NO_EARLY_STOP"))
    a1 = alone(vi._K)
    s1, c1 = in_step()
    print(f"copy {trial}: K_p at {vi._K.data_ptr():#x} pitch {vi._K.stride(0)}: untuned alone {a0:.4f} ms, in step stein {s:.4f} circuits {c:.4f}; "
          f"windows {times} kept {kept}: alone {a1:.4f} ms, in step stein {s1:.4f} circuits {c1:.4f}", flush=True)
