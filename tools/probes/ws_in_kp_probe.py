"""Probe: the contraction's workspace INSIDE the allocation of K_p (the rows behind the matrix) against a workspace of
its own: ALLOCS padded copies of K_p held at once, each timed both ways."""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tensornetworks_amd import backend

n = 16
N = 1 << n
dev = torch.device("cuda", 0)
g = torch.Generator(device="cpu").manual_seed(0)
S = torch.randn((N, n), generator=g, dtype=torch.float64).to(dev)
q = torch.rand(N, generator=g, dtype=torch.float64).to(dev)
q /= q.sum()
ld = backend.gram_ld(n)
need = backend.stein_sym_workspace_bytes(dev, n)
tail_rows = -(-need // (ld * 8)) + 8
key = backend._ws_key(dev, "qfsym")


def clock(K):
    for _ in range(3):
        backend.stein_quadform_sym(K, q, n)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(12)]
    for a, b in ev:
        a.record(); k2, y = backend.stein_quadform_sym(K, q, n); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev])), y


held = []
front = os.environ.get("FRONT")          # workspace in front of the matrix instead of behind it
for a in range(int(os.environ.get("ALLOCS", "6"))):
    big = torch.empty((N + tail_rows) * ld, dtype=torch.float64, device=dev)
    held.append(big)
    if front:
        K = big[tail_rows * ld:].view(N, ld)[:, :N]
        tail = big[: tail_rows * ld].view(torch.uint8)
    else:
        K = big[: N * ld].view(N, ld)[:, :N]
        tail = big[N * ld:].view(torch.uint8)
    backend.stein_gram(S, n, 1.0, out=K)
    backend.release_workspaces()
    t_own, y0 = clock(K)
    backend._workspaces[key] = tail
    backend._ws_windows[key] = (0, need)
    t_in, y1 = clock(K)
    assert torch.equal(y0, y1)
    backend.release_workspaces()
    print(f"copy {a} at {big.data_ptr():#x}: own workspace {t_own:.4f} ms, workspace in the K_p allocation {t_in:.4f} ms", flush=True)
