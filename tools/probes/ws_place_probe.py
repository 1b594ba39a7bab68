"""Probe: is a "slow" K_p allocation slow whatever workspace the contraction writes its partials to?  ALLOCS padded copies
of K_p held at once; per copy the contraction is timed with WS different workspace buffers (fresh device allocations of
different sizes in between, so that each lands somewhere else)."""
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


def clock(K):
    for _ in range(3):
        backend.stein_quadform_sym(K, q, n)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(12)]
    for a, b in ev:
        a.record(); backend.stein_quadform_sym(K, q, n); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))


held, spacers = [], []
for a in range(int(os.environ.get("ALLOCS", "4"))):
    K = backend.stein_gram(S, n, 1.0, ld=backend.gram_ld(n))
    held.append(K)
    row = []
    for w in range(int(os.environ.get("WS", "4"))):
        backend.release_workspaces()
        spacers.append(torch.empty((3 + 5 * w + a) << 20, dtype=torch.uint8, device=dev))   # keeps the old block busy and shifts the next
        t = clock(K)
        wsb = backend._workspaces[(dev.index, "qfsym", int(torch.cuda.current_stream(dev).cuda_stream))]
        row.append((round(t, 4), hex(wsb.data_ptr())))
    print(f"K_p copy {a} at {K.data_ptr():#x}: " + "  ".join(f"{t} ms (ws {p})" for t, p in row), flush=True)
