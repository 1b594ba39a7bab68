"""Probe: contraction time against the distance (in allocation order) between K_p and its workspace: per copy of K_p,
workspaces allocated behind 0, 1, 2, ... spacer allocations of SPACER_GIB each."""
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
need = backend.stein_sym_workspace_bytes(dev, n)
key = backend._ws_key(dev, "qfsym")
gib = int(os.environ.get("SPACER_GIB", "16"))
nsp = int(os.environ.get("SPACERS", "7"))


def clock(K):
    for _ in range(3):
        backend.stein_quadform_sym(K, q, n)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(12)]
    for a, b in ev:
        a.record(); backend.stein_quadform_sym(K, q, n); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))


held = []
for a in range(int(os.environ.get("ALLOCS", "3"))):
    K = backend.stein_gram(S, n, 1.0, ld=backend.gram_ld(n))
    held.append(K)
    cands, spacers = [], []
    for i in range(nsp + 1):
        cands.append(torch.empty(need, dtype=torch.uint8, device=dev))
        if i < nsp:
            spacers.append(torch.empty(gib << 30, dtype=torch.uint8, device=dev))
    del spacers
    torch.cuda.empty_cache()
    row = []
    for c in cands:
        backend.release_workspaces()
        backend._workspaces[key] = c
        backend._ws_windows[key] = (0, need)
        row.append(round(clock(K), 4))
    backend.release_workspaces()
    print(f"K_p copy {a} at {K.data_ptr():#x}: workspaces {[hex(c.data_ptr()) for c in cands]}\\n   ms {row}", flush=True)
    del cands
    torch.cuda.empty_cache()
