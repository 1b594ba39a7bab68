"""Probe: symmetric (band) contraction vs the full-matrix kernel for small matrices, n = 9 .. 15."""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tensornetworks_amd import backend

dev = torch.device("cuda", 0)
for n in range(8, 16):
    N = 1 << n
    g = torch.Generator(device="cpu").manual_seed(n)
    S = torch.randn((N, n), generator=g, dtype=torch.float64).to(dev)
    q = torch.rand(N, generator=g, dtype=torch.float64).to(dev)
    q /= q.sum()
    K = backend.stein_gram(S, n, 1.0, ld=backend.gram_ld(n))
    Kd = backend.stein_gram(S, n, 1.0)

    def clock(fn):
        for _ in range(5):
            fn()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
        for a, b in ev:
            a.record(); fn(); b.record()
        torch.cuda.synchronize()
        return float(np.median([a.elapsed_time(b) for a, b in ev])) * 1e3

    t_sym = clock(lambda: backend.stein_quadform_sym(K, q, n))
    t_symd = clock(lambda: backend.stein_quadform_sym(Kd, q, n))
    t_full = clock(lambda: backend.stein_quadform(Kd, q, n, want_y=True))
    print(f"n={n}: sym (padded) {t_sym:.1f} us, sym (dense) {t_symd:.1f} us, full {t_full:.1f} us", flush=True)
