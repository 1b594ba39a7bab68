"""Probe: symmetric contraction time at n = 16 as a function of the row pitch of K_p and of the allocation it lives in
(several matrices held at once: the round-1 "placement" effect), for the library named by BORNVI_LIB."""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tensornetworks_amd import backend

n = int(os.environ.get("N", "16"))
N = 1 << n
dev = torch.device("cuda", 0)
g = torch.Generator(device="cpu").manual_seed(0)
S = torch.randn((N, n), generator=g, dtype=torch.float64).to(dev)
q = torch.rand(N, generator=g, dtype=torch.float64).to(dev)
q /= q.sum()
tag = os.path.basename(os.environ.get("BORNVI_LIB", "default"))
ref = None
for pad in [int(x) for x in os.environ.get("PADS", "0,32,64,160").split(",")]:
    held = []
    times = []
    for a in range(int(os.environ.get("ALLOCS", "3"))):
        K = backend.stein_gram(S, n, 1.0, ld=N + pad)
        held.append(K)
        for _ in range(3):
            k2, y = backend.stein_quadform_sym(K, q, n)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(15)]
        for e0, e1 in ev:
            e0.record(); k2, y = backend.stein_quadform_sym(K, q, n); e1.record()
        torch.cuda.synchronize()
        times.append(float(np.median([e0.elapsed_time(e1) for e0, e1 in ev])))
        if ref is None:
            ref = (k2.clone(), y.clone())
        else:
            assert os.environ.get("NOCHECK") or (torch.equal(y, ref[1]) and torch.equal(k2, ref[0])), "results differ between pitches / allocations"
    bytes_ = 4.0 * N * (N + 32)
    print(f"{tag} n={n} pad={pad:4d} ms per allocation {[round(t, 4) for t in times]}  best {bytes_ / min(times) / 1e6:.0f} GB/s  worst {bytes_ / max(times) / 1e6:.0f} GB/s", flush=True)
    del held, K
    torch.cuda.empty_cache()
