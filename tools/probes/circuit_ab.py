"""A/B of circuit-engine builds (BORNVI_LIB): time of the full parameter-shift batch at n = 16, L = 6 and n = 20, L = 8,
and a checksum of the rows (variants that keep the arithmetic must print the same checksum)."""
import hashlib
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tensornetworks_amd import backend

dev = torch.device("cuda", 0)
tag = os.path.basename(os.environ.get("BORNVI_LIB", "default"))
for kv in filter(None, os.environ.get("BORNVI_OPTS", "").split(",")):     # e.g. BORNVI_OPTS=alternate_walk=1
    name, value = kv.split("=")
    backend.set_option(dev, name, int(value))
    tag += f" {name}={value}"
for n, L in ((16, 6), (20, 8)):
    P = backend.num_params("hardware_efficient", n, L)
    g = torch.Generator().manual_seed(0)
    theta = (0.1 * torch.randn(P, generator=g, dtype=torch.float32)).double().to(dev)
    out = torch.empty((2 * P + 1, 1 << n), dtype=torch.float64, device=dev)
    fn = lambda: backend.paramshift_probs("hardware_efficient", n, L, theta, 0, P, include_base=True, out=out)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    reps = 20 if n == 16 else 5
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = [a.elapsed_time(b) for a, b in ev]
    h = hashlib.sha256(out[:: max(1, (2 * P + 1) // 37)].cpu().numpy().tobytes()).hexdigest()[:16]
    print(f"{tag} n={n} L={L}: median {np.median(ts):.3f} ms  min {min(ts):.3f}  rows sha {h}", flush=True)
    del out
    backend.release_workspaces()
    torch.cuda.empty_cache()
