"""Timing of the batched contraction Y = K_p Q^T at n = 16 (BASELINE config 3's matrix) for B = 577 (one
parameter-shift batch) and smaller B, matrix cores vs the looped GEMV; JSON line for profiles/."""
import json
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
K = backend.stein_gram(S, n, 1.0)
out = {"n": n, "fp64_mfma_peak_tflops": 78.6, "hbm_peak_gbs": 8000.0, "cases": []}
for B in [int(x) for x in os.environ.get("BS", "8,64,128,577").split(",")]:
    Q = torch.rand((B, N), generator=g, dtype=torch.float64).to(dev)
    Q /= Q.sum(dim=1, keepdim=True)
    res = {}
    for mode in (1, 0):
        if mode == 0 and B > 64:
            continue
        backend.set_engine_option(dev, "batched_quadform", mode)
        k2, Y = backend.stein_quadform(K, Q, n)
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(3)]
        for e0, e1 in ev:
            e0.record(); backend.stein_quadform(K, Q, n); e1.record()
        torch.cuda.synchronize()
        res[mode] = float(np.median([e0.elapsed_time(e1) for e0, e1 in ev]))
    backend.set_engine_option(dev, "batched_quadform", 1)
    flop = 2.0 * N * N * B
    floor_ms = max(8.0 * N * N / 8000e9, flop / 78.6e12) * 1e3
    case = {"B": B, "mfma_ms": round(res[1], 3), "looped_gemv_ms": round(res[0], 3) if 0 in res else None,
            "tflops": round(flop / (res[1] * 1e-3) / 1e12, 2), "frac_of_fp64_mfma_peak": round(flop / (res[1] * 1e-3) / 78.6e12, 4),
            "roofline_floor_ms": round(floor_ms, 3), "ratio_to_floor": round(res[1] / floor_ms, 3)}
    out["cases"].append(case)
    print(case, flush=True)
print(json.dumps(out))
