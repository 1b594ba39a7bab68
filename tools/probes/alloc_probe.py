"""Probe: is a "slow" K_p allocation slow for every access shape (a property of the memory) or only for the symmetric
contraction?  Several 32 GiB matrices held at once; per matrix: symmetric contraction, full-matrix contraction,
torch.sum (linear read), and the pointer."""
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


def clock(fn, reps=9):
    for _ in range(2):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for e0, e1 in ev:
        e0.record(); fn(); e1.record()
    torch.cuda.synchronize()
    return float(np.median([e0.elapsed_time(e1) for e0, e1 in ev]))


held = []
for a in range(int(os.environ.get("ALLOCS", "6"))):
    K = backend.stein_gram(S, n, 1.0)
    held.append(K)
    t_sym = clock(lambda: backend.stein_quadform_sym(K, q, n))
    t_full = clock(lambda: backend.stein_quadform(K, q, n, want_y=True))
    t_sum = clock(lambda: K.sum())
    t_tri = clock(lambda: K[: N // 2].sum())
    print(f"alloc {a} ptr {K.data_ptr():#x}: sym {t_sym:.3f} ms  full {t_full:.3f} ms ({8.0 * N * N / t_full / 1e6:.0f} GB/s)  "
          f"sum {t_sum:.3f} ms ({8.0 * N * N / t_sum / 1e6:.0f} GB/s)  sum of first half {t_tri:.3f} ms", flush=True)
free_b, total_b = torch.cuda.mem_get_info(dev)
print(f"free {free_b / 2**30:.1f} GiB of {total_b / 2**30:.1f} GiB")
