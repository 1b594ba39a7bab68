"""What does a plain read-only stream over 32 GiB reach on this box?  (reference point for the contraction's GB/s)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tensornetworks_amd import backend as be
from tensornetworks_amd.bayesian_network import synthetic_network
from tensornetworks_amd.stein_utils import score_matrix
dev = torch.device("cuda:0")
n = 16
N = 1 << n
bn, lat, obs, x = synthetic_network(n, seed=0)
S = score_matrix(bn, x, lat, device=dev)
K = be.stein_gram(S, n, 1.0)
q = torch.full((N,), 1.0 / N, dtype=torch.float64, device=dev)


def clock(fn, reps=5):
    fn(); fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


t = clock(lambda: torch.sum(K))
print(f"torch.sum over K (34.36 GB read): {t:.3f} ms = {34.36 / t:.2f} TB/s")
t = clock(lambda: torch.mv(K, q))
print(f"torch.mv K q (34.36 GB read, rocBLAS gemv): {t:.3f} ms = {34.36 / t:.2f} TB/s")
half = K[: N // 2]
t = clock(lambda: torch.sum(half))
print(f"torch.sum over the first half of K (17.18 GB): {t:.3f} ms = {17.18 / t:.2f} TB/s")
t = clock(lambda: be.stein_quadform_sym(K, q, n))
print(f"symmetric contraction (17.19 GB algorithmic, 18.29 GB measured): {t:.3f} ms = {17.19 / t:.2f} / {18.29 / t:.2f} TB/s")
t = clock(lambda: be.stein_quadform(K, q, n, want_y=True))
print(f"full-matrix contraction (34.36 GB): {t:.3f} ms = {34.36 / t:.2f} TB/s")
