"""Times the symmetric contraction (quadform_sym + reduce) at n = 16; BORNVI_LIB selects a library variant."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tensornetworks_amd import backend
from tensornetworks_amd.bayesian_network import synthetic_network
from tensornetworks_amd.stein_utils import score_matrix
n = 16
dev = torch.device("cuda:0")
bn, lat, obs, x = synthetic_network(n, seed=0)
S = score_matrix(bn, x, lat, device=dev)
K = backend.stein_gram(S, n, 1.0)
q = torch.rand(1 << n, dtype=torch.float64, device=dev); q /= q.sum()
for _ in range(3):
    k2, y = backend.stein_quadform_sym(K, q, n)
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
for a, b in ev:
    a.record(); k2, y = backend.stein_quadform_sym(K, q, n); b.record()
torch.cuda.synchronize()
ms = sorted(a.elapsed_time(b) for a, b in ev)[len(ev) // 2]
print(os.environ.get("BORNVI_LIB", "default"), f"{ms:.3f} ms  {4 * (1 << n) * ((1 << n) + 32) / ms / 1e6:.0f} GB/s  ksd2 {k2.item():.15e}")
