"""Runs the two matrix-core kernels of the path at n = 16 (for a rocprofv3 --pmc pass): the Gram build on the trainer's
padded pitch and the batched contraction Y = K_p Q^T at B = 128, three launches each."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tensornetworks_amd import backend
from tensornetworks_amd.bayesian_network import synthetic_network
from tensornetworks_amd.stein_utils import score_matrix
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda:0")
bn, lat, obs, x = synthetic_network(n, seed=0)
S = score_matrix(bn, x, lat, device=dev)
ld = backend.gram_ld(n)
K = backend.stein_gram(S, n, 1.0, ld=ld)
for rep in range(3):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); backend.stein_gram(S, n, 1.0, out=K); b.record(); torch.cuda.synchronize()
    print(f"gram build rep {rep}: {a.elapsed_time(b):.2f} ms")
Q = torch.rand((128, 1 << n), dtype=torch.float64, device=dev)
Q /= Q.sum(dim=1, keepdim=True)
for rep in range(3):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); backend.stein_quadform(K, Q, n, want_y=False); b.record(); torch.cuda.synchronize()
    print(f"batched quadform B=128 rep {rep}: {a.elapsed_time(b):.2f} ms")
