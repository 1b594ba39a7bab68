"""Do the HBM-bound contraction and the compute-bound circuit kernel run concurrently on two streams?
A = symmetric contraction at n = 16 (17.7 GB from HBM); B = a batch of n = 12 circuits (one tile per state, no
inter-pass HBM traffic) sized to take about as long; C = the n = 16 circuit batch of a step (8.8 GB of HBM traffic)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tensornetworks_amd import backend as be
from tensornetworks_amd.bayesian_network import synthetic_network
from tensornetworks_amd.stein_utils import score_matrix
dev = torch.device("cuda:0")
n = 16
bn, lat, obs, x = synthetic_network(n, seed=0)
S = score_matrix(bn, x, lat, device=dev)
K = be.stein_gram(S, n, 1.0)
q = torch.rand(1 << n, dtype=torch.float64, device=dev); q /= q.sum()
th12 = torch.rand((int(sys.argv[1]) if len(sys.argv) > 1 else 6000, 3 * 12 * 4), dtype=torch.float64, device=dev)
th16 = torch.rand(3 * 16 * 6, dtype=torch.float64, device=dev)
sA, sB = torch.cuda.Stream(dev), torch.cuda.Stream(dev)

def A():
    be.stein_quadform_sym(K, q, n)
def B():
    be.circuit_probs("hardware_efficient", 12, 4, th12)
def C():
    be.paramshift_probs("hardware_efficient", 16, 6, th16, 0, 288, include_base=True)

def timed(fns):
    for f, s in fns:                      # warm
        with torch.cuda.stream(s): f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        for f, s in fns:
            with torch.cuda.stream(s): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 5 * 1e3

a, b, c = timed([(A, sA)]), timed([(B, sB)]), timed([(C, sB)])
print(f"A alone {a:.2f} ms   B alone {b:.2f} ms   C alone {c:.2f} ms")
print(f"A || B  {timed([(A, sA), (B, sB)]):.2f} ms  (sum {a + b:.2f}, max {max(a, b):.2f})")
print(f"A || C  {timed([(A, sA), (C, sB)]):.2f} ms  (sum {a + c:.2f}, max {max(a, c):.2f})")
for w in (1, 2, 3):
    be.set_option(dev, "fast_workgroups_per_cu", w)
    c = timed([(C, sB)]); b = timed([(B, sB)])
    print(f"circuit kernels at {w} wg/cu: C alone {c:.2f}  A || C {timed([(A, sA), (C, sB)]):.2f}   B alone {b:.2f}  A || B {timed([(A, sA), (B, sB)]):.2f}")
