"""Where does the 2.85 <-> 3.35 ms spread of the n = 16 symmetric contraction come from?  Several copies of K_p held
at once (timed round robin), copies at shifted offsets inside a larger buffer, and fresh allocations."""
import sys, os, time
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
q = torch.rand(N, dtype=torch.float64, device=dev); q /= q.sum()


def t_contract(K, reps=6):
    for _ in range(2):
        be.stein_quadform_sym(K, q, n)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); be.stein_quadform_sym(K, q, n); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return round(sum(ts) / len(ts), 3)


Ks = [be.stein_gram(S, n, 1.0) for _ in range(3)]
for rnd in range(2):
    print("held together:", [(hex(K.data_ptr()), t_contract(K)) for K in Ks], flush=True)
del Ks[1:]
big = torch.empty(N * N + (64 << 20), dtype=torch.float64, device=dev)
for off_bytes in (0, 4096, 65536, 1 << 20, 2 << 20, 3 << 20, 16 << 20, 256 << 20):
    v = big[off_bytes // 8: off_bytes // 8 + N * N].view(N, N)
    v.copy_(Ks[0])
    print(f"copy at base+{off_bytes:>10d} (0x{v.data_ptr():x}): {t_contract(v)} ms", flush=True)
