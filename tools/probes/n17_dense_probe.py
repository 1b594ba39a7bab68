"""The largest dense Gram the API admits (n = 17: 2^17 x 2^17 doubles = 137 GB of the 288 GB) end to end on the GPU: build
(gram_tables_kernel<17>), bitwise symmetry of sampled blocks, the symmetric contraction against the matrix-free one."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from tensornetworks_amd import backend
from tensornetworks_amd.bayesian_network import synthetic_network
from tensornetworks_amd.stein_utils import score_matrix

n = 17
dev = torch.device("cuda:0")
N = 1 << n
bn, lat, obs, x = synthetic_network(n, seed=0)
S = score_matrix(bn, x, lat, device=dev)
ld = backend.gram_ld(n)
print("free GB before", torch.cuda.mem_get_info()[0] / 1e9, "ld", ld, flush=True)
t0 = time.perf_counter()
K = backend.stein_gram(S, n, 1.0, ld=ld)
torch.cuda.synchronize()
print(f"first build {time.perf_counter() - t0:.3f} s", flush=True)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(); backend.stein_gram(S, n, 1.0, out=K); b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b)
print(f"build {ms:.2f} ms = {8.0 * N * N / ms / 1e6:.0f} GB/s written", flush=True)
for (r0, c0) in ((0, 0), (1000, 90000), (N - 4096, 0), (65536, 65536 + 4096)):
    blk = K[r0:r0 + 4096, c0:c0 + 4096]
    mir = K[c0:c0 + 4096, r0:r0 + 4096].T
    assert torch.equal(blk, mir), (r0, c0)
g = torch.Generator(device="cpu").manual_seed(3)
q = torch.rand(N, generator=g, dtype=torch.float64).to(dev)
q /= q.sum()
k_sym, y_sym = backend.stein_quadform_sym(K, q, n)
a.record(); k_sym, y_sym = backend.stein_quadform_sym(K, q, n); b.record(); torch.cuda.synchronize()
print(f"symmetric contraction {a.elapsed_time(b):.2f} ms", flush=True)
k_kron, y_kron = backend.stein_matvec_kron(S, q, n, 1.0)
scale = float(y_kron.abs().max())
err_y = float((y_sym - y_kron).abs().max()) / scale
err_k = abs(float(k_sym) - float(k_kron)) / abs(float(k_kron))
print(f"dense vs matrix-free: max |dy| / max|y| = {err_y:.2e}, ksd2 rel {err_k:.2e}", flush=True)
assert err_y < 1e-10 and err_k < 1e-10
print("n = 17 dense ok")
