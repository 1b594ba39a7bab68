"""Times the dense Gram build (bornvi_stein_gram_build) at n = 16 on repeated calls into the same buffer."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tensornetworks_amd import backend
from tensornetworks_amd.bayesian_network import synthetic_network
from tensornetworks_amd.stein_utils import score_matrix
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda:0")
bn, lat, obs, x = synthetic_network(n, seed=0)
S = score_matrix(bn, x, lat, device=dev)
for rep in range(4):
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    K = backend.stein_gram(S, n, 1.0)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b)
    print(f"rep {rep}: {ms:.2f} ms  ({8 * 4**n / ms / 1e6:.0f} GB/s written)")
    if rep < 3: del K
# memset reference: same bytes
K.zero_(); torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(); K.zero_(); b.record(); torch.cuda.synchronize()
print(f"zero_: {a.elapsed_time(b):.2f} ms")
