import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tensornetworks_amd import backend as be
from oracle import circuit as oc
dev = torch.device("cuda:0")
ans, n, L, kb = "basic", 14, 1, 11
be.set_option(dev, "tile_bits", kb)
be.set_option(dev, "fast_workgroups_per_cu", 1)
be.set_option(dev, "direct_stages", 0)
B = 33
th = np.random.default_rng(5).uniform(-np.pi, np.pi, (B, oc.num_params(ans, n, L)))
ref = {b: oc.probs(ans, n, L, th[b]) for b in (0, 1, 31, 32)}
q = be.circuit_probs(ans, n, L, torch.as_tensor(th, device=dev)).cpu().numpy()
for b in (0, 1, 31, 32):
    print("row", b, "err vs own", np.abs(q[b] - ref[b]).max(), "vs ref32", np.abs(q[b] - ref[32]).max(), "sum", q[b].sum())
