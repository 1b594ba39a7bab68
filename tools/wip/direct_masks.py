import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tensornetworks_amd import backend as be
from oracle import circuit as oc
dev = torch.device("cuda:0")
cases = [("all_to_all", 14, 3, 13, 3), ("hardware_efficient", 16, 2, 11, 3), ("basic", 15, 2, 12, 3), ("hardware_efficient", 14, 2, 11, 100)]
for ans, n, L, kb, B in cases:
    be.set_option(dev, "tile_bits", kb)
    be.set_option(dev, "fast_workgroups_per_cu", 1 if B > 10 else 0)
    rng = np.random.default_rng(7 * n + L + kb)
    th2 = rng.uniform(-np.pi, np.pi, (3, oc.num_params(ans, n, L)))
    pick = rng.integers(0, 3, B)
    ref = np.stack([oc.probs(ans, n, L, t) for t in th2])
    for mask in (0, 1, 2, 3):
        be.set_option(dev, "direct_stages", mask)
        q = be.circuit_probs(ans, n, L, torch.as_tensor(th2[pick], device=dev)).cpu().numpy()
        err = np.abs(q - ref[pick])
        print(ans, n, L, kb, "B", B, "mask", mask, "max err %.3e" % err.max(), "bad rows", int((err.max(axis=1) > 1e-12).sum()))
