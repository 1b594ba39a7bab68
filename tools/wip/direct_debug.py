"""A/B of the direct first / last stages of the fast circuit kernel against the oracle (debug aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tensornetworks_amd import backend as be
from oracle import circuit as oc
dev = torch.device("cuda:0")
cases = [("all_to_all", 14, 3, 13), ("hardware_efficient", 14, 3, 13), ("all_to_all", 13, 2, 11), ("basic", 15, 2, 12), ("hardware_efficient", 16, 2, 11)]
for ans, n, L, kb in cases:
    be.set_option(dev, "tile_bits", kb)
    rng = np.random.default_rng(7 * n + L + kb)
    th = rng.uniform(-np.pi, np.pi, (3, oc.num_params(ans, n, L)))
    ref = np.stack([oc.probs(ans, n, L, t) for t in th])
    for mask in (0, 1, 2, 3):
        be.set_option(dev, "direct_stages", mask)
        q = be.circuit_probs(ans, n, L, torch.as_tensor(th, device=dev)).cpu().numpy()
        err = np.abs(q - ref)
        print(ans, n, L, kb, "mask", mask, "max err", err.max(), "bad", int((err > 1e-12).sum()), "per circuit", [float(e.max()) for e in err])
