import sys, numpy as np, torch
sys.path.insert(0, ".")
from tensornetworks_amd import backend as be
from oracle import circuit as oc
dev = torch.device("cuda", 0)
be.set_option(dev, "reg_wires", 3)
bad = 0
for ansatz in oc.ANSATZ_TYPES:
    for n, L, kb in [(9, 2, 0), (10, 2, 0), (12, 3, 0), (13, 2, 0), (12, 2, 9), (13, 2, 11), (14, 3, 11), (14, 3, 13), (15, 2, 12), (16, 2, 13)]:
        be.set_option(dev, "tile_bits", kb if kb else 13)
        rng = np.random.default_rng(7 * n + L + kb)
        th = rng.uniform(-np.pi, np.pi, (3, oc.num_params(ansatz, n, L)))
        q = be.circuit_probs(ansatz, n, L, torch.as_tensor(th, dtype=torch.float64, device=dev)).cpu().numpy()
        err = max(np.abs(q[b] - oc.probs(ansatz, n, L, th[b])).max() for b in range(3))
        print(ansatz, n, L, kb, "max err", err, flush=True)
        bad += err > 1e-13
print("BAD", bad)
sys.exit(1 if bad else 0)
