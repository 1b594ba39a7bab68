import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tensornetworks_amd import backend as be
from oracle import circuit as oc
dev = torch.device("cuda:0")
# one-layer circuits: pass 0 (INIT) does almost everything -> errors of the direct-out stage are visible directly
for ans, n, L, kb in [("hardware_efficient", 16, 2, 11), ("basic", 15, 2, 12), ("hardware_efficient", 14, 2, 11)]:
    be.set_option(dev, "tile_bits", kb)
    th = np.random.default_rng(3).uniform(-np.pi, np.pi, (1, oc.num_params(ans, n, L)))
    be.set_option(dev, "direct_stages", 0)
    q0 = be.circuit_probs(ans, n, L, torch.as_tensor(th, device=dev)).cpu().numpy()[0]
    be.set_option(dev, "direct_stages", 2)
    q2 = be.circuit_probs(ans, n, L, torch.as_tensor(th, device=dev)).cpu().numpy()[0]
    d = np.abs(q0 - q2)
    idx = np.argsort(-d)[:12]
    print(ans, n, L, kb, "max", d.max(), "nbad", int((d > 1e-15).sum()), "sum q2", q2.sum())
    for i in idx:
        print("   ", format(int(i), f"0{n}b"), q0[i], q2[i])
    # is q2 a permutation of q0?
    print("   sorted equal:", np.allclose(np.sort(q0), np.sort(q2), rtol=0, atol=1e-18))
    r = (q2 / np.maximum(q0, 1e-300))[q0 > 1e-8]
    print("   ratio q2/q0 - 1: min %.3e max %.3e mean %.3e" % (r.min() - 1, r.max() - 1, r.mean() - 1))
