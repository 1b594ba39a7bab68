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
th = np.random.default_rng(5).uniform(-np.pi, np.pi, (1, oc.num_params(ans, n, L)))
ref = oc.probs(ans, n, L, th[0])
B = 33
q = be.circuit_probs(ans, n, L, torch.as_tensor(np.repeat(th, B, 0), device=dev)).cpu().numpy()
d = np.abs(q[32] - ref)
bad = np.nonzero(d > 1e-12)[0]
print("bad elements", len(bad), "of", 1 << n, "sum q32", q[32].sum())
if len(bad):
    print("first bad", [format(int(i), f"0{n}b") for i in bad[:8]])
    print("last bad", [format(int(i), f"0{n}b") for i in bad[-4:]])
    orb = np.bitwise_or.reduce(bad); andb = np.bitwise_and.reduce(bad)
    print("or ", format(int(orb), f"0{n}b"), "and", format(int(andb), f"0{n}b"))
    # is q[32] a permutation of ref?
    print("sorted equal:", np.allclose(np.sort(q[32]), np.sort(ref), rtol=0, atol=1e-16))
    # does q[32] equal ref with some index bit flipped?
    for bit in range(n):
        idx = np.arange(1 << n) ^ (1 << bit)
        if np.allclose(q[32], ref[idx], atol=1e-14): print("equals ref with bit", bit, "flipped")
