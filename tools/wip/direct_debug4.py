import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tensornetworks_amd import backend as be
from oracle import circuit as oc
dev = torch.device("cuda:0")
ans, n, L, kb = "hardware_efficient", 14, 2, 11
be.set_option(dev, "tile_bits", kb)
be.set_option(dev, "fast_workgroups_per_cu", 1)
th = np.random.default_rng(5).uniform(-np.pi, np.pi, (1, oc.num_params(ans, n, L)))
ref = oc.probs(ans, n, L, th[0])
for B in (33, 64):
  for dbgf in (0, 256):
    be.set_option(dev, "debug_flags", dbgf)
    for mask in (0,):
        be.set_option(dev, "direct_stages", mask)
        q = be.circuit_probs(ans, n, L, torch.as_tensor(np.repeat(th, B, 0), device=dev)).cpu().numpy()
        err = np.abs(q - ref).max(axis=1)
        bad = np.nonzero(~(err < 1e-12))[0]
        print("B", B, "dbg", dbgf, "mask", mask, "bad rows", len(bad), bad[:20], "max err", np.nanmax(err) if len(bad) else err.max(), "nan rows", int(np.isnan(q).any(axis=1).sum()))
