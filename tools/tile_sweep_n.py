"""circuit time of a full parameter-shift batch vs tile size for n = 17 .. 20 (which tile_bits_multi should 'auto' pick?)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tensornetworks_amd import backend as be
dev = torch.device("cuda:0")
L = 6
for n in (14, 15, 16, 17, 18, 19, 20):
    P = 3 * n * L
    th = torch.rand(P, dtype=torch.float64, device=dev)
    row = []
    for k in (11, 12, 13):
        be.set_option(dev, "tile_bits_multi", k)
        out = torch.empty((2 * P + 1, 1 << n), dtype=torch.float64, device=dev)
        for _ in range(2):
            be.paramshift_probs("hardware_efficient", n, L, th, 0, P, include_base=True, out=out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            be.paramshift_probs("hardware_efficient", n, L, th, 0, P, include_base=True, out=out)
        torch.cuda.synchronize()
        row.append(round((time.perf_counter() - t0) / 3 * 1e3, 2))
        del out
    print(f"n={n} L={L} circuits={2 * P + 1}: k=11/12/13 -> {row} ms", flush=True)
be.set_option(dev, "tile_bits_multi", 0)
