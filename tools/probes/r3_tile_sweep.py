#!/usr/bin/env python3
"""Tile-size sweep of the 8-amplitude kernel (reg_wires = 3): ms per full parameter-shift batch (hardware_efficient, L as given)
for 2^11 / 2^12 / 2^13 tiles, n as given:  r3_tile_sweep.py n,L [n,L ...]"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from tensornetworks_amd import backend as be   # noqa: E402
from oracle import circuit as oc                # noqa: E402

dev = torch.device("cuda", 0)
be.set_option(dev, "reg_wires", 3)
for a in sys.argv[1:]:
    n, L = (int(x) for x in a.split(","))
    P = oc.num_params("hardware_efficient", n, L)
    th = torch.as_tensor(0.1 * np.random.default_rng(0).standard_normal(P), dtype=torch.float64, device=dev)
    row = {"n": n, "L": L}
    for k in (11, 12, 13):
        be.set_option(dev, "tile_bits_multi", k)
        out = be.paramshift_probs("hardware_efficient", n, L, th, 0, P, include_base=True)
        torch.cuda.synchronize()
        ts = []
        for _ in range(7 if n < 19 else 3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = be.paramshift_probs("hardware_efficient", n, L, th, 0, P, include_base=True, out=out)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        row[f"k{k}_ms"] = round(float(np.median(ts)), 4)
        del out
        be.release_workspaces()
        torch.cuda.empty_cache()
    print(json.dumps(row), flush=True)
