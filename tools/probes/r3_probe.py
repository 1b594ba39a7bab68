#!/usr/bin/env python3
"""Same-box A/B of the two persistent circuit kernels: 16 amplitudes per thread (reg_wires = 4, circuit_pass_fast_kernel)
against 8 per thread (reg_wires = 3, circuit_pass_r3_kernel).  Rows must agree to rounding (the stage cut differs, so the
order of the fused gates differs: not bitwise); times are per parameter-shift batch."""
import hashlib
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from tensornetworks_amd import backend as be   # noqa: E402
from oracle import circuit as oc                # noqa: E402

dev = torch.device("cuda", 0)


def batch_time(ansatz, n, L, reps=5):
    P = oc.num_params(ansatz, n, L)
    th = torch.as_tensor(0.1 * np.random.default_rng(0).standard_normal(P), dtype=torch.float64, device=dev)
    out = None
    out = be.paramshift_probs(ansatz, n, L, th, 0, P, include_base=True, out=out)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = be.paramshift_probs(ansatz, n, L, th, 0, P, include_base=True, out=out)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts)), float(np.min(ts)), out


def main():
    cases = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]] or [(12, 4), (14, 3), (16, 6)]
    res = []
    for n, L in cases:
        row = {"n": n, "L": L}
        outs = {}
        for r in (4, 3):
            be.set_option(dev, "reg_wires", r)
            be.set_option(dev, "read_map", int(os.environ.get("R3_READ_MAP", "1")) if r == 3 else 0)
            med, mn, out = batch_time("hardware_efficient", n, L)
            row[f"r{r}_ms"] = round(med, 4)
            row[f"r{r}_min_ms"] = round(mn, 4)
            sums = out.sum(dim=1)
            row[f"r{r}_sum_err"] = float((sums - 1).abs().max())
            outs[r] = out if n <= 16 else out[:8].clone()
            if n > 16:
                del out
                be.release_workspaces()
        d = (outs[3][: outs[4].shape[0]] - outs[4]).abs().max().item()
        row["max_abs_diff_r3_r4"] = d
        res.append(row)
        print(json.dumps(row), flush=True)
        del outs
        be.release_workspaces()
        torch.cuda.empty_cache()
    be.set_option(dev, "reg_wires", 4)
    be.set_option(dev, "read_map", 0)


if __name__ == "__main__":
    main()
