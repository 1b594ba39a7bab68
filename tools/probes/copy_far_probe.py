"""Probe: rate of a device-to-device copy (read one buffer, write another: the circuit engine's traffic mix) against the
distance, in allocation order, between the two buffers: destinations allocated behind 0, 1, 2, ... spacers."""
import os
import sys
import numpy as np
import torch

dev = torch.device("cuda", 0)
size = int(os.environ.get("SIZE_GIB", "4")) << 30
gib = int(os.environ.get("SPACER_GIB", "16"))
nsp = int(os.environ.get("SPACERS", "10"))
for trial in range(int(os.environ.get("TRIALS", "2"))):
    A = torch.empty(size, dtype=torch.uint8, device=dev).view(torch.float64)
    A.normal_()
    inner = torch.empty(2 * size, dtype=torch.uint8, device=dev).view(torch.float64)     # both halves in ONE allocation
    cands, spacers = [], []
    for i in range(nsp + 1):
        cands.append(torch.empty(size, dtype=torch.uint8, device=dev).view(torch.float64))
        if i < nsp:
            spacers.append(torch.empty(gib << 30, dtype=torch.uint8, device=dev))
    del spacers
    torch.cuda.empty_cache()

    def rate(src, dst):
        for _ in range(2):
            dst.copy_(src)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(7)]
        for a, b in ev:
            a.record(); dst.copy_(src); b.record()
        torch.cuda.synchronize()
        return 2 * size / np.median([a.elapsed_time(b) for a, b in ev]) / 1e6

    h = inner.numel() // 2
    inner[:h].copy_(A)
    print(f"trial {trial}: A at {A.data_ptr():#x}; halves of one allocation {rate(inner[:h], inner[h:]):.0f} GB/s; A -> destination behind k spacers of {gib} GiB: "
          + " ".join(f"{rate(A, c):.0f}" for c in cands) + " GB/s", flush=True)
    del cands, inner, A
    torch.cuda.empty_cache()
