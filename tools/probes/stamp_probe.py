"""Where a wave of the fast circuit kernel spends its cycles (diagnostic build -DBORNVI_STAMPS=1 via BORNVI_LIB): phase
totals of wave 0 of every workgroup over one parameter-shift batch, as fractions of the stamped time."""
import ctypes as C
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tensornetworks_amd import backend, _ext

dev = torch.device("cuda", 0)
names = ["wait for the prefetched tile", "tile -> LDS / direct first stage", "issue prefetch", "stages", "tile out",
         "end of trip (matrices, barrier)", "loop overhead", "pipeline-start trip"]
for n, L in ((16, 6), (20, 8)):
    P = backend.num_params("hardware_efficient", n, L)
    g = torch.Generator().manual_seed(0)
    theta = (0.1 * torch.randn(P, generator=g, dtype=torch.float32)).double().to(dev)
    out = torch.empty((2 * P + 1, 1 << n), dtype=torch.float64, device=dev)
    h = _ext.handle_for(dev)
    buf = (C.c_ulonglong * 16)()
    for _ in range(2):
        backend.paramshift_probs("hardware_efficient", n, L, theta, 0, P, include_base=True, out=out)
    h.call("bornvi_debug_circuit_stamps", buf)          # clear
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    backend.paramshift_probs("hardware_efficient", n, L, theta, 0, P, include_base=True, out=out)
    e1.record()
    torch.cuda.synchronize()
    h.call("bornvi_debug_circuit_stamps", buf)
    tot = sum(buf[i] for i in range(8))
    print(f"n={n} L={L}: batch {e0.elapsed_time(e1):.3f} ms (stamped build); {buf[8]} workgroup-launches; "
          f"{tot / max(1, buf[8]) / 1e3:.1f} k cycles per workgroup-launch")
    for i in range(8):
        print(f"   {names[i]:36s} {100.0 * buf[i] / max(1, tot):5.1f} %")
    del out
    backend.release_workspaces()
    torch.cuda.empty_cache()
