"""Probe: K_p in memory allocated with hipExtMallocWithFlags (uncached / fine-grained / contiguous) against ordinary
hipMalloc memory: the contraction streams K_p once, so it does not need the caches -- do they do better left to the partial
sums?  ALLOCS copies per kind, held at once; the C ABI is called with raw pointers."""
import ctypes as C
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tensornetworks_amd import backend, _ext

n = 16
N = 1 << n
dev = torch.device("cuda", 0)
g = torch.Generator(device="cpu").manual_seed(0)
S = torch.randn((N, n), generator=g, dtype=torch.float64).to(dev)
q = torch.rand(N, generator=g, dtype=torch.float64).to(dev)
q /= q.sum()
hip = C.CDLL("libamdhip64.so")
hip.hipExtMallocWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipFree.argtypes = [C.c_void_p]
h = _ext.handle_for(dev)
ld = backend.gram_ld(n)
nbytes = N * ld * 8
ws = torch.empty(backend.stein_sym_workspace_bytes(dev, n), dtype=torch.uint8, device=dev)
y = torch.empty(N, dtype=torch.float64, device=dev)
k2 = torch.empty(1, dtype=torch.float64, device=dev)
ref = None
for kind, flag in (("default", 0), ("uncached", 3), ("finegrained", 1), ("contiguous", 4), ("default", 0)):
    ptrs, times = [], []
    for a in range(int(os.environ.get("ALLOCS", "3"))):
        p = C.c_void_p()
        rc = hip.hipExtMallocWithFlags(C.byref(p), nbytes, flag)
        if rc != 0:
            print(f"{kind}: hipExtMallocWithFlags failed ({rc})", flush=True)
            break
        ptrs.append(p)
        h.call("bornvi_stein_gram_build_rows_ld", n, 1.0, C.c_void_p(S.data_ptr()), 0, N, p, ld, _ext.stream_ptr(dev))

        def run():
            h.call("bornvi_stein_quadform_sym_ld", n, p, ld, C.c_void_p(q.data_ptr()), C.c_void_p(k2.data_ptr()),
                   C.c_void_p(y.data_ptr()), C.c_void_p(ws.data_ptr()), ws.numel(), _ext.stream_ptr(dev))
        for _ in range(3):
            run()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(12)]
        for e0, e1 in ev:
            e0.record(); run(); e1.record()
        torch.cuda.synchronize()
        times.append(round(float(np.median([e0.elapsed_time(e1) for e0, e1 in ev])), 4))
        if ref is None:
            ref = y.clone()
        assert torch.equal(ref, y)
    print(f"{kind:12s} ms per allocation {times}", flush=True)
    torch.cuda.synchronize()
    for p in ptrs:
        hip.hipFree(p)
