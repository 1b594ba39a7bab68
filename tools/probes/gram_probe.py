"""Time of the Gram build (n = 16 by default, padded pitch) for the library in BORNVI_LIB, HIP events, plus a plain
fill of the same buffer (torch) as the write-only reference.   python tools/probes/gram_probe.py [n] [reps]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from tensornetworks_amd import backend

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
N = 1 << n
g = torch.Generator(device="cpu").manual_seed(1)
S = (torch.rand((N, n), generator=g, dtype=torch.float64) * 2 - 1).to(dev)
ld = backend.gram_ld(n) if hasattr(backend, "gram_ld") else N
buf = torch.empty((N, ld), dtype=torch.float64, device=dev)
K = buf[:, :N]

def timed(fn):
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return ts

backend.stein_gram(S, n, 1.0, out=K); torch.cuda.synchronize()
t = timed(lambda: backend.stein_gram(S, n, 1.0, out=K))
import hashlib
digest = hashlib.sha256(K[:, :].contiguous().cpu().numpy().tobytes()).hexdigest()[:16] if n <= 14 else hashlib.sha256(K[:2048].contiguous().cpu().numpy().tobytes() + K[N - 2048:].contiguous().cpu().numpy().tobytes()).hexdigest()[:16]
sym = bool(torch.equal(K[:4096, :4096], K[:4096, :4096].T)) and bool(torch.equal(K[:1024, N - 1024:], K[N - 1024:, :1024].T))
f = timed(lambda: buf.fill_(1.0))
gb = N * N * 8 / 1e9
print(f"lib={os.environ.get('BORNVI_LIB', 'shipped')} n={n} ld={ld} gram ms {[round(x, 3) for x in t]} -> {gb / min(t) * 1e3:.0f} GB/s"
      f" | digest {digest} tables={os.environ.get('BORNVI_GRAM_TABLES', '1')} | fill ms {[round(x, 3) for x in f]} -> {N * ld * 8 / 1e9 / min(f) * 1e3:.0f} GB/s | sym {sym}")
