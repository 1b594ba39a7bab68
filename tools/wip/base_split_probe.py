"""577 circuits in one batch (19 rounds of the 1024 persistent workgroups) vs 576 (18 rounds) + the base circuit alone"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tensornetworks_amd import backend as be
dev = torch.device("cuda:0")
n, L = 16, 6
P = 3 * n * L
th = torch.rand(P, dtype=torch.float64, device=dev)


def clock(fn, reps=10):
    fn(); fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


full = torch.empty((2 * P + 1, 1 << n), dtype=torch.float64, device=dev)
sh = torch.empty((2 * P, 1 << n), dtype=torch.float64, device=dev)
b1 = torch.empty((1, 1 << n), dtype=torch.float64, device=dev)
t577 = clock(lambda: be.paramshift_probs("hardware_efficient", n, L, th, 0, P, include_base=True, out=full))
t576 = clock(lambda: be.paramshift_probs("hardware_efficient", n, L, th, 0, P, include_base=False, out=sh))
t1 = clock(lambda: be.paramshift_probs("hardware_efficient", n, L, th, 0, 0, include_base=True, out=b1, ws_tag="base"))
print(f"577 together {t577:.3f} ms;  576 {t576:.3f} ms + base alone {t1:.3f} ms = {t576 + t1:.3f} ms")
