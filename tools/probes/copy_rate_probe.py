"""What a plain device-to-device copy reaches on this box (bytes read + bytes written per second), next to a fill and a
read-only reduction: the ceilings the pass kernel's HBM side is compared with."""
import torch
dev = torch.device("cuda:0")

def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return best

for gib in (1, 4, 8):
    n = gib * (1 << 30) // 8
    src = torch.rand(n, dtype=torch.float64, device=dev)
    dst = torch.empty_like(src)
    t_copy = timed(lambda: dst.copy_(src))
    t_fill = timed(lambda: dst.fill_(1.0))
    t_sum = timed(lambda: src.sum())
    t_add = timed(lambda: torch.add(src, 1.0, out=dst))
    t_inpl = timed(lambda: src.add_(1.0))
    gb = gib * (1 << 30) / 1e9
    print(f"{gib} GiB: copy_ {2 * gb / t_copy * 1e3:.0f} GB/s (r+w), out-of-place add {2 * gb / t_add * 1e3:.0f}, in-place add {2 * gb / t_inpl * 1e3:.0f}, "
          f"fill {gb / t_fill * 1e3:.0f}, sum (read) {gb / t_sum * 1e3:.0f}", flush=True)
    del src, dst
