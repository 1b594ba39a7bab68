"""Is the graph-replayed step host-bound?  n = 8, L = 4: host seconds per graph.replay() call (no synchronisation inside the
loop) against the device time of the same replays, and the same for a graph holding several steps."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
import bench

dev = torch.device("cuda:0")
vi, x = bench.make_vi("n8_L4_dense", dev, overlap=0)
vi._prepare_stein(x)
opt_state = vi.make_optimizer(0.005, 100000, True, "adam", (0.9, 0.999), capturable=True)
step = vi.make_graphed_step(*opt_state, 10.0, warmup=3)
g = step.graph
for _ in range(200):
    step()
torch.cuda.synchronize()
for K in (2000, 2000):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    a.record()
    for _ in range(K):
        g.replay()
    t_host = time.perf_counter() - t0
    b.record()
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"replay only: host loop {t_host / K * 1e6:.1f} us per replay, device span {a.elapsed_time(b) / K * 1e3:.1f} us, wall {t_all / K * 1e6:.1f} us")
    t0 = time.perf_counter()
    for _ in range(K):
        step()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    print(f"step() (replay + host bookkeeping): host loop {t_host / K * 1e6:.1f} us, wall {(time.perf_counter() - t0) / K * 1e6:.1f} us")
