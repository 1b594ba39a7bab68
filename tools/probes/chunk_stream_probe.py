"""Probe: do the 2P + 1 circuits of a step run faster as small chunks on several streams (each chunk's ping-pong states
stay in the 256 MiB Infinity Cache between passes; the streams' launches overlap each other's ramp-up and tails)?
Compares one call for the whole batch with chunked multi-stream execution, eager and replayed from a HIP graph."""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tensornetworks_amd import backend

n, L, ansatz = int(os.environ.get("N", "16")), int(os.environ.get("L", "6")), "hardware_efficient"
dev = torch.device("cuda", 0)
P = backend.num_params(ansatz, n, L)
g = torch.Generator().manual_seed(0)
theta = (0.1 * torch.randn(P, generator=g, dtype=torch.float32)).double().to(dev)
out = torch.empty((2 * P + 1, 1 << n), dtype=torch.float64, device=dev)


def clock(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


base = clock(lambda: backend.paramshift_probs(ansatz, n, L, theta, 0, P, include_base=True, out=out))
ref = out.clone()
print(f"n={n} L={L}: whole batch, one call: {base:.3f} ms", flush=True)
CONFIGS = [(1, 4, 12), (1, 4, 16), (2, 2, 16), (2, 2, 24)] if os.environ.get("SMALL_CHUNKS") else [(0, 2, 144), (0, 3, 96), (0, 2, 72), (0, 4, 72)]
for wgs_per_cu, nstreams, chunk_params in CONFIGS:
    if n >= 18 and wgs_per_cu > 1:
        continue
    backend.set_option(dev, "fast_workgroups_per_cu", wgs_per_cu)
    streams = [torch.cuda.Stream(device=dev) for _ in range(nstreams)]
    chunks = [(p0, min(P, p0 + chunk_params)) for p0 in range(0, P, chunk_params)]

    def run_chunks():
        main = torch.cuda.current_stream(dev)
        ev = torch.cuda.Event()
        ev.record(main)
        for ci, (p0, p1) in enumerate(chunks):
            s = streams[ci % nstreams]
            with torch.cuda.stream(s):
                if ci < nstreams:
                    s.wait_event(ev)
                rows = out[1 + 2 * p0: 1 + 2 * p1]
                backend.paramshift_probs(ansatz, n, L, theta, p0, p1, include_base=False, out=rows, ws_tag="chunk")
                if ci == 0:
                    backend.paramshift_probs(ansatz, n, L, theta, 0, 0, include_base=True, out=out[:1], ws_tag="base")
        for s in streams:
            e = torch.cuda.Event()
            e.record(s)
            main.wait_event(e)

    out.zero_()
    t_eager = clock(run_chunks)
    ok = torch.equal(out, ref)
    # the same under one graph
    side = torch.cuda.Stream(device=dev)
    t_graph = float("nan")
    try:
        if not os.environ.get("WITH_GRAPH"):
            raise RuntimeError("graph leg skipped (WITH_GRAPH unset)")
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            run_chunks()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=side):
            run_chunks()
        t_graph = clock(gr.replay)
        ok = ok and torch.equal(out, ref)
    except Exception as e:  # noqa
        print("   graph capture failed:", type(e).__name__, str(e)[:200])
    print(f"  wgs/cu {wgs_per_cu} streams {nstreams} chunk {2 * chunk_params} circuits ({len(chunks)} chunks, "
          f"{nstreams * 2 * chunk_params * (32 << n) / 2**20:.0f} MiB in flight): eager {t_eager:.3f} ms, graph {t_graph:.3f} ms, bits equal {ok}", flush=True)
backend.set_option(dev, "fast_workgroups_per_cu", 0)
