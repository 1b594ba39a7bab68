#!/usr/bin/env python3
"""KSD-gradient steps/s on MI355X (BASELINE.json metric), one JSON line on rank 0.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Workload (config.workload): BASELINE config 3 -- quantum Born machine n = 16 qubits, L = 6 layers,
`hardware_efficient` ansatz (P = 288, 577 circuits of 421 gates per step), synthetic 17-node
Bayesian network (seed 0), dense 2^16 x 2^16 fp64 Stein Gram (32 GiB),
Adam + cosine schedule + clip as run_sprinkler_quantum_ksd.py:35-43.

One step = one epoch body of ksd_vi_quantum.py:110-161 with TVD tracking off: base circuit -> q,
KSD = sqrt(clamp(q^T K_p q)), dL/dq, 2P parameter-shift circuits -> grad, clip, Adam, scheduler.
S and K_p are built once before the timed region (they do not depend on theta).

Timing: W warm-up steps, then R repeats of EXACTLY K steps, each repeat bracketed by a barrier +
torch.cuda.synchronize() on both sides and reduced with MAX over the ranks; `value` = K / median repeat
(R >= 5 and R * K steps >= 2 s, so that a 2 % change can be ranked; min / max are reported beside it).

N > 1 (strong scaling: the same step, split): the 2P shifted circuits are dealt to the ranks, the upper
triangle of K_p is sharded in strip pairs; one all-reduce (K q) and one all-gather (gradient scalars) per
step over RCCL.  `series` carries the same measurement for BASELINE config 4 (n = 20, L = 8, matrix-free
contraction), the parameter-shift shard the >= 3.5x-at-8-GPUs target is quoted on.
"""
import argparse
import glob
import json
import math
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
FP64_MFMA_PEAK_TFLOPS = 78.6   # v_mfma_f64_16x16x4_f64, dense

WORKLOADS = {
    # name: (n, layers, ansatz, gram_mode)
    "n16_L6_dense": (16, 6, "hardware_efficient", "dense"),
    "n16_L6_kron": (16, 6, "hardware_efficient", "kron"),
    "n8_L4_dense": (8, 4, "hardware_efficient", "dense"),
    "n20_L8_kron": (20, 8, "hardware_efficient", "kron"),
    "n12_L4_dense": (12, 4, "hardware_efficient", "dense"),
}
ADV_WORKLOADS = {"adv_n12_b65536": (12, 4, 65536)}     # BASELINE config 5: (n, layers, REINFORCE batch)
# reported beside the headline at every N: BASELINE config 4; the headline's circuit with the matrix-free contraction; config 2
SERIES_WORKLOADS = ("n20_L8_kron", "n16_L6_kron", "n8_L4_dense")
SERIES_WORKLOAD = SERIES_WORKLOADS[0]


def mean_ms(pairs):
    return float(np.mean([a.elapsed_time(b) for a, b in pairs])) if pairs else 0.0


class Dist:
    """The torch.distributed facts bench.py needs (rank 0 / world 1 when the job is a single process)."""

    def __init__(self, dev):
        import torch.distributed as dist
        self.dist = dist
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.backend = os.environ.get("BORNVI_DIST_BACKEND", "nccl")   # "gloo": rehearsal with ranks sharing one GPU
        self.dev = dev
        if self.world > 1:
            if self.backend == "nccl":
                dist.init_process_group("nccl", device_id=dev)
            else:
                dist.init_process_group(self.backend)

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()

    def max_over_ranks(self, value):
        if self.world == 1:
            return float(value)
        t = torch.tensor([value], dtype=torch.float64, device=self.dev if self.backend == "nccl" else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def gather_objects(self, obj):
        if self.world == 1:
            return [obj]
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out

    def close(self):
        if self.world > 1:
            self.dist.barrier()
            self.dist.destroy_process_group()


def gate_apply_microbench(dev, n=16, total_bytes=4 << 30, reps=20, warm=5):
    """Un-fused gate kernels on a batch of states >> Infinity Cache: algorithmic == real traffic,
    32 * 2^n bytes per gate per state."""
    from tensornetworks_amd import backend
    B = total_bytes // (16 << n)
    st = torch.randn((B, 2 << n), dtype=torch.float64, device=dev).view(torch.complex128).view(B, 1 << n)
    st /= st.abs().pow(2).sum(1, keepdim=True).sqrt()
    c, sn = np.cos(0.15), np.sin(0.15)                      # RY(0.3) = exp(-i 0.3 Y / 2)
    U = np.array([[c, -sn], [sn, c]], dtype=np.complex128)
    out = {"n": n, "batch": int(B), "bytes_per_launch": int(32 * B << n), "gbs": {}}
    cases = [("ry_wire0", lambda: backend.gate1q_apply(st, n, 0, U)),
             (f"ry_wire{n // 2}", lambda: backend.gate1q_apply(st, n, n // 2, U)),
             (f"ry_wire{n - 1}", lambda: backend.gate1q_apply(st, n, n - 1, U)),
             ("cnot_0_1", lambda: backend.cnot_apply(st, n, 0, 1))]
    for name, fn in cases:
        for _ in range(warm):
            fn()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for a, b in ev:
            a.record(); fn(); b.record()
        torch.cuda.synchronize(dev)
        ms = float(np.median([a.elapsed_time(b) for a, b in ev]))
        # (a CNOT moves only the half of the state whose control bit is set: 16 * 2^n bytes per state, not 32 * 2^n)
        nbytes = out["bytes_per_launch"] // 2 if name.startswith("cnot") else out["bytes_per_launch"]
        out["gbs"][name] = round(nbytes / (ms * 1e-3) / 1e9, 1)
    vals = [v for k, v in out["gbs"].items() if k.startswith("ry")]
    out["ry_mean_gbs"] = round(float(np.mean(vals)), 1)
    out["frac_of_hbm_peak"] = round(out["ry_mean_gbs"] / HBM_PEAK_GBS, 4)
    out["byte_model"] = "one-qubit gate: 32 * 2^n bytes per state (read + write every amplitude); CNOT: 16 * 2^n (the control = 1 half)"
    del st
    return out


FP64_MFMA_PEAK_TFLOPS = 78.6      # MI355X_MICROARCH.md: dense fp64 matrix peak (= the fp64 vector peak on gfx950)


def mfma_extras(vi, dev, n):
    """The two matrix-core kernels of the path, measured live on the trainer's own (padded) K_p: the batched contraction
    Y = K_p Q^T (kernels_batched.hip: v_mfma_f64_16x16x4) for B = 128 and B = 2P + 1 distributions, and the Gram build
    (kernels_stein.hip: gram_tables_kernel, one MFMA product + two table terms per 16 x 16 tile; HBM-write bound)."""
    from tensornetworks_amd import backend
    N = 1 << n
    out = {}
    K = vi._K
    if K is None or n < 8:
        return None

    def timed(fn, reps=3):
        fn()
        torch.cuda.synchronize(dev)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for a, b in ev:
            a.record(); fn(); b.record()
        torch.cuda.synchronize(dev)
        return float(np.median([a.elapsed_time(b) for a, b in ev]))

    bq = {}
    for B in (128, 2 * vi.born_machine.num_ansatz_params + 1):
        Q = torch.rand((B, N), dtype=torch.float64, device=dev)
        Q /= Q.sum(dim=1, keepdim=True)
        ms = timed(lambda: backend.stein_quadform(K, Q, n, want_y=False))
        flop = 2.0 * N * N * B
        floor_ms = max(8.0 * N * N / (HBM_PEAK_GBS * 1e9), flop / (FP64_MFMA_PEAK_TFLOPS * 1e12)) * 1e3
        bq[f"B{B}"] = {"ms": round(ms, 3), "tflops": round(flop / (ms * 1e-3) / 1e12, 2),
                       "frac_of_fp64_mfma_peak": round(flop / (ms * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS, 4),
                       "roofline_floor_ms": round(floor_ms, 3), "over_floor": round(ms / floor_ms, 3)}
        del Q
    out["batched_quadform"] = {"kernel": "quadform_batched_kernel (v_mfma_f64_16x16x4_f64, 64 x 64 wave tiles) + rowdot",
                               "K_row_pitch": int(K.stride(0)), "peak_tflops": FP64_MFMA_PEAK_TFLOPS, **bq,
                               "note": "the KSD of B distributions at once (diagnostic of SURVEY 0.5); the training step needs B = 1"}
    ld = int(K.stride(0))
    scratch = torch.empty((N, ld), dtype=torch.float64, device=dev)[:, :N] if ld != N else torch.empty((N, N), dtype=torch.float64, device=dev)
    ms = timed(lambda: backend.stein_gram(vi._S, n, vi.base_kernel_length_scale, out=scratch, ld=ld if ld != N else None))
    flop = 2.0 * n * N * N                           # the score product S S^T on the matrix cores (the two bit products: tables)
    out["gram_build"] = {"kernel": f"gram_tables_kernel<{n}>", "ms": round(ms, 3),
                         "written_gbs": round(8.0 * N * N / (ms * 1e-3) / 1e9, 1),
                         "frac_of_hbm_peak": round(8.0 * N * N / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         "mfma_tflops": round(flop / (ms * 1e-3) / 1e12, 2),
                         "note": "one-off per observation (outside the step); 8 * 4^n bytes written, 2 n 4^n flop on the matrix cores "
                                 "(fp64 MFMA holds the vector ALU on gfx950: the kernel is priced in vector-pipe cycles, DESIGN 4.3)"}
    del scratch
    return out


def b2_pennylane(ansatz, n, layers, theta64):
    """BASELINE.md B2 (opportunistic): PennyLane default.qubit + parameter-shift driven by this build's own circuit
    definition, if PennyLane happens to be importable on the box.  It is not installed in this image and nothing here
    tries to obtain it."""
    try:
        import pennylane as qml
    except Exception:
        print("[bench] PennyLane unavailable -- B1 reported", file=sys.stderr)
        return "PennyLane unavailable -- B1 reported"
    try:
        from oracle import circuit as oc
        gl = oc.gate_list(ansatz, n, layers)
        dev = qml.device("default.qubit", wires=n, shots=None)

        @qml.qnode(dev, interface="torch", diff_method="parameter-shift")
        def pqc(w):
            for kind, wires, p in gl:
                if kind == "H":
                    qml.Hadamard(wires=wires[0])
                elif kind in ("RX", "RY", "RZ"):
                    getattr(qml, kind)(w[p], wires=wires[0])
                elif kind == "CNOT":
                    qml.CNOT(wires=list(wires))
                else:
                    qml.CZ(wires=list(wires))
            return qml.probs(wires=range(n))
        w = torch.tensor(theta64, dtype=torch.float64, requires_grad=True)
        t0 = time.perf_counter()
        q = pqc(w)
        t_fwd = time.perf_counter() - t0
        return {"forward_seconds": round(t_fwd, 4), "sum_q": float(q.sum()),
                "note": "one default.qubit forward; 2P parameter-shift evaluations cost about 2P times this"}
    except Exception as e:       # a PennyLane of an unknown version: report, never fail the bench
        return f"PennyLane importable but the B2 leg failed: {type(e).__name__}: {e}"


def cpu_baseline(n, layers, ansatz, gram_mode, S_host, theta64, max_params=None):
    """ONE full KSD-gradient step of the oracle's C port (oracle/cpu_port.c, OpenMP) on the host cores: all 2P + 1
    circuits, then the contraction in the same form the GPU leg uses -- dense: y = K_p q over the whole 2^n x 2^n matrix,
    streamed in freshly built 8 GiB row blocks (building K_p is outside the step, as on the GPU); kron: the matrix-free
    Kronecker mat-vec.  No extrapolation.  A reported baseline, not a target."""
    from oracle import cpu_port as cp
    if not cp.available():
        return None
    N = 1 << n
    P = theta64.size
    # un-timed warm-up of the port's OpenMP thread pool (its first parallel region in a process that has torch's own
    # OpenMP runtime loaded takes ~1 s)
    cp.circuit_probs("basic", 4, 1, np.zeros((2 * cp.max_threads(), 8)))
    cp.kron_matvec(np.zeros((64, 6)), np.full(64, 1.0 / 64), 6, 1.0)
    # max_params (the n = 20 series leg): a BOUNDED sample -- the base circuit and the shifted circuits of the first
    # max_params parameters are run and timed, the circuit and gradient-dot times are scaled to all 2P + 1 circuits (every
    # circuit costs the same: same gates, same state size); the contraction is timed whole
    Ps = P if (max_params is None or max_params >= P) else int(max_params)
    t0 = time.perf_counter()
    probs, used = cp.paramshift_probs(ansatz, n, layers, theta64, 0, Ps, include_base=True)
    t_circ = (time.perf_counter() - t0) * (2 * P + 1) / (2 * Ps + 1)
    q = np.ascontiguousarray(probs[0])
    shifted = probs[1:]
    if gram_mode == "dense":
        rows_blk = max(1, min(N, (8 << 30) // (8 * N)))
        y = np.empty(N)
        t_gemv, t_build = 0.0, 0.0
        for r0 in range(0, N, rows_blk):
            r1 = min(N, r0 + rows_blk)
            tb = time.perf_counter()
            K_rows = cp.gram_rows(S_host, n, 1.0, r0, r1)        # outside the step (theta-independent)
            t_build += time.perf_counter() - tb
            tg = time.perf_counter()
            y[r0:r1], _ = cp.gemv_rows(K_rows, n, r0, r1, q)
            t_gemv += time.perf_counter() - tg
            del K_rows
        form = f"dense GEMV over all {N} rows in {rows_blk}-row blocks ({t_gemv * 1e3:.1f} ms; K_p build {t_build:.1f} s not counted)"
    else:
        t_gemv = float("inf")
        for _ in range(3):
            tg = time.perf_counter()
            y, _ = cp.kron_matvec(S_host, q, n, 1.0)
            t_gemv = min(t_gemv, time.perf_counter() - tg)
        form = f"Kronecker mat-vec (best of 3: {t_gemv * 1e3:.1f} ms)"
    tf = time.perf_counter()
    ksd2 = float(q @ y)
    loss = math.sqrt(max(ksd2, 1e-12))
    grad = 0.5 * ((shifted[0::2] - shifted[1::2]) @ (y / loss))
    t_fin = (time.perf_counter() - tf) * P / Ps
    step_s = t_circ + t_gemv + t_fin
    return {"value": round(1.0 / step_s, 6), "unit": "steps/s", "cores": int(used), "kind": "port",
            "host_cpus": os.cpu_count(), "torch_threads": torch.get_num_threads(),
            "sample": (f"one full step: {probs.shape[0]} circuits ({t_circ:.2f} s) + {form} + gradient dots "
                       f"({t_fin * 1e3:.0f} ms), oracle/cpu_port.c with OpenMP, nothing extrapolated") if Ps == P else
                      (f"bounded sample: {probs.shape[0]} of the step's {2 * P + 1} circuits run and timed (scaled to {t_circ:.1f} s for all: "
                       f"every circuit is the same work) + {form} + gradient dots (scaled, {t_fin * 1e3:.0f} ms), oracle/cpu_port.c with OpenMP"),
            "step_seconds": round(step_s, 3), "loss": loss, "grad_norm": float(np.linalg.norm(grad)) if Ps == P else None,
            "b2_pennylane": b2_pennylane(ansatz, n, layers, theta64)}


def make_vi(workload, dev, process_group=None, overlap=0):
    from tensornetworks_amd.bayesian_network import synthetic_network
    from tensornetworks_amd.ksd_vi_quantum import KSDVariationalInference
    n, layers, ansatz, gram_mode = WORKLOADS[workload]
    bn, lat, obs, x = synthetic_network(n, seed=0)
    torch.manual_seed(0)
    vi = KSDVariationalInference(bn, lat, obs, qbm_num_latent_vars=n, qbm_ansatz_layers=layers,
                                 qbm_ansatz_type=ansatz, pytorch_device=str(dev), gram_mode=gram_mode,
                                 process_group=process_group)
    if overlap >= 0:
        vi.overlap_streams = {0: False, 1: True, 2: "partition"}[overlap]
    if int(os.environ.get("BORNVI_PLACEMENT_TRIES", "-1")) >= 0:      # A/B of the K_p placement selection (default: the trainer's)
        vi.gram_placement_tries = int(os.environ["BORNVI_PLACEMENT_TRIES"])
    g = torch.Generator().manual_seed(0)
    P = vi.born_machine.num_ansatz_params
    with torch.no_grad():     # theta0 = 0.1 * randn(P) float32, `small_random` (quantum_born_machine.py:43-45)
        vi.born_machine.theta.copy_((0.1 * torch.randn(P, generator=g, dtype=torch.float32)).to(dev))
    return vi, x


def run_repeat(vi, step_fn, opt_state, clip, K, D, dev):
    """EXACTLY K steps between barrier + synchronize on both sides; MAX over the ranks of the elapsed seconds."""
    params, opt, sched = opt_state
    losses = []
    D.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(K):
        losses.append(step_fn(params, opt, sched, clip)[0])
    torch.cuda.synchronize(dev)
    D.barrier()
    return D.max_over_ranks(time.perf_counter() - t0), losses


def phase_means(timers):
    return {k: round(mean_ms(v), 4) for k, v in (timers or {}).items()}


def dist_selftest(dev, D):
    """N > 1: the sharded step against an UN-sharded recomputation of the same step on every rank (n = 12, L = 4, dense
    Gram): all-reduced K q and q^T K q, gathered gradient, loss; and every rank holds the same bits."""
    from tensornetworks_amd import paramshift_shard as shard
    vs, x = make_vi("n12_L4_dense", dev)                       # default process group: sharded
    v1, _ = make_vi("n12_L4_dense", dev, process_group=shard.SOLO)
    with torch.no_grad():
        v1.born_machine.theta.copy_(vs.born_machine.theta)
    vs._prepare_stein(x)
    v1._prepare_stein(x)
    ls, gs, qs = vs.ksd_and_grad()
    l1, g1, q1 = v1.ksd_and_grad()
    k2s, ys = vs._stein_contract(qs)
    k21, y1 = v1._stein_contract(q1)
    torch.cuda.synchronize(dev)
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max().clamp_min(1e-300))
    errs = {"max_rel_err_y": rel(ys, y1), "max_rel_err_ksd2": rel(k2s, k21), "max_rel_err_grad": rel(gs, g1),
            "max_rel_err_loss": rel(ls, l1), "q_bits_equal": bool(torch.equal(qs, q1))}
    # the same bits on every rank (identical optimiser updates need no broadcast): compare with rank 0's copy
    mine = torch.cat([gs, ys, ls])
    ref = mine.clone() if D.backend == "nccl" else mine.cpu().clone()
    if D.world > 1:
        D.dist.broadcast(ref, src=0)
    same = bool(torch.equal(ref.cpu(), mine.cpu()))
    allr = D.gather_objects({"rank": D.rank, **errs, "same_bits_as_rank0": same})
    ok = all(r["same_bits_as_rank0"] and r["q_bits_equal"] and r["max_rel_err_y"] < 1e-12 and r["max_rel_err_grad"] < 1e-10
             and r["max_rel_err_loss"] < 1e-12 for r in allr)
    out = {"ok": ok, "ranks": D.world, "workload": "n12_L4_dense", "strip_pair_shard": vs._K_pairs is not None,
           "max_rel_err_y": max(r["max_rel_err_y"] for r in allr), "max_rel_err_grad": max(r["max_rel_err_grad"] for r in allr),
           "max_rel_err_loss": max(r["max_rel_err_loss"] for r in allr), "per_rank": allr}
    del vs, v1
    torch.cuda.empty_cache()
    if not ok:
        raise SystemExit("bench.py --dist-selftest FAILED: " + json.dumps(out))
    return out


def measure(workload, dev, D, args, steps, warmup, repeats, want_extras):
    """Builds the trainer for `workload`, runs warm-up + repeats; returns everything rank 0 needs for the record."""
    from tensornetworks_amd import backend
    n, layers, ansatz, gram_mode = WORKLOADS[workload]
    vi, x = make_vi(workload, dev, overlap=args.overlap)
    P = vi.born_machine.num_ansatz_params
    theta0 = vi.born_machine.theta.detach().double().cpu().numpy()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    vi._prepare_stein(x)
    torch.cuda.synchronize(dev)
    precompute_s = time.perf_counter() - t0
    if args.overlap < 0:
        vi.choose_overlap()
    use_graph = (args.graph == 1 or (args.graph < 0 and n <= 13)) and not args.host_sync and D.world == 1 and args.overlap == 0
    opt_state = vi.make_optimizer(0.005, warmup + steps * 64, True, "adam", (0.9, 0.999), capturable=use_graph)
    clip = 10.0
    # --host-sync 1: every step ends with loss.item() like the reference's epoch (the GPU idles while the host
    # handles it); 0 (default): the same steps with the read-back of the K losses deferred to the end of the repeat
    # (NaN/Inf guard on the device), so the K steps run back to back
    step_fn = vi.training_step if args.host_sync else vi.training_step_async
    losses = []
    if use_graph:
        # latency-bound sizes: the whole step (circuits, contraction, gradient, clip, Adam) replayed from ONE HIP graph
        graphed = vi.make_graphed_step(*opt_state, clip, warmup=max(3, warmup), device_adam=args.device_adam != 0)
        if graphed.adam is not None:      # the optimiser kernel records every epoch's loss on the device: nothing to clone
            step_fn = lambda params, opt, sched, clip_: graphed()
        else:
            step_fn = lambda params, opt, sched, clip_: tuple(t.clone() if i == 0 else t for i, t in enumerate(graphed()))
    else:
        for _ in range(warmup):
            losses.append(step_fn(*opt_state, clip)[0])
    vi.timers = None if use_graph else {}
    elapsed = []
    first, ls = run_repeat(vi, step_fn, opt_state, clip, steps, D, dev)
    elapsed.append(first)
    losses += ls
    R = repeats if repeats > 0 else int(min(50, max(5, math.ceil(2.0 / max(first, 1e-6)))))
    for _ in range(R - 1):
        e, ls = run_repeat(vi, step_fn, opt_state, clip, steps, D, dev)
        elapsed.append(e)
        losses += ls
    timers = vi.timers
    if use_graph:          # phase spans cannot be recorded inside a graph: take them from a few eager steps afterwards
        vi.timers = {}
        for _ in range(5):
            vi.training_step_async(*opt_state, clip)
        torch.cuda.synchronize(dev)
        timers = vi.timers
    vi.timers = None
    # labelled extra: the same K steps with the other setting of prefix sharing (bit-identical shifted distributions,
    # fewer circuit-passes); not part of `value`
    extra_share = None
    if want_extras and not use_graph:
        backend.set_option(dev, "prefix_share", 0 if args.prefix_share else 1)
        for _ in range(2):
            step_fn(*opt_state, clip)
        vi.timers = {}
        e2, _ = run_repeat(vi, step_fn, opt_state, clip, steps, D, dev)
        extra_share = {"prefix_share": 0 if args.prefix_share else 1, "steps_per_sec": round(steps / e2, 4),
                       "ms_per_step": round(1e3 * e2 / steps, 4), "circuits_ms": round(mean_ms(vi.timers.get("circuits")), 4),
                       "note": "same step with prefix sharing of the parameter-shift batch switched "
                               + ("off" if args.prefix_share else "on") + " (opt-in, bornvi_set_option prefix_share): a shifted "
                               "circuit starts from the base circuit's state at the first pass its parameter touches; rows bit-identical"}
        vi.timers = None
        backend.set_option(dev, "prefix_share", 1 if args.prefix_share else 0)
    extra_adjoint = None
    if want_extras and D.world == 1:
        # labelled extra: the same step with the OPT-IN adjoint gradient engine (one forward + one backward walk over
        # the gates instead of 2P shifted circuits; same gradient to rounding); not part of `value`
        vi.grad_engine = "adjoint"
        astep = vi.training_step_async
        for _ in range(2):
            astep(*opt_state, clip)
        e3, _ = run_repeat(vi, astep, opt_state, clip, steps, D, dev)
        vi.grad_engine = "paramshift"
        extra_adjoint = {"grad_engine": "adjoint", "steps_per_sec": round(steps / e3, 4), "ms_per_step": round(1e3 * e3 / steps, 4),
                         "note": "opt-in (SURVEY 8(f) row 4): adjoint differentiation replaces the 2P parameter-shift circuit "
                                 "evaluations; eager launches, gate-block kernels (kernels_adjoint.hip)"}
    if use_graph and graphed.adam is not None:
        losses = graphed.adam.history()[0].cpu().tolist()      # all epochs, the graph's eager warm-up steps included
    return {"vi": vi, "P": P, "theta0": theta0, "elapsed": elapsed, "timers": timers, "losses": losses, "graph": use_graph,
            "extra_adjoint": extra_adjoint,
            "precompute_s": precompute_s, "extra_share": extra_share, "n": n, "layers": layers, "ansatz": ansatz,
            "gram_mode": gram_mode}


def pmc_traffic(workload, passes, tile_bits, variant="r4"):
    """Measured HBM bytes per launch from a rocprofv3 --pmc run of THIS plan (tools/pmc_traffic.sh + pmc_summarize.py),
    newest matching file under profiles/; a file whose recorded plan signature differs is ignored."""
    best = None
    for f in sorted(glob.glob(os.path.join(REPO, "profiles", f"r*_pmc_traffic_{workload}.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        sig = d.get("signature")
        if sig and (sig.get("passes"), sig.get("tile_bits"), sig.get("kernel", "r4")) == (passes, tile_bits, variant):
            best = (f, d)
    if best is None:
        return {}, None
    f, d = best
    src = f"{os.path.relpath(f, REPO)} (builder's rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE run, commit {d['signature'].get('commit', '?')}; " \
          "not measured in this run)"
    return d["kernels"], src


def kernel_table(m, D, args):
    """Per-kernel roofline rows from the live event spans of the timed region.  ALGORITHMIC bytes per launch = what the
    kernel's own algorithm has to move through HBM:
      circuit pass: every state of the batch in (16 * 2^n; none in the first pass) and out (16 * 2^n, or 8 * 2^n
                    probabilities in the last pass); the SURVEY 8(d) UN-FUSED accounting (32 * 2^n per gate per state)
                    is reported beside it as `survey_8d_unfused_equivalent_gbs`
      contraction:  dense: the upper triangle of K_p, 4 * 2^n * (2^n + 32) bytes per GPU share (SURVEY 8(d) counts the
                    full matrix, 8 * 4^n: `survey_8d_full_matrix_gbs`), or the rank's rows for the row shard;
                    kron: pack + the passes the Kronecker plan RUNS (each moves its (n+2)//2 packed states in and
                    out) + combine."""
    from tensornetworks_amd import _ext
    vi, P, n, layers, ansatz, gram_mode = m["vi"], m["P"], m["n"], m["layers"], m["ansatz"], m["gram_mode"]
    world = D.world
    timers = m["timers"]
    from tensornetworks_amd import backend
    dev = vi._S.device
    # the plan the handle runs: 8 amplitudes per thread (reg_wires = 3, the default, with the read map) where eligible
    reg_wires, read_map = backend.get_option(dev, "reg_wires"), backend.get_option(dev, "read_map")
    plan_flags = args.tile_bits
    compact = None
    if reg_wires == 3:
        rm_flag = 0x100 if read_map != 0 else 0           # (-1 = by the kernel: on with reg_wires = 3)
        compact = _ext.plan_compact_words(_ext.ANSATZ_IDS[ansatz], n, layers, args.tile_bits | rm_flag)
        if compact[0] is not None:
            plan_flags = args.tile_bits | rm_flag | _ext.R3
        else:
            compact = None
    if compact is None and read_map == 1:
        plan_flags |= 0x100
    plan = _ext.plan_words(_ext.ANSATZ_IDS[ansatz], n, layers, plan_flags)
    n_passes, n_gates = int(plan[3]), int(plan[11])
    circuits_rank = 1 + 2 * len(range(0, P, world))            # rank 0: the parameters 0, W, 2W, ...
    circ_ms, stein_ms, fin_ms = mean_ms(timers.get("circuits")), mean_ms(timers.get("stein")), mean_ms(timers.get("finish"))
    base_ms = mean_ms(timers.get("base_circuit"))     # > 0 only in overlap mode: base circuit launched separately
    # the parameter-shift dot fused into the shifted circuits' last pass (bornvi_paramshift_dot_*): that pass runs inside
    # the trainer's "finish" span (with the tiny partial-sum kernel and the all-gather, whose span is subtracted)
    fused_dot = bool(getattr(vi, "fused_dot", False)) and not vi.overlap_streams and vi.grad_engine == "paramshift" and \
        backend.paramshift_dot_supported(ansatz, n, layers, dev, len(range(0, P, world)))
    circ_launches = n_passes * (2 if base_ms > 0 else 1) + (1 if fused_dot else 0)
    circ_kernel_ms = circ_ms + base_ms + (max(fin_ms - mean_ms(timers.get("allgather")), 0.0) if fused_dot else 0.0)
    N = 1 << n
    unfused_bytes = 32.0 * N * n_gates * circuits_rank
    first_pass = _ext.plan_param_first_pass(plan)[list(range(0, P, world))] if n_passes > 1 else None
    share_on = n_passes > 1 and bool(args.prefix_share)
    active = [1 + 2 * int((first_pass <= i).sum()) if share_on else circuits_rank for i in range(n_passes)] if n_passes > 1 else [circuits_rank]
    # support of |0..0> (fast tables, FH_ZINFO = word 7 of a pass header): pass 0 writes only the tiles somebody reads,
    # pass 1 loads only the slots not known to be zero -- the algorithmic bytes are what is left
    w0 = r1 = 1.0
    zero_note = ""
    if n_passes > 2 and not any(o.startswith(("zero_support=0", "direct_stages=")) for o in args.opt) and not args.debug_flags:
        slots = 8 if compact is not None else 16
        fw = compact if compact is not None else _ext.plan_fast_words(_ext.ANSATZ_IDS[ansatz], n, layers, plan_flags)
        zword = 6 if compact is not None else 7                # CH_ZINFO / FH_ZINFO of a pass header
        if fw[0] is not None:
            gmask, zslots = int(fw[0][fw[1][0] + zword]), int(fw[0][fw[1][1] + zword]) & 0xFFFF
            w0, r1 = 0.5 ** bin(gmask).count("1"), (slots - bin(zslots).count("1")) / float(slots)
            if gmask:
                zero_note = (f"; pass 0 writes {w0:.4g} of a state (tiles outside the support of |0..0> that nobody reads are "
                             f"left out), pass 1 reads {r1:.4g} of one (slots known to be zero are not loaded)")
    # (last pass: 8 * 2^n probabilities per circuit -- with the fused dot only the base circuit writes them)
    circ_bytes = sum(a * ((16.0 * N * (r1 if i == 1 else 1.0) if i > 0 else 0.0) +
                          (16.0 * N * (w0 if i == 0 else 1.0) if i < n_passes - 1 else (8.0 * N / a if fused_dot else 8.0 * N)))
                     for i, a in enumerate(active))
    rows_rank = -(-N // world)
    sym = gram_mode == "dense" and vi.symmetric_contraction and (world == 1 or vi._K_pairs is not None)
    stein_name = ("quadform_sym_kernel" if sym else "quadform_kernel") if gram_mode == "dense" else "kron_matvec"
    full_bytes = 8.0 * N * rows_rank
    stein_launches = 1
    stein_note = None
    if gram_mode != "dense":
        kplan = _ext.plan_words(-1, n, 0, args.tile_bits)
        kpasses = int(kplan[3])
        npk = (n + 2) // 2                                   # real vectors packed two per complex state
        pack_b = 8.0 * N * n + 8.0 * N + 16.0 * N * npk
        pass_b = kpasses * 32.0 * N * npk
        comb_b = 16.0 * N * npk + 8.0 * N * n + 16.0 * N
        stein_bytes = pack_b + pass_b + comb_b
        stein_launches = kpasses + 2
        stein_note = (f"matrix-free K_p q: pack ({pack_b / 1e6:.0f} MB) + {kpasses} fused passes over {npk} packed states "
                      f"({pass_b / 1e6:.0f} MB: 32 * 2^n * {npk} per pass) + combine ({comb_b / 1e6:.0f} MB); ms spans all of them")
    elif sym:
        stein_bytes = 4.0 * N * (N + 32) / world
        stein_note = ("K_p is bitwise symmetric: only its upper triangle is read (algorithmic bytes = 4 * 2^n * (2^n + 32)); "
                      "ms includes the column-partial reduce and the final sum")
    else:
        stein_bytes = full_bytes
    pmc, pmc_src = ({}, None)
    if world == 1 and not args.prefix_share:
        pmc, pmc_src = pmc_traffic(m.get("workload", args.workload if m.get("is_main") else ""), n_passes, int(plan[2]),
                                   "r3" if compact is not None else "r4")
    t_circ = next((kv.get("hbm_bytes_per_launch") for kname, kv in pmc.items() if kname.startswith("circuit_pass")), None)
    t_stein = None
    if sym:
        parts = [kv["hbm_bytes_per_launch"] for kname, kv in pmc.items() if kname.startswith("quadform_sym")]
        t_stein = sum(parts) if parts else None
    kern = {
        "circuit_pass_kernel": {"bound": "hbm", "launches_per_step": circ_launches,
                                "kernel": ("circuit_pass_r3_kernel (8 amplitudes per thread, four waves per SIMD" +
                                           (", read map" if plan_flags & 0x100 else "") + ")") if compact is not None
                                          else "circuit_pass_fast_kernel (16 amplitudes per thread, two waves per SIMD)",
                                "fused_paramshift_dot": fused_dot,
                                "achieved": round(circ_bytes / (circ_kernel_ms * 1e-3) / 1e9, 1) if circ_ms else None,
                                "peak": HBM_PEAK_GBS, "unit": "GB/s", "ms_per_step": round(circ_kernel_ms, 4),
                                "avg_launch_ms": round(circ_kernel_ms / circ_launches, 4),
                                "algorithmic_bytes_per_launch": circ_bytes / circ_launches, "traffic": t_circ,
                                "traffic_source": pmc_src if t_circ else None,
                                "circuit_passes_run": sum(active), "circuit_passes_without_prefix_sharing": circuits_rank * n_passes,
                                "survey_8d_unfused_equivalent_gbs": round(unfused_bytes / (circ_kernel_ms * 1e-3) / 1e9, 1) if circ_ms else None,
                                "note": "fused LDS-tiled engine: algorithmic bytes = each state read and written once per pass" + zero_note +
                                        "; the same work as un-fused gate-apply (SURVEY 8(d): 32 * 2^n bytes per gate per state) "
                                        "would need the survey_8d_unfused_equivalent_gbs rate"},
        stein_name: {
            "bound": "hbm", "launches_per_step": stein_launches,
            "achieved": round(stein_bytes / (stein_ms * 1e-3) / 1e9, 1) if stein_ms else None,
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "ms_per_step": round(stein_ms, 4),
            "avg_launch_ms": round(stein_ms, 4), "algorithmic_bytes_per_launch": stein_bytes, "traffic": t_stein,
            "traffic_source": pmc_src if t_stein else None,
            "survey_8d_full_matrix_gbs": round(full_bytes / (stein_ms * 1e-3) / 1e9, 1) if (sym and stein_ms) else None,
            "note": stein_note},
    }
    for v in kern.values():
        v["frac"] = round(v["achieved"] / v["peak"], 4) if v["achieved"] else None
        if v["traffic"] and v["ms_per_step"]:
            per_step = v["traffic"] * (v["launches_per_step"] if v is kern["circuit_pass_kernel"] else 1)
            v["traffic_gbs"] = round(per_step / (v["ms_per_step"] * 1e-3) / 1e9, 1)
            v["traffic_frac"] = round(v["traffic_gbs"] / v["peak"], 4)
    config = {"n_qubits": n, "layers": layers, "ansatz": ansatz, "params": P, "circuits_per_step": 1 + 2 * P,
              "gates_per_circuit": n_gates, "gram": gram_mode, "tile_bits": int(plan[2]), "passes": n_passes}
    phases = {"circuits": round(circ_ms, 4), "base_circuit": round(base_ms, 4), "stein": round(stein_ms, 4),
              "finish": round(fin_ms, 4), "allreduce": round(mean_ms(timers.get("allreduce")), 4),
              "allgather": round(mean_ms(timers.get("allgather")), 4)}
    return kern, config, phases


def summarize(m, steps):
    per = sorted(1e3 * e / steps for e in m["elapsed"])
    med = float(np.median(per))
    return med, {"n": len(per), "steps_each": steps, "ms_per_step_median": round(med, 4), "ms_per_step_min": round(per[0], 4),
                 "ms_per_step_max": round(per[-1], 4), "timed_seconds_total": round(float(sum(m["elapsed"])), 3)}


def bench_adversarial(args, dev, D):
    """BASELINE config 5: adversarial-VI epochs/s at n = 12 qubits, REINFORCE batch 65,536, classifier
    forward/backward, one classifier step + one Born-machine step per epoch (the reference's defaults,
    adversarial_vi.py:104-107).  A step = one epoch; phases from event spans inside the trainer."""
    import contextlib
    import io
    from tensornetworks_amd.adversarial_vi import AdversarialVariationalInference
    from tensornetworks_amd.bayesian_network import synthetic_network
    n, layers, batch = ADV_WORKLOADS[args.workload]
    # milder CPTs than the KSD benchmarks: with U(0.01, 0.99) tables some states have p(z) < 1e-9, for which the
    # reference's rule (adversarial_vi.py:91-96) makes the reward infinite
    bn, lat, obs, x = synthetic_network(n, 0, p_low=0.25, p_high=0.75)
    torch.manual_seed(0)
    adv = AdversarialVariationalInference(bn, lat, obs, born_machine_config={'ansatz_layers': layers, 'conditioning_dim': 0},
                                          classifier_config={}, device=str(dev))
    kw = dict(batch_size=batch, lr_born_machine=0.003, lr_classifier=0.03, k_classifier_steps=1, k_born_steps=1, verbose=False,
              adam_betas=(0.5, 0.999))
    graph = None if args.graph < 0 else bool(args.graph)
    with contextlib.redirect_stdout(io.StringIO()):
        adv.train(x, num_epochs=max(4, args.warmup), graph_epochs=graph, **kw)
        elapsed = []
        hist = None
        graphed = 0
        for _ in range(max(3, args.repeats)):
            D.barrier()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            hist = adv.train(x, num_epochs=args.steps, graph_epochs=graph, **kw)
            torch.cuda.synchronize(dev)
            D.barrier()
            elapsed.append(D.max_over_ranks(time.perf_counter() - t0))
            graphed = adv.graphed_epochs
        adv.timers = {}                       # phase spans: a few eager epochs afterwards (no events inside a graph)
        adv.train(x, num_epochs=max(4, args.steps // 2), graph_epochs=False, **kw)
    per = sorted(1e3 * e / args.steps for e in elapsed)
    med = float(np.median(per))
    P = adv.born_machine.num_ansatz_params
    rec = {"metric": "adversarial_vi_epochs_per_sec", "value": round(1e3 / med, 4), "unit": "epochs/s", "n_gpus": D.world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(med, 4), "higher_is_better": True, "scaling": "replicas only",
           "vs_baseline": None, "dtype": "f64 circuits / f32 classifier", "data": "synthetic",
           "config": {"workload": args.workload, "n_qubits": n, "layers": layers, "params": P, "reinforce_batch": batch,
                      "k_classifier_steps": 1, "k_born_steps": 1, "circuits_per_epoch": 2 + 2 * P,
                      "classifier": "BinaryClassifierMLP defaults (classifier_pytorch.py:27-41)",
                      "bayesian_network": f"synthetic n={n} seed=0, CPT entries U(0.25, 0.75)"},
           "repeats": {"n": len(per), "ms_per_step_median": round(med, 4), "ms_per_step_min": round(per[0], 4), "ms_per_step_max": round(per[-1], 4)},
           "launch": (f"{graphed} of {args.steps} epochs per call replayed from one HIP graph (the first two run eagerly and the "
                      "third captures)" if graphed else "eager kernel launches") + (f"; capture failed: {adv.graph_error}" if adv.graph_error else ""),
           "phase_ms": {**phase_means(adv.timers), "note": "event spans of eager epochs run after the timed region"},
           "roofline": None, "cpu_baseline": None,
           "note": "torch-op composition around the HIP circuit engine (sampling, table gathers, MLP): no kernel of its own, "
                   "so no roofline row; the Born step's backward is 2P parameter-shift circuits",
           "loss_last": {"classifier": hist['loss_classifier'][-1], "born_machine": hist['loss_born_machine'][-1]}}
    if D.rank == 0:
        print(json.dumps(rec))
    D.close()


def launch_ranks(n_ranks, argv):
    """`python bench.py --gpus N` started as ONE process (no WORLD_SIZE / RANK in the environment): run the N ranks as
    children of `python -m torch.distributed.run` -- the launch line the driver itself uses -- on a free local port, relay
    their output and return their exit code.  This process must not have initialised the GPU (it has not: nothing before
    the call touches HIP), and it never replaces itself (no exec): it waits for the child."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL needs it on this driver
    if torch.cuda.device_count() < n_ranks and "BORNVI_DIST_BACKEND" not in env:
        # fewer GPUs than ranks (a rehearsal on a one-GPU box): RCCL cannot put two ranks on one device
        env["BORNVI_DIST_BACKEND"] = "gloo"
        print(f"[bench] {n_ranks} ranks on {torch.cuda.device_count()} GPU(s): rehearsal with the gloo backend", file=sys.stderr)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    print("[bench] starting ranks:", " ".join(cmd), file=sys.stderr)
    rc = subprocess.run(cmd, env=env).returncode
    if rc != 0:
        raise SystemExit(rc)
    return rc


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--repeats", type=int, default=0, help="repeats of the K timed steps (0 = at least 5 and at least 2 s in all)")
    ap.add_argument("--workload", default="n16_L6_dense", choices=sorted(WORKLOADS) + sorted(ADV_WORKLOADS))
    ap.add_argument("--series", default="auto", help="'auto': also measure n20_L8_kron (BASELINE config 4) beside the default "
                                                      "workload; 'none'; or a workload name")
    ap.add_argument("--dist-selftest", action="store_true", help="N > 1: check the sharded step against an un-sharded recomputation first")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gate-bench", action="store_true")
    ap.add_argument("--tile-bits", type=int, default=0)
    ap.add_argument("--debug-flags", type=int, default=0, help="timing-only kernel ablations (results invalid)")
    ap.add_argument("--wgs-per-cu", type=int, default=-1, help="generic circuit kernel: persistent workgroups per CU (0 = one per tile)")
    ap.add_argument("--fast-path", type=int, default=-1, help="0: force the generic circuit kernel (A/B against the fast one)")
    ap.add_argument("--fast-wgs-per-cu", type=int, default=-1, help="fast circuit kernel: persistent workgroups per CU (0 = occupancy query)")
    ap.add_argument("--opt", action="append", default=[], help="backend option name=value (e.g. low_bits=7), repeatable")
    ap.add_argument("--host-sync", type=int, default=0, help="1: loss.item() after every step (reference epoch); 0: read the losses back after the K steps")
    ap.add_argument("--prefix-share", type=int, default=0,
                    help="1: headline WITH prefix sharing of the shifted circuits (opt-in extra, SURVEY 8(f) row 4); the default "
                         "headline runs all 2P+1 circuits in full and reports the shared variant under 'extras'")
    ap.add_argument("--no-extras", action="store_true", help="skip the labelled extra leg (profiling runs: one variant per trace)")
    ap.add_argument("--graph", type=int, default=-1,
                    help="1: replay the step from one HIP graph (make_graphed_step); 0: eager launches; -1 (default): graph for "
                         "the latency-bound sizes n <= 13, eager above")
    ap.add_argument("--overlap", type=int, default=0,
                    help="how circuits and contraction share the GPU: 0 in sequence (default), 1 second plain stream, 2 two "
                         "CU-masked streams (half the CUs each), -1 measured choice between 0 and 2 (choose_overlap)")
    ap.add_argument("--device-adam", type=int, default=1, help="graph replay: 0 = torch's capturable Adam + scheduler (A/B)")
    ap.add_argument("--no-dist-selftest", action="store_true", help="N > 1: skip the sharded-vs-unsharded check (on by default)")
    args = ap.parse_args(argv)

    # ---- N > 1 without a rendezvous in the environment: start the N ranks ourselves.  The parent never touches the GPU
    # (no HIP call before this point: `import torch` and argparse only); the ranks are fresh child processes of
    # torch.distributed.run, rank 0's JSON line passes through on stdout, the exit code is the children's. ----
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            return launch_ranks(args.gpus, list(sys.argv[1:] if argv is None else argv))
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={os.environ['WORLD_SIZE']} ranks; "
                         "start one rank per GPU (python -m torch.distributed.run --nproc-per-node N bench.py --gpus N), or "
                         "run `python bench.py --gpus N` without a rendezvous in the environment and it starts them itself")

    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    ndev = torch.cuda.device_count()
    dev = torch.device("cuda", local_rank % ndev)     # (several ranks may share a GPU in a rehearsal)
    torch.cuda.set_device(dev)
    D = Dist(dev)
    world, rank = D.world, D.rank
    if world > 1 and not args.no_dist_selftest:
        args.dist_selftest = True

    from tensornetworks_amd import backend
    if args.tile_bits:
        backend.set_option(dev, "tile_bits", args.tile_bits)
    if args.debug_flags:
        backend.set_option(dev, "debug_flags", args.debug_flags)
    for kv in args.opt:
        name, val = kv.split("=")
        backend.set_option(dev, name, int(val))
    if args.wgs_per_cu >= 0:
        backend.set_option(dev, "workgroups_per_cu", args.wgs_per_cu)
    if args.fast_path >= 0:
        backend.set_option(dev, "fast_path", args.fast_path)
    if args.fast_wgs_per_cu >= 0:
        backend.set_option(dev, "fast_workgroups_per_cu", args.fast_wgs_per_cu)
    backend.set_option(dev, "prefix_share", 1 if args.prefix_share else 0)

    if args.workload in ADV_WORKLOADS:
        return bench_adversarial(args, dev, D)

    selftest = dist_selftest(dev, D) if (args.dist_selftest and world > 1) else None

    m = measure(args.workload, dev, D, args, args.steps, args.warmup, args.repeats,
                want_extras=not args.debug_flags and not args.no_extras)
    m["is_main"] = True
    m["workload"] = args.workload
    kern, config, phases = kernel_table(m, D, args)
    med_ms, rep = summarize(m, args.steps)
    phases_per_rank = D.gather_objects({"rank": rank, **phases})
    vi = m["vi"]
    rec = None
    if rank == 0:
        dom_name = max(kern, key=lambda k_: kern[k_]["ms_per_step"])
        roof = dict(kern[dom_name])
        roof["kernel"] = dom_name
        value = 1e3 / med_ms
        rec = {
            "metric": "ksd_gradient_steps_per_sec", "value": round(value, 4), "unit": "steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(med_ms, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": args.workload, **config, "bayesian_network": f"synthetic n={m['n']} seed=0 (SURVEY 8d)",
                       "optimizer": "adam lr=0.005 cosine clip=10",
                       "parallelism": f"paramshift + gram {'strip-pair' if vi._K_pairs is not None else 'row'} shard x{world}",
                       "dist_backend": D.backend if world > 1 else None},
            "repeats": rep, "roofline": roof, "kernels": kern,
            "extras": {"prefix_sharing": m["extra_share"], "adjoint_engine": m["extra_adjoint"]},
            "launch": "one HIP graph replay per step (make_graphed_step)" if m["graph"] else "eager kernel launches",
            "overlap": {"mode": {False: "sequential", True: "second stream", "partition": "cu-partition"}[vi.overlap_streams],
                        "measured_choice": vi.overlap_choice},
            "gram_placement": vi.gram_placement,
            "phase_ms": {**phases, "note": "event spans on the stream the kernels are launched on, mean over the timed region; "
                                           "allreduce / allgather are inside stein / finish"},
            "phase_ms_per_rank": phases_per_rank,
            "precompute_seconds": round(m["precompute_s"], 3),
            "host_sync": "per step (loss.item())" if args.host_sync else "deferred: K steps back to back, losses read after the repeat",
            "loss_first_last": [float(m["losses"][0]), float(m["losses"][-1])],
        }
        if selftest is not None:
            rec["dist_selftest"] = selftest
    if rank == 0 and world == 1 and m["gram_mode"] == "dense" and not args.no_extras and not args.debug_flags and vi._K is not None:
        # the two matrix-core kernels of the path on this run's own K_p (MFMA utilisation against the fp64 matrix peak)
        rec["extras"]["mfma"] = mfma_extras(vi, dev, m["n"])
    S_host = vi._S.cpu().numpy() if (rank == 0 and world == 1 and not args.no_cpu_baseline) else None
    theta0, main_n = m["theta0"], m["n"]
    main_cfg = (m["n"], m["layers"], m["ansatz"], m["gram_mode"])
    del vi, m
    backend.release_workspaces()
    torch.cuda.empty_cache()

    # ---- beside the headline, same protocol, at every N: BASELINE config 4 (n = 20, L = 8, matrix-free), the headline's
    # circuit with the matrix-free contraction (what gram_mode="auto" weighs against the dense form) and config 2 (n = 8) ----
    series_names = {"auto": list(SERIES_WORKLOADS) if args.workload == "n16_L6_dense" else [], "none": []}.get(args.series, [args.series])
    if args.debug_flags:
        series_names = []
    for series_name in series_names:
        if series_name not in WORKLOADS:
            raise SystemExit(f"unknown --series workload {series_name}")
        ks = max(3, min(args.steps, 10))
        ms_ = measure(series_name, dev, D, args, ks, 2, 3, want_extras=False)
        ms_["workload"] = series_name
        kern_s, config_s, phases_s = kernel_table(ms_, D, args)
        med_s, rep_s = summarize(ms_, ks)
        pr_s = D.gather_objects({"rank": rank, **phases_s})
        if rank == 0:
            entry = {"workload": series_name, "metric": "ksd_gradient_steps_per_sec", "value": round(1e3 / med_s, 4),
                     "unit": "steps/s", "n_gpus": world, "ms_per_step": round(med_s, 4), "repeats": rep_s,
                     "config": config_s, "kernels": kern_s, "phase_ms": phases_s, "phase_ms_per_rank": pr_s,
                     "launch": "one HIP graph replay per step (make_graphed_step)" if ms_["graph"] else "eager kernel launches",
                     "scaling": "strong", "loss_first_last": [float(ms_["losses"][0]), float(ms_["losses"][-1])]}
            dom_s = max(kern_s, key=lambda k_: kern_s[k_]["ms_per_step"])
            entry["roofline"] = {**kern_s[dom_s], "kernel_row": dom_s}
            if world == 1 and not args.no_cpu_baseline and series_name == SERIES_WORKLOADS[0]:
                # the same step on the host cores (oracle's C port: circuits + the contraction in the form the GPU leg uses)
                n_s, layers_s, ansatz_s, gram_s = WORKLOADS[series_name]
                cb = cpu_baseline(n_s, layers_s, ansatz_s, gram_s, ms_["vi"]._S.cpu().numpy(), ms_["theta0"], max_params=48)
                entry["cpu_baseline"] = cb
                if cb:
                    entry["gpu_over_cpu"] = round(entry["value"] / cb["value"], 1)
            rec.setdefault("series", []).append(entry)
        del ms_
        backend.release_workspaces()
        torch.cuda.empty_cache()

    if rank == 0:
        if world == 1 and not args.no_gate_bench:
            rec["gate_apply"] = gate_apply_microbench(dev, n=min(main_n, 16) if main_n >= 10 else 16)
        if world == 1 and not args.no_cpu_baseline:
            n_, layers_, ansatz_, gram_ = main_cfg
            rec["cpu_baseline"] = cpu_baseline(n_, layers_, ansatz_, gram_, S_host, theta0)
            if rec["cpu_baseline"]:
                rec["gpu_over_cpu"] = round(rec["value"] / rec["cpu_baseline"]["value"], 1)
        print(json.dumps(rec))
    D.close()


if __name__ == "__main__":
    main()
