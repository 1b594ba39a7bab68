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

N > 1 (strong scaling: the same step, split): the 2P shifted circuits and the rows of K_p are
sharded over the ranks; two small all-gathers per step (K q rows, gradient scalars) over RCCL.
"""
import argparse
import contextlib
import io
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)

WORKLOADS = {
    # name: (n, layers, ansatz, gram_mode)
    "n16_L6_dense": (16, 6, "hardware_efficient", "dense"),
    "n16_L6_kron": (16, 6, "hardware_efficient", "kron"),
    "n8_L4_dense": (8, 4, "hardware_efficient", "dense"),
    "n20_L8_kron": (20, 8, "hardware_efficient", "kron"),
    "n12_L4_dense": (12, 4, "hardware_efficient", "dense"),
}


def mean_ms(pairs):
    return float(np.mean([a.elapsed_time(b) for a, b in pairs])) if pairs else 0.0


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
        out["gbs"][name] = round(out["bytes_per_launch"] / (ms * 1e-3) / 1e9, 1)
    vals = [v for k, v in out["gbs"].items() if k.startswith("ry")]
    out["ry_mean_gbs"] = round(float(np.mean(vals)), 1)
    out["frac_of_hbm_peak"] = round(out["ry_mean_gbs"] / HBM_PEAK_GBS, 4)
    del st
    return out


def cpu_baseline(n, layers, ansatz, S_host, theta64, total_circuits):
    """The oracle's C port (oracle/cpu_port.c, OpenMP) on the host cores, on a bounded sample of the
    same step, extrapolated linearly in circuits and Gram rows.  A reported baseline, not a target."""
    from oracle import cpu_port as cp
    if not cp.available():
        return None
    T = cp.max_threads()
    P = theta64.size
    npar = max(1, min(P, 2 * T))
    t0 = time.perf_counter()
    probs, used = cp.paramshift_probs(ansatz, n, layers, theta64, 0, npar, include_base=True)
    t_circ = time.perf_counter() - t0
    n_circ = probs.shape[0]
    rows = min(1 << n, 1024)
    K_rows = cp.gram_rows(S_host, n, 1.0, 0, rows)            # building K is outside the step
    q = np.ascontiguousarray(probs[0])
    cp.gemv_rows(K_rows, n, 0, rows, q)                         # warm
    t_gemv = float("inf")                                       # best of 7: the 0.5 ms sample is at the mercy of one
    for _ in range(7):                                          # descheduled OpenMP thread (seen: 30 ms once)
        t0 = time.perf_counter()
        cp.gemv_rows(K_rows, n, 0, rows, q)
        t_gemv = min(t_gemv, time.perf_counter() - t0)
    step_s = t_circ * total_circuits / n_circ + t_gemv * (1 << n) / rows
    return {"value": round(1.0 / step_s, 6), "unit": "steps/s", "cores": int(used), "kind": "port",
            "host_cpus": os.cpu_count(),
            "sample": f"{n_circ} of {total_circuits} circuits ({t_circ:.2f} s) + {rows} of {1 << n} Gram rows GEMV "
                      f"({t_gemv * 1e3:.1f} ms), oracle/cpu_port.c with OpenMP, extrapolated linearly",
            "est_step_seconds": round(step_s, 3)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="n16_L6_dense", choices=sorted(WORKLOADS))
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
    ap.add_argument("--overlap", type=int, default=0,
                    help="how circuits and contraction share the GPU: 0 in sequence (default), 1 second plain stream, 2 two "
                         "CU-masked streams (half the CUs each), -1 measured choice between 0 and 2 (choose_overlap)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    ndev = torch.cuda.device_count()
    dev = torch.device("cuda", local_rank % ndev)     # (several ranks may share a GPU in a rehearsal)
    torch.cuda.set_device(dev)
    import torch.distributed as dist
    dist_backend = os.environ.get("BORNVI_DIST_BACKEND", "nccl")   # "gloo": rehearsal with ranks sharing one GPU
    if world > 1:
        if dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(dist_backend)
    if args.gpus != world and rank == 0:
        print(f"[bench] note: --gpus {args.gpus} but WORLD_SIZE={world}; using {world}", file=sys.stderr)

    from tensornetworks_amd import backend, _ext
    from tensornetworks_amd.bayesian_network import synthetic_network
    from tensornetworks_amd.ksd_vi_quantum import KSDVariationalInference

    n, layers, ansatz, gram_mode = WORKLOADS[args.workload]
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
    bn, lat, obs, x = synthetic_network(n, seed=0)
    torch.manual_seed(0)
    vi = KSDVariationalInference(bn, lat, obs, qbm_num_latent_vars=n, qbm_ansatz_layers=layers,
                                 qbm_ansatz_type=ansatz, pytorch_device=str(dev), gram_mode=gram_mode)
    if args.overlap >= 0:
        vi.overlap_streams = {0: False, 1: True, 2: "partition"}[args.overlap]
    g = torch.Generator().manual_seed(0)
    P = vi.born_machine.num_ansatz_params
    with torch.no_grad():     # theta0 = 0.1 * randn(P) float32, `small_random` (quantum_born_machine.py:43-45)
        vi.born_machine.theta.copy_((0.1 * torch.randn(P, generator=g, dtype=torch.float32)).to(dev))
    theta0 = vi.born_machine.theta.detach().double().cpu().numpy()

    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    vi._prepare_stein(x)
    torch.cuda.synchronize(dev)
    precompute_s = time.perf_counter() - t0
    if args.overlap < 0:
        vi.choose_overlap()

    total_steps = args.steps + args.warmup
    params, opt, sched = vi.make_optimizer(0.005, total_steps, True, "adam", (0.9, 0.999))
    clip = 10.0
    # --host-sync 1: every step ends with loss.item() like the reference's epoch (the GPU idles while the host
    # handles it); 0 (default): the same steps with the read-back of the K losses deferred to the end of the timed
    # region (NaN/Inf guard on the device), so the K steps run back to back
    step_fn = vi.training_step if args.host_sync else vi.training_step_async
    losses = []
    for _ in range(args.warmup):
        losses.append(step_fn(params, opt, sched, clip)[0])

    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    vi.timers = {}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses.append(step_fn(params, opt, sched, clip)[0])
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    timers = vi.timers
    vi.timers = None
    # labelled extra: the same K steps with the other setting of prefix sharing (bit-identical shifted distributions,
    # fewer circuit-passes); not part of `value`
    extra_share = None
    if not args.debug_flags and not args.no_extras:
        backend.set_option(dev, "prefix_share", 0 if args.prefix_share else 1)
        for _ in range(2):
            step_fn(params, opt, sched, clip)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)
        vi.timers = {}
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step_fn(params, opt, sched, clip)
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        e2 = time.perf_counter() - t1
        if world > 1:
            tt = torch.tensor([e2], dtype=torch.float64, device=dev if dist_backend == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            e2 = float(tt.item())
        extra_share = {"prefix_share": 0 if args.prefix_share else 1, "steps_per_sec": round(args.steps / e2, 4),
                       "ms_per_step": round(1e3 * e2 / args.steps, 4), "circuits_ms": round(mean_ms(vi.timers.get("circuits")), 4),
                       "note": "same step with prefix sharing of the parameter-shift batch switched "
                               + ("off" if args.prefix_share else "on") + " (opt-in, bornvi_set_option prefix_share): a shifted "
                               "circuit starts from the base circuit's state at the first pass its parameter touches; rows bit-identical"}
        vi.timers = None
        backend.set_option(dev, "prefix_share", 1 if args.prefix_share else 0)
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev if dist_backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    if rank == 0:
        plan = _ext.plan_words(_ext.ANSATZ_IDS[ansatz], n, layers, args.tile_bits)   # same defaults as the handle
        n_passes, n_gates = int(plan[3]), int(plan[11])
        circuits_rank = 1 + 2 * len(range(0, P, world))            # rank 0: the parameters 0, W, 2W, ...
        circ_ms, stein_ms, fin_ms = mean_ms(timers.get("circuits")), mean_ms(timers.get("stein")), mean_ms(timers.get("finish"))
        base_ms = mean_ms(timers.get("base_circuit"))     # > 0 only in overlap mode: base circuit launched separately
        circ_launches = n_passes * (2 if base_ms > 0 else 1)
        circ_kernel_ms = circ_ms + base_ms
        N = 1 << n
        # ALGORITHMIC bytes per launch = what the kernel's own algorithm has to move through HBM:
        #   circuit pass: every state of the batch in (16 * 2^n; none in the first pass) and out (16 * 2^n, or
        #                 8 * 2^n probabilities in the last pass); the SURVEY 8(d) UN-FUSED accounting (32 * 2^n per
        #                 gate per state) is reported beside it as `survey_8d_unfused_equivalent_gbs`
        #   contraction:  the upper triangle of K_p, 4 * 2^n * (2^n + 32) bytes per GPU share (SURVEY 8(d) counts the
        #                 full matrix, 8 * 4^n: `survey_8d_full_matrix_gbs`), or the rank's rows for the row shard
        unfused_bytes = 32.0 * N * n_gates * circuits_rank
        # prefix sharing: a shifted circuit joins the batch in the first pass its parameter touches (it reads the base
        # circuit's state there), so pass i moves only the circuits already active
        first_pass = _ext.plan_param_first_pass(plan)[list(range(0, P, world))] if n_passes > 1 else None
        share_on = n_passes > 1 and bool(args.prefix_share)
        active = [1 + 2 * int((first_pass <= i).sum()) if share_on else circuits_rank for i in range(n_passes)] if n_passes > 1 else [circuits_rank]
        circ_bytes = sum(a * ((16.0 * N if i > 0 else 0.0) + (16.0 * N if i < n_passes - 1 else 8.0 * N)) for i, a in enumerate(active))
        circuit_passes_run = sum(active)
        rows_rank = -(-N // world)
        sym = gram_mode == "dense" and vi.symmetric_contraction and (world == 1 or vi._K_pairs is not None)
        stein_name = ("quadform_sym_kernel" if sym else "quadform_kernel") if gram_mode == "dense" else "kron_matvec"
        full_bytes = 8.0 * N * rows_rank
        if gram_mode != "dense":
            stein_bytes = 16.0 * N * n * (n + 1)
        elif sym:
            stein_bytes = 4.0 * N * (N + 32) / world
        else:
            stein_bytes = full_bytes
        # real HBM traffic per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (tools/pmc_traffic.sh +
        # tools/pmc_summarize.py), recorded for the default workload only
        pmc = {}
        pmc_file = os.path.join(REPO, "profiles", "r01_pmc_traffic_n16_L6_dense.json")
        if args.workload == "n16_L6_dense" and world == 1 and not args.tile_bits and os.path.exists(pmc_file):
            pmc = json.load(open(pmc_file))["kernels"]
        t_circ = None
        for kname, kv in pmc.items():
            if kname.startswith("circuit_pass"):
                t_circ = kv.get("hbm_bytes_per_launch")
        t_stein = None
        if sym and "quadform_sym_kernel" in pmc:
            t_stein = pmc["quadform_sym_kernel"]["hbm_bytes_per_launch"] + pmc["quadform_sym_reduce_kernel"]["hbm_bytes_per_launch"]
        kern = {
            "circuit_pass_kernel": {"bound": "hbm", "launches_per_step": circ_launches,
                                    "achieved": round(circ_bytes / (circ_kernel_ms * 1e-3) / 1e9, 1) if circ_ms else None,
                                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "ms_per_step": round(circ_kernel_ms, 4),
                                    "avg_launch_ms": round(circ_kernel_ms / circ_launches, 4),
                                    "algorithmic_bytes_per_launch": circ_bytes / circ_launches, "traffic": t_circ,
                                    "circuit_passes_run": circuit_passes_run, "circuit_passes_without_prefix_sharing": circuits_rank * n_passes,
                                    "survey_8d_unfused_equivalent_gbs": round(unfused_bytes / (circ_kernel_ms * 1e-3) / 1e9, 1) if circ_ms else None,
                                    "note": "fused LDS-tiled engine: algorithmic bytes = each ACTIVE state read and written once per "
                                            "pass (a shifted circuit starts from the base circuit's state at the first pass its parameter touches); traffic = measured HBM bytes per launch (PMC); the same work as un-fused "
                                            "gate-apply (SURVEY 8(d): 32 * 2^n bytes per gate per state) would need the "
                                            "survey_8d_unfused_equivalent_gbs rate"},
            stein_name: {
                "bound": "hbm", "launches_per_step": 1,
                "achieved": round(stein_bytes / (stein_ms * 1e-3) / 1e9, 1) if stein_ms else None,
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "ms_per_step": round(stein_ms, 4),
                "avg_launch_ms": round(stein_ms, 4), "algorithmic_bytes_per_launch": stein_bytes, "traffic": t_stein,
                "survey_8d_full_matrix_gbs": round(full_bytes / (stein_ms * 1e-3) / 1e9, 1) if (sym and stein_ms) else None,
                "note": ("K_p is bitwise symmetric: only its upper triangle is read (algorithmic bytes = 4 * 2^n * (2^n + 32)); "
                         "ms includes the column-partial reduce and the final sum; traffic = measured HBM bytes (PMC)") if sym else None},
        }
        for v in kern.values():
            v["frac"] = round(v["achieved"] / v["peak"], 4) if v["achieved"] else None
            if v["traffic"] and v["ms_per_step"]:
                per_step = v["traffic"] * v["launches_per_step"]
                v["traffic_gbs"] = round(per_step / (v["ms_per_step"] * 1e-3) / 1e9, 1)
                v["traffic_frac"] = round(v["traffic_gbs"] / v["peak"], 4)
        dom_name = max(kern, key=lambda k_: kern[k_]["ms_per_step"])
        roof = dict(kern[dom_name])
        roof["kernel"] = dom_name
        value = args.steps / elapsed
        rec = {
            "metric": "ksd_gradient_steps_per_sec", "value": round(value, 4), "unit": "steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": args.workload, "n_qubits": n, "layers": layers, "ansatz": ansatz,
                       "params": P, "circuits_per_step": 1 + 2 * P, "gates_per_circuit": n_gates,
                       "gram": gram_mode, "bayesian_network": f"synthetic n={n} seed=0 (SURVEY 8d)",
                       "optimizer": "adam lr=0.005 cosine clip=10",
                       "parallelism": f"paramshift + gram {'strip-pair' if vi._K_pairs is not None else 'row'} shard x{world}",
                       "dist_backend": dist_backend if world > 1 else None,
                       "tile_bits": int(plan[2]), "passes": n_passes},
            "roofline": roof, "kernels": kern, "extras": {"prefix_sharing": extra_share},
            "overlap": {"mode": {False: "sequential", True: "second stream", "partition": "cu-partition"}[vi.overlap_streams],
                        "measured_choice": vi.overlap_choice},
            "gram_placement": vi.gram_placement,
            "phase_ms": {"circuits": round(circ_ms, 4), "base_circuit": round(base_ms, 4), "stein": round(stein_ms, 4),
                         "finish": round(fin_ms, 4),
                         "note": "event spans; with the contraction on a second stream the 'circuits' and "
                                 "'base_circuit'+'stein' spans overlap in time" if base_ms > 0 else "event spans"},
            "precompute_seconds": round(precompute_s, 3),
            "host_sync": "per step (loss.item())" if args.host_sync else "deferred: K steps back to back, losses read after the timed region",
            "loss_first_last": [float(losses[0]), float(losses[-1])],
        }
        if world == 1 and not args.no_gate_bench:
            vi._K = None
            torch.cuda.empty_cache()
            rec["gate_apply"] = gate_apply_microbench(dev, n=min(n, 16) if n >= 10 else 16)
        if world == 1 and not args.no_cpu_baseline:
            S_host = vi._S.cpu().numpy()
            rec["cpu_baseline"] = cpu_baseline(n, layers, ansatz, S_host, theta0, 1 + 2 * P)
            if rec["cpu_baseline"]:
                rec["gpu_over_cpu"] = round(value / rec["cpu_baseline"]["value"], 1)
        print(json.dumps(rec))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
