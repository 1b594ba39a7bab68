"""KSD variational inference with the quantum Born machine on MI355X.

Drop-in for the reference trainer (ksd_vi_quantum.py:18-191): same constructor and `train`
signatures, attributes, history keys, printed messages and update rule.  What changed is where the
arithmetic happens:

  reference epoch body (ksd_vi_quantum.py:110-161)          here
  ------------------------------------------------          -------------------------------------------
  q = pqc(theta)                    1 PennyLane circuit     \\
  N^2 get_stein_kernel_kp_value calls (K_p rebuilt every     } one batched HIP launch sequence:
     epoch although it does not depend on theta)            /  base + 2P shifted circuits (LDS-tiled),
  loss = sqrt(clamp(sum, 1e-12)); loss.backward()              y = K_p q (dense GEMV or Kronecker mat-vec),
     -> autograd over N^2 terms, then 2P PennyLane circuits    loss, grad_p = 1/2 (y/loss).(q+_p - q-_p)
  clip_grad_norm_, optimizer.step(), scheduler.step()       same torch.optim objects

S (scores) and K_p are computed once per `train()` call on the GPU.  With a torch.distributed
process group the 2P shifted circuits are sharded over the ranks (paramshift_shard.py).
"""
import os
import time
from functools import partial

import numpy as np
import torch
import torch.nn.utils as nn_utils
import torch.optim as optim

from . import backend
from . import paramshift_shard as shard
from .quantum_born_machine import QuantumBornMachine
from .stein_utils import base_hamming_kernel_torch, score_matrix, stein_gram_matrix, tvd_table
from .utils import calculate_tvd, generate_all_binary_outcomes

_ROCTX = os.environ.get("BORNVI_ROCTX", "0") == "1"
DENSE_GRAM_MAX_N = 16     # 8 * 4^16 bytes = 32 GiB of the 288 GB HBM; beyond that the matrix-free form


class _EventSpan:
    """`with` block that records a (start, end) torch.cuda.Event pair on the current stream -- the
    stream every bornvi kernel of the block is launched on -- when timers are enabled."""

    def __init__(self, timers, name):
        self.timers, self.name = timers, name

    def __enter__(self):
        if _ROCTX:                       # BORNVI_ROCTX=1: named ranges for rocprofv3 --marker-trace / roctx consumers
            torch.cuda.nvtx.range_push(f"bornvi:{self.name}")
        if self.timers is not None:
            self.ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            self.ev[0].record()
        return self

    def __exit__(self, *exc):
        if _ROCTX:
            torch.cuda.nvtx.range_pop()
        if self.timers is not None:
            self.ev[1].record()
            self.timers.setdefault(self.name, []).append(self.ev)
        return False


def cosine_annealing_lr(epoch, base_lr, T_max, eta_min):
    """optim.lr_scheduler.CosineAnnealingLR(T_max, eta_min) after `epoch` scheduler steps (scalar or array): its closed
    form, which its recursive form follows to rounding, beyond T_max as well (both are periodic in 2 T_max)."""
    e = np.asarray(epoch, dtype=np.float64)
    v = np.where(e == 0, base_lr, eta_min + (base_lr - eta_min) * (1.0 + np.cos(np.pi * e / T_max)) / 2.0)
    return v if e.shape else float(v)


class DeviceAdam:
    """optim.Adam(lr, betas) + CosineAnnealingLR(T_max, eta_min) + clip_grad_norm_ + the NaN/Inf guard as ONE launch per
    epoch (backend.clip_adam_step): moments, step count and epoch count live on the device, the schedule is a table the
    kernel indexes with its own epoch count.  For the HIP-graph replay of the latency-bound sizes, where torch's fused
    optimiser, its tensor-valued schedule and the float64 cast of theta were 9 of the step's 15 launches."""

    def __init__(self, theta, lr, betas=(0.9, 0.999), eps=1e-8, T_max=None, eta_min=0.0, capacity=1 << 16):
        if not (theta.is_cuda and theta.dtype == torch.float32 and theta.is_contiguous()):
            raise backend.BornviError("DeviceAdam needs a contiguous float32 theta on the GPU")
        dev = theta.device
        self.theta, self.base_lr, self.betas, self.eps = theta, float(lr), (float(betas[0]), float(betas[1])), float(eps)
        self.T_max, self.eta_min, self.capacity = (None if T_max is None else int(T_max)), float(eta_min), int(capacity)
        if theta.grad is None:
            theta.grad = torch.zeros_like(theta)
        self.exp_avg = torch.zeros(theta.numel(), dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros_like(self.exp_avg)
        self.counters = torch.zeros(2, dtype=torch.int32, device=dev)        # [good steps, epochs since the table's start]
        self.theta64 = theta.detach().to(torch.float64).clone()              # the circuits' input, kept current by the kernel
        self.norm = torch.zeros((), dtype=torch.float32, device=dev)
        self.lr_table = torch.empty(self.capacity, dtype=torch.float64, device=dev)
        self.loss_hist = torch.zeros(self.capacity, dtype=torch.float64, device=dev)    # written by the kernel, one entry
        self.norm_hist = torch.zeros(self.capacity, dtype=torch.float32, device=dev)    # per epoch of the window
        self._hist_done = []                 # (losses, norms) of the windows before the current one
        self.epochs = 0                      # epochs run so far (host count)
        self._window = 0                     # epoch of lr_table[0]
        self._fill()

    def lr_at(self, epoch):
        """Learning rate of epoch `epoch` (0-based): CosineAnnealingLR's closed form (periodic beyond T_max, like its
        recursive form), or the constant rate without a schedule."""
        if self.T_max is None:
            shape = np.shape(epoch)
            return np.full(shape, self.base_lr) if shape else self.base_lr
        return cosine_annealing_lr(epoch, self.base_lr, self.T_max, self.eta_min)

    def _fill(self):
        vals = self.lr_at(self._window + np.arange(self.capacity))
        self.lr_table.copy_(torch.from_numpy(np.ascontiguousarray(vals)))

    def step(self, grad64, loss, max_norm):
        """One epoch's hand-off (enqueued on the current stream; capturable).  -> the gradient norm, a device scalar
        owned by this object.  Follow it with advance() on the host."""
        return backend.clip_adam_step(grad64.reshape(-1), max_norm, loss, self.theta.view(-1), self.theta.grad.view(-1),
                                      self.theta64.view(-1), self.exp_avg, self.exp_avg_sq, self.counters, self.lr_table,
                                      self.betas[0], self.betas[1], self.eps, norm_out=self.norm,
                                      loss_history=self.loss_hist, norm_history=self.norm_hist)

    def advance(self):
        """Host side of an epoch (after step() or a replay of it): counts it, and moves the schedule table on when the
        device's epoch count is about to run off its end."""
        self.epochs += 1
        if self.epochs - self._window >= self.capacity:
            self._hist_done.append((self.loss_hist.clone(), self.norm_hist.clone()))
            self._window = self.epochs
            self._fill()
            self.counters[1:].zero_()

    def history(self, begin=0, end=None):
        """(losses float64, gradient norms float32) of epochs [begin, end) as device tensors: what the kernel recorded
        (the reference's per-epoch loss.item() and clip_grad_norm_ values, ksd_vi_quantum.py:163-166)."""
        end = self.epochs if end is None else int(end)
        k = self.epochs - self._window
        losses = torch.cat([c[0] for c in self._hist_done] + [self.loss_hist[:k]])
        norms = torch.cat([c[1] for c in self._hist_done] + [self.norm_hist[:k]])
        return losses[begin:end], norms[begin:end]

    def last_lr(self):
        """What scheduler.get_last_lr()[0] reads after this many epochs: the rate of the next one."""
        return self.lr_at(self.epochs)


class KSDVariationalInference:
    def __init__(self,
                 bayesian_network,
                 latent_vars_names: list,
                 observed_vars_names: list,
                 qbm_num_latent_vars: int,
                 qbm_ansatz_layers: int = 1,
                 qbm_conditioning_dim: int = 0,
                 qbm_pennylane_device_name: str = "default.qubit",
                 qbm_ansatz_type: str = "hardware_efficient",
                 qbm_init_method: str = "small_random",
                 base_kernel_length_scale: float = 1.0,
                 pytorch_device: str = 'cpu',
                 *, gram_mode: str = "auto", process_group=None):
        """Arguments up to `pytorch_device` are the reference's (ksd_vi_quantum.py:19-30).
        Keyword-only extras: gram_mode in {"auto", "dense", "kron"} (dense Gram matrix vs matrix-free
        Kronecker mat-vec; "auto" = dense up to n = 16); process_group = torch.distributed group over
        which the parameter-shift circuits are sharded (None = default group if initialised;
        paramshift_shard.SOLO = never shard)."""
        if int(qbm_num_latent_vars) != len(latent_vars_names):
            # the scores are [2^len(latent_vars_names), len(latent_vars_names)] while the circuit has
            # qbm_num_latent_vars qubits: the device kernels would index one with the other's sizes
            raise ValueError(f"qbm_num_latent_vars ({qbm_num_latent_vars}) must equal len(latent_vars_names) "
                             f"({len(latent_vars_names)})")
        self.bn = bayesian_network
        self.latent_vars_names = latent_vars_names
        self.observed_vars_names = observed_vars_names
        self.num_latent_vars = qbm_num_latent_vars
        self.num_observed_vars = len(observed_vars_names)
        self.pytorch_device = pytorch_device
        self.base_kernel_length_scale = base_kernel_length_scale
        if gram_mode not in ("auto", "dense", "kron"):
            raise ValueError("gram_mode must be 'auto', 'dense' or 'kron'")
        self.gram_mode = gram_mode
        self.process_group = process_group

        self.born_machine = QuantumBornMachine(
            num_latent_vars=self.num_latent_vars,
            ansatz_layers=qbm_ansatz_layers,
            conditioning_dim=qbm_conditioning_dim,
            device_name=qbm_pennylane_device_name,
            ansatz_type=qbm_ansatz_type,
            init_method=qbm_init_method
        ).to(pytorch_device)
        self.born_machine.process_group = process_group

        self._all_states = None
        self.num_possible_latent_states = 2 ** self.num_latent_vars

        self.base_kernel_func = partial(base_hamming_kernel_torch,
                                        num_vars=self.num_latent_vars,
                                        length_scale=base_kernel_length_scale)

        self._score_function_cache = {}
        self._S = None          # scores [2^n, n] on the GPU
        self._K = None          # dense Gram (dense mode): all rows, or this rank's row block when sharded
        self._K_rows = None     # (row_begin, row_end) held in self._K (row shard)
        self._K_pairs = None    # (pair_begin, pair_end, rows of the lower block) held in self._K (strip-pair shard)
        self._stein_key = None
        self.timers = None      # optional {name: [(start_event, end_event), ...]} filled by ksd_and_grad
        self.symmetric_contraction = True   # dense mode: contract with the upper triangle of K_p only
        # How the shifted circuits and the contraction of a step share the GPU (they need nothing from each other):
        #   False        in sequence on the current stream
        #   True         contraction on a second plain stream: paid with the one-workgroup-per-tile circuit kernel; the
        #                persistent one fills every CU's registers and LDS, so the contraction's waves only get in
        #                between passes (5.4-6.3 ms against 5.5 ms in sequence at n = 16)
        #   "partition"  two CU-masked streams: the instruction-bound circuit passes on one half of the CUs, the
        #                HBM-bound contraction on the other half (bornvi_stream_create_cu_range)
        #                (measured on the MI355X at n = 16: 5.43 ms against 5.36 ms in sequence -- each kernel alone
        #                keeps ~80 % of its speed on half the CUs, but together they contend for HBM; no gain)
        #   None         = False.  `choose_overlap()` decides between False and "partition" by measurement.
        self.overlap_streams = None
        self.overlap_choice = None
        # Gradient engine.  "paramshift" (default) = the reference's rule: 2P shifted circuit evaluations
        # (diff_method="parameter-shift", quantum_born_machine.py:58).  "adjoint" = OPT-IN extra:
        # one forward and one backward walk over the gates (bornvi_adjoint_state / _vjp) -- the same gradient to
        # rounding from about three circuit evaluations; every rank computes it whole (nothing to shard).
        self.grad_engine = "paramshift"
        # The parameter-shift dot product  sum_z dL/dq_z (q+ - q-)(z)  inside the shifted circuits' last pass instead of a
        # pass over their stored probabilities (bornvi_paramshift_dot_begin / _finish): still all 2P circuit evaluations,
        # same gradient to rounding; taken where the library offers it (multi-pass plans of the 8-amplitude kernel),
        # else the probabilities are written and dotted as before.  False: always the un-fused path (A/B).
        self.fused_dot = True
        # A dense K_p >= 1 GiB is placed by measurement: up to this many copies are built (each in fresh memory while the
        # earlier ones are held), the contraction is timed on each, the fastest stays (_place_gram).  The contraction's
        # rate depends on where the driver put K_p RELATIVE to the workspace its partial sums go to -- 2.55 or 2.78 ms
        # at n = 16 for the same kernel and matrix, stable for the life of the allocations, equal alone and inside the
        # training step (tools/probes/step_placement_probe.py, ws_place_probe.py, ws_far_probe.py) -- and a process
        # cannot see physical addresses.  One-time cost: ~50 ms and 2^(2n+3) bytes per extra copy, freed at once; the
        # search stops at the first pair that streams at 83 % of the HBM peak.  1 = take the first copy.
        self.gram_placement_tries = 4
        self.gram_placement = None          # {"contraction_ms_per_pair": [[copy, workspace, ms], ...], "kept": [copy, workspace]}
        self._aux_stream = None

    # ---- reference attribute kept lazily (2^n Python tuples) -------------------------------------------
    @property
    def all_latent_states_tuples(self):
        if self._all_states is None:
            self._all_states = generate_all_binary_outcomes(self.num_latent_vars)
        return self._all_states

    def _get_precomputed_s_p(self, z_tuple, x_dict):
        """Score vector of one state (reference :58-68), served from the batched device result."""
        if z_tuple in self._score_function_cache:
            return self._score_function_cache[z_tuple]
        if self._S is None or self._stein_key != self._key(x_dict):
            self._prepare_stein(x_dict, announce=False)
        idx = 0
        for b in z_tuple:
            idx = (idx << 1) | int(b)
        s = self._S[idx].to(self.pytorch_device)
        self._score_function_cache[z_tuple] = s
        return s

    def _precompute_all_s_p(self, x_dict):
        """reference :70-75 -- one kernel launch instead of 2^n * (n+1) network enumerations."""
        self._score_function_cache.clear()
        print("Precomputing score functions s_p(x,z)...")
        self._prepare_stein(x_dict, announce=False)
        print("Score functions precomputed.")

    def _key(self, x_dict):
        return tuple(sorted((x_dict or {}).items()))

    def _use_dense(self):
        if self.gram_mode == "dense":
            return True
        if self.gram_mode == "kron":
            return False
        return self.num_latent_vars <= DENSE_GRAM_MAX_N

    def _prepare_stein(self, x_dict, announce=True):
        """Scores and (dense mode) the Gram matrix, once per observation.  With W > 1 ranks each rank
        builds and keeps only its block of N/W rows of K_p (row shard of the quadratic form)."""
        dev = backend.compute_device(self.pytorch_device)
        n = self.num_latent_vars
        S_new = score_matrix(self.bn, x_dict, self.latent_vars_names, device=dev)
        # K_p is a function of (S, n, length scale) only: a second train() on the same observation and network keeps the
        # matrix it has (32 GiB and a placement search at n = 16) -- the scores themselves are recomputed like the
        # reference does (one launch)
        sig = (float(self.base_kernel_length_scale), self._use_dense(), bool(self.symmetric_contraction),
               shard.world(self.process_group), int(self.gram_placement_tries))
        if (self._use_dense() and getattr(self, "_K", None) is not None and getattr(self, "_K_sig", None) == sig
                and self._S is not None and self._S.shape == S_new.shape and torch.equal(self._S, S_new)):
            self._S = S_new
            self._stein_key = self._key(x_dict)
            return
        self._S = S_new
        self._K_sig = sig
        self._K = None
        self._K_rows = None
        self._K_pairs = None
        if self._use_dense():
            rank, ws = shard.world(self.process_group)
            sp = backend.sym_pair_shard(n, rank, ws) if (ws > 1 and self.symmetric_contraction) else None
            if sp is not None:
                # strip-pair shard of the symmetric contraction: this rank keeps two row blocks of K_p (a long and
                # a short part of the upper triangle) and reads only 1/W of the triangle per step
                (pa, pb), (l0, l1), (h0, h1) = sp
                self._K_pairs = (pa, pb, l1 - l0)

                def build():
                    # (padded row pitch: backend.gram_ld -- the strips' row streams must not share an HBM channel)
                    K = torch.empty(((l1 - l0) + (h1 - h0), backend.gram_ld(n)), dtype=torch.float64, device=dev)[:, : 1 << n]
                    if l1 > l0:
                        backend.stein_gram(self._S, n, self.base_kernel_length_scale, rows=(l0, l1), out=K[: l1 - l0])
                        backend.stein_gram(self._S, n, self.base_kernel_length_scale, rows=(h0, h1), out=K[l1 - l0:])
                    return K

                def contract(K, q):
                    return backend.stein_quadform_sym_pairs(K[: l1 - l0], K[l1 - l0:], pa, pb, q, n)
            else:
                self._K_rows = shard.shard_range(1 << n, rank, ws)
                r0, r1 = self._K_rows

                def build():
                    # one GPU, symmetric contraction: padded row pitch (backend.gram_ld); the full-matrix and row-shard
                    # kernels read contiguous rows
                    pad = ws == 1 and self.symmetric_contraction
                    return backend.stein_gram(self._S, n, self.base_kernel_length_scale, rows=self._K_rows,
                                              ld=backend.gram_ld(n) if pad else None)

                def contract(K, q):
                    if ws == 1:
                        return backend.stein_quadform_sym(K, q, n) if self.symmetric_contraction else backend.stein_quadform(K, q, n, want_y=True)
                    return backend.stein_quadform_rows(K, r0, r1, q, n)
            sym = self.symmetric_contraction and (ws == 1 or sp is not None)
            self._K = self._place_gram(build, contract, backend.stein_sym_workspace_bytes(dev, n) if sym else 0)
        self._stein_key = self._key(x_dict)

    def _place_gram(self, build, contract, ws_bytes=0):
        """Builds K_p and, for large matrices, picks a well-placed copy.  The contraction streams the matrix from HBM
        and its rate depends on where the driver put it relative to the contraction's workspace: round 2, same kernel,
        same box, n = 16: 2.55 ms or 2.78 ms, stable for the life of the two allocations, the same alone and inside the
        training step, following the (K_p, workspace) PAIR -- a workspace inside K_p's own allocation is always the
        slow case (tools/probes/ws_in_kp_probe.py), one 64+ GiB further on usually the fast one (ws_far_probe.py).
        (Round 1's 2.84 / 3.32 ms were the same effect amplified by 8x more partial-sum stores.)
        So: build up to `gram_placement_tries` copies (each in fresh memory while the earlier ones are still held), and
        behind each a fresh workspace (`ws_bytes` > 0: the symmetric contraction's, which then lies one matrix further
        on than the last), time the contraction on every (copy, workspace) pair, keep the fastest pair, free the rest.
        Same matrix, same results; stops as soon as one pair streams at 83 % of the HBM peak (about every second first
        copy does: then nothing extra is built); at worst `gram_placement_tries` copies are held at once for ~0.2 s."""
        K = build()
        nbytes = K.numel() * K.element_size()
        tries = int(self.gram_placement_tries)
        self.gram_placement = None
        if tries <= 1 or nbytes < (1 << 30):
            return K
        dev = K.device
        free_b, _ = torch.cuda.mem_get_info(dev)
        q = torch.full((1 << self.num_latent_vars,), 1.0 / (1 << self.num_latent_vars), dtype=torch.float64, device=dev)

        def clock(Kc):
            contract(Kc, q)
            torch.cuda.synchronize(dev)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(3):
                contract(Kc, q)
            b.record()
            torch.cuda.synchronize(dev)
            return a.elapsed_time(b) / 3

        Ks = [K]
        Ws = [backend.fresh_workspace(dev, ws_bytes)] if ws_bytes else [None]
        took = {}

        def time_new_pairs():
            for i, Kc in enumerate(Ks):
                for j, w in enumerate(Ws):
                    if (i, j) not in took:
                        if w is not None:
                            backend.set_workspace(dev, "qfsym", w)
                        took[(i, j)] = clock(Kc)

        # good enough = the upper triangle (half of these rows) at 83 % of the MI355X's 8 TB/s: what the kernel reaches on
        # a well-placed pair (n = 16: 2.59 ms; fast pairs run 2.555-2.58, the rest 2.61-2.80).  A first copy that is
        # already there costs nothing extra -- no second copy is built.
        good_ms = (nbytes / 2) / (0.83 * 8e12) * 1e3
        time_new_pairs()
        while len(Ks) < tries and free_b > (len(Ks) + 1) * (nbytes + ws_bytes) + (8 << 30):
            if min(took.values()) <= good_ms:
                break
            Ks.append(build())
            if ws_bytes:
                Ws.append(backend.fresh_workspace(dev, ws_bytes))
            time_new_pairs()
        bi, bj = min(took, key=took.get)
        K = Ks[bi]
        if Ws[bj] is not None:
            backend.set_workspace(dev, "qfsym", Ws[bj])
        self.gram_placement = {"contraction_ms_per_pair": [[i, j, round(t, 4)] for (i, j), t in sorted(took.items())],
                               "kept": [bi, bj], "note": "[K_p copy, workspace, ms]"}
        del Ks, Ws
        torch.cuda.empty_cache()
        return K

    def choose_overlap(self, reps=4):
        """Decides by measurement whether the circuits and the contraction of a step take turns on the whole chip or
        run side by side on two halves of its CUs ("partition"); call after `_prepare_stein`.  Only where both are
        sizeable: one GPU, dense Gram, circuits of several passes.  Runs 2 + `reps` gradient evaluations per mode at the current theta (no optimiser
        step, nothing is kept); both modes produce the same numbers."""
        rank, ws = shard.world(self.process_group)
        n = self.num_latent_vars
        self.overlap_streams = False
        if ws != 1 or not self._use_dense() or n < 14:
            return
        dev = self._S.device
        theta64 = self.born_machine.theta.detach().to(device=dev, dtype=torch.float64).contiguous()
        timers, self.timers = self.timers, None
        took = {}
        try:
            for mode in (False, "partition"):
                self.overlap_streams = mode
                for _ in range(2):
                    self.ksd_and_grad(theta64)
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                for _ in range(reps):
                    self.ksd_and_grad(theta64)
                torch.cuda.synchronize(dev)
                took[mode] = (time.perf_counter() - t0) / reps * 1e3
        finally:
            self.timers = timers
        self.overlap_streams = "partition" if took["partition"] < 0.97 * took[False] else False
        self.overlap_choice = {"sequential_ms": round(took[False], 4), "partition_ms": round(took["partition"], 4),
                               "chosen": "partition" if self.overlap_streams else "sequential"}

    def _timed(self, name):
        return _EventSpan(self.timers, name)

    def _stein_contract(self, q):
        """(ksd2 [1], y = K_p q [2^n]) for the current q, on the GPU."""
        n = self.num_latent_vars
        if self._K is None:
            return backend.stein_matvec_kron(self._S, q, n, self.base_kernel_length_scale)
        rank, ws = shard.world(self.process_group)
        if self._K_pairs is not None:
            # every rank adds its share of (K q, q.y); one all-reduce of 2^n + 1 doubles
            pa, pb, nlo = self._K_pairs
            msg = backend.stein_quadform_sym_pairs(self._K[:nlo], self._K[nlo:], pa, pb, q, n)
            with self._timed("allreduce"):
                shard.all_reduce_sum(msg, self.process_group)
            return msg[1 << n:], msg[: 1 << n]
        r0, r1 = self._K_rows
        if ws == 1:
            if self.symmetric_contraction:      # K_p from our builder is bitwise symmetric: read half of it
                return backend.stein_quadform_sym(self._K, q, n)
            ksd2, Y = backend.stein_quadform(self._K, q, n, want_y=True)
            return ksd2, Y[0]
        # row shard: every rank contributes its rows of K q plus its partial of q.y in one all-gather
        chunk = -(-(1 << n) // ws)
        msg = torch.zeros(chunk + 1, dtype=torch.float64, device=q.device)
        part = backend.stein_quadform_rows(self._K, r0, r1, q, n)
        msg[: r1 - r0] = part[:-1]
        msg[chunk] = part[-1]
        full = torch.empty((ws, chunk + 1), dtype=torch.float64, device=q.device)
        with self._timed("allreduce"):          # (an all-gather here: the row shard's exchange of K q rows)
            shard.all_gather_flat(full.view(-1), msg, self.process_group)
        y = full[:, :chunk].reshape(-1)[: 1 << n].contiguous()
        ksd2 = full[:, chunk].sum().reshape(1)      # fixed rank order: identical on every rank
        return ksd2, y

    # ---- one KSD-gradient step on the device -------------------------------------------------------------
    def ksd_and_grad(self, theta64=None):
        """Runs the device part of one epoch for the current theta: returns (loss [1] float64 on the GPU,
        grad [P] float64 on the GPU, q [2^n]).  Requires `_prepare_stein` (train() calls it)."""
        bm = self.born_machine
        n, L, at = self.num_latent_vars, bm.ansatz_layers, bm.ansatz_type
        dev = self._S.device
        if theta64 is None:
            theta64 = bm.theta.detach().to(device=dev, dtype=torch.float64).contiguous()
        P = theta64.numel()
        rank, ws = shard.world(self.process_group)
        lo, hi, step = shard.shard_params(P, rank, ws)
        n_local = len(range(lo, hi, step))
        if self.grad_engine == "adjoint":
            with self._timed("circuits"):
                state, q = backend.adjoint_state(at, n, L, theta64)
            with self._timed("stein"):
                ksd2, y = self._stein_contract(q)
            with self._timed("finish"):
                loss, _, dldq = backend.ksd_grad_finish(n, None, 0, y, ksd2, want_dldq=True)
                grad = backend.adjoint_vjp(at, n, L, theta64, state, dldq)
            return loss, grad, q
        if self.grad_engine != "paramshift":
            raise ValueError("grad_engine must be 'paramshift' or 'adjoint'")
        overlap = self.overlap_streams
        if overlap is None:
            overlap = False
        if overlap == "partition":
            main = torch.cuda.current_stream(dev)
            ncu = torch.cuda.get_device_properties(dev).multi_processor_count
            nc = ncu // 2
            sc, sa = backend.cu_range_stream(dev, 0, nc), backend.cu_range_stream(dev, nc, ncu - nc)
            start = torch.cuda.Event()
            start.record(main)                      # theta64 is produced on the main stream
            try:
                with torch.cuda.stream(sa):
                    sa.wait_event(start)
                    backend.set_engine_option(dev, "circuit_cus", ncu - nc)
                    with self._timed("base_circuit"):
                        q = backend.paramshift_probs(at, n, L, theta64, 0, 0, include_base=True, ws_tag="base")[0]
                    with self._timed("stein"):
                        ksd2, y = self._stein_contract(q)
                    stein_done = torch.cuda.Event()
                    stein_done.record(sa)
                with torch.cuda.stream(sc):
                    sc.wait_event(start)
                    backend.set_engine_option(dev, "circuit_cus", nc)
                    with self._timed("circuits"):
                        shifted = backend.paramshift_probs(at, n, L, theta64, lo, hi, include_base=False, p_stride=step,
                                                           ws_tag="part")
                    circ_done = torch.cuda.Event()
                    circ_done.record(sc)
            finally:
                backend.set_engine_option(dev, "circuit_cus", 0)
            main.wait_event(stein_done)
            main.wait_event(circ_done)
            for tns in (ksd2, y, q, shifted):
                tns.record_stream(main)
            theta64.record_stream(sa)
            theta64.record_stream(sc)
        elif not overlap and self.fused_dot and backend.paramshift_dot_supported(at, n, L, dev, n_local):
            # The dot product with dL/dq fused into the shifted circuits' last pass (kernels_circuit8.hip): base circuit
            # and all but the last pass of the shifted ones -> q -> contraction -> last pass of the shifted circuits with
            # w = y.  Their probabilities are never written or re-read (8 GB each way at n = 20).
            with self._timed("circuits"):
                q, token = backend.paramshift_dot_begin(at, n, L, theta64, lo, hi, p_stride=step)
            with self._timed("stein"):
                ksd2, y = self._stein_contract(q)
            with self._timed("finish"):
                loss, grad_local = backend.paramshift_dot_finish(token, y, ksd2)
                with self._timed("allgather"):
                    grad = shard.all_gather_grad(grad_local, P, self.process_group)
            return loss, grad, q
        elif not overlap:
            with self._timed("circuits"):
                probs = backend.paramshift_probs(at, n, L, theta64, lo, hi, include_base=True, p_stride=step)
            q = probs[0]
            shifted = probs[1:]
            with self._timed("stein"):
                ksd2, y = self._stein_contract(q)
        else:
            # The contraction needs only q of the base circuit, the shifted circuits need neither: the base
            # circuit and then the contraction run on a second HIP stream while the 2P shifted circuits run on
            # the main one (they are bound by different resources: HBM vs LDS/FMA + HBM).
            main = torch.cuda.current_stream(dev)
            if self._aux_stream is None:
                self._aux_stream = torch.cuda.Stream(device=dev)
            aux = self._aux_stream
            start = torch.cuda.Event()
            start.record(main)                      # theta64 is produced on the main stream
            with torch.cuda.stream(aux):
                aux.wait_event(start)
                with self._timed("base_circuit"):
                    q = backend.paramshift_probs(at, n, L, theta64, 0, 0, include_base=True, ws_tag="base")[0]
                with self._timed("stein"):
                    ksd2, y = self._stein_contract(q)
                stein_done = torch.cuda.Event()
                stein_done.record(aux)
            with self._timed("circuits"):
                shifted = backend.paramshift_probs(at, n, L, theta64, lo, hi, include_base=False, p_stride=step)
            main.wait_event(stein_done)
            for tns in (ksd2, y, q):
                tns.record_stream(main)
            theta64.record_stream(aux)
        with self._timed("finish"):
            loss, grad_local, _ = backend.ksd_grad_finish(n, shifted, n_local, y, ksd2)
            with self._timed("allgather"):
                grad = shard.all_gather_grad(grad_local, P, self.process_group)
        return loss, grad, q

    def make_optimizer(self, lr_born_machine, num_epochs, use_lr_scheduler=True, optimizer_type="adam",
                       adam_betas=(0.9, 0.999), capturable=False):
        """Optimiser and scheduler exactly as the reference builds them (ksd_vi_quantum.py:92-103).
        capturable=True (Adam on the GPU only): the learning rate lives in a device tensor and the step counter on the
        device, so that the whole step can be replayed from a HIP graph (`make_graphed_step`); same update rule."""
        params = list(self.born_machine.parameters())
        # same optimisers and hyper-parameters as the reference; when theta lives on the GPU the single-kernel
        # ("fused") implementation of the same torch.optim class is selected: identical update rule, ~0.1 ms
        # less host time per step
        fused = {"fused": True} if all(p.is_cuda for p in params) else {}
        if capturable:
            if optimizer_type != "adam" or not fused:
                raise backend.BornviError("a graph-capturable step needs Adam with theta on the GPU")
            fused["capturable"] = True
            lr_born_machine_arg = torch.tensor(float(lr_born_machine), dtype=torch.float32, device=params[0].device)
        else:
            lr_born_machine_arg = lr_born_machine
        if optimizer_type == "adam":
            optimizer_born = optim.Adam(params, lr=lr_born_machine_arg, betas=adam_betas, **fused)
        elif optimizer_type == "sgd":
            optimizer_born = optim.SGD(params, lr=lr_born_machine, momentum=0.9, **fused)
        else:
            optimizer_born = optim.Adam(params, lr=lr_born_machine, **fused)
        scheduler = None
        if use_lr_scheduler:
            scheduler = optim.lr_scheduler.CosineAnnealingLR(optimizer_born, T_max=num_epochs,
                                                             eta_min=lr_born_machine / 10)
        return params, optimizer_born, scheduler

    def training_step_async(self, params, optimizer_born, scheduler, gradient_clip_norm):
        """The same epoch body with NO host synchronisation: returns (loss [1] float64 on the GPU, grad norm 0-dim
        float32 on the GPU, q); the caller reads the values when it needs them (e.g. after K steps), so the GPU runs
        the steps back to back instead of idling while the host handles `loss.item()`.  The NaN/Inf guard of the
        reference (:147-148, "Skipping update") runs on the device: the fused optimiser kernel skips the update
        when `found_inf` is set (the torch.amp.GradScaler mechanism of torch.optim).  One deviation: the LR
        scheduler also advances on such a step (the host cannot know).  Needs theta and a fused optimiser on the GPU."""
        theta = self.born_machine.theta
        if not (theta.is_cuda and theta.dtype == torch.float32 and len(params) == 1 and optimizer_born.defaults.get("fused")):
            raise backend.BornviError("training_step_async needs a float32 theta on the GPU and a fused torch optimiser")
        optimizer_born.zero_grad()
        loss_t, grad64, q = self.ksd_and_grad()
        g32, grad_norm, found_inf = backend.clip_cast_grad_guard(grad64, gradient_clip_norm, loss_t)
        theta.grad = g32
        optimizer_born.found_inf = found_inf
        try:
            optimizer_born.step()
        finally:
            del optimizer_born.found_inf
        if scheduler is not None:
            scheduler.step()
        return loss_t, grad_norm, q

    @staticmethod
    def _device_adam_for(theta, optimizer_born, scheduler):
        """A DeviceAdam equal to (optimizer_born, scheduler) if they are what make_optimizer builds for "adam" and nothing
        has stepped yet; None otherwise (the torch objects run the update then)."""
        if type(optimizer_born) is not optim.Adam or len(optimizer_born.param_groups) != 1 or len(optimizer_born.state) != 0:
            return None
        g = optimizer_born.param_groups[0]
        if g.get("weight_decay", 0) != 0 or g.get("amsgrad") or g.get("maximize") or g.get("differentiable"):
            return None
        T_max, eta_min, lr = None, 0.0, float(g["lr"])
        if scheduler is not None:
            if type(scheduler) is not optim.lr_scheduler.CosineAnnealingLR or scheduler.last_epoch != 0:
                return None
            T_max, eta_min, lr = scheduler.T_max, scheduler.eta_min, float(scheduler.base_lrs[0])
        return DeviceAdam(theta, lr, g["betas"], g["eps"], T_max, eta_min)

    def make_graphed_step(self, params, optimizer_born, scheduler, gradient_clip_norm, warmup=3, record=None,
                          device_adam=True):
        """The epoch body of `training_step_async` captured ONCE into a HIP graph (torch.cuda.CUDAGraph: our kernels
        are launched on torch's current stream, so the capture records them together with the cast, the fused Adam
        kernel and the guard) and replayed per step: one graph launch instead of ~15 kernel launches and their host
        work.  For the latency-bound sizes (n <= 13: BASELINE config 2 spends its step in launch overhead, SURVEY
        section 7.4).  Returns step() -> (loss [1], grad_norm, q): tensors OWNED BY THE GRAPH, overwritten by the next
        step (clone what must be kept).  Needs `make_optimizer(..., capturable=True)`; `warmup` eager steps run first
        (they are real optimiser steps).  device_adam (default): clip, guard, Adam, schedule and the float64 cast of
        theta are one launch of ours (`DeviceAdam`; 6 graph nodes at n = 8 instead of 10 plus 6 eager launches per
        step) -- equal to torch's update to rounding; the torch optimiser and scheduler objects are then left untouched,
        `step.adam` is the state and `step.last_lr()` the schedule's rate.  Otherwise (or when the optimiser is not the
        plain Adam + cosine pair of make_optimizer) torch's capturable fused Adam is captured and the scheduler advances
        on the host after each replay (it fills the learning-rate tensor the captured Adam kernel reads)."""
        theta = self.born_machine.theta
        if not (theta.is_cuda and theta.dtype == torch.float32 and len(params) == 1 and optimizer_born.defaults.get("capturable")):
            raise backend.BornviError("make_graphed_step needs a float32 theta on the GPU and make_optimizer(capturable=True)")
        if self.timers is not None:
            raise backend.BornviError("event timers cannot be recorded inside a graph capture: set timers = None")
        dev = theta.device
        if theta.grad is None:
            theta.grad = torch.zeros_like(theta)
        found = torch.zeros((), dtype=torch.float32, device=dev)
        adam = self._device_adam_for(theta, optimizer_born, scheduler) if device_adam else None

        def own_body():
            loss_t, grad64, q = self.ksd_and_grad(theta64=adam.theta64)
            return loss_t, adam.step(grad64, loss_t, gradient_clip_norm), q

        def torch_body():
            loss_t, grad64, q = self.ksd_and_grad()
            # (the clipped gradient and the guard flag are written straight into theta.grad and the flag tensor the fused
            # Adam kernel reads: no copy nodes in the graph)
            _, grad_norm, _ = backend.clip_cast_grad_guard(grad64, gradient_clip_norm, loss_t,
                                                           out=theta.grad.view(grad64.shape), found_out=found)
            optimizer_born.found_inf = found
            try:
                optimizer_born.step()
            finally:
                del optimizer_born.found_inf
            return loss_t, grad_norm, q

        body = own_body if adam is not None else torch_body

        def host_advance():
            if adam is not None:
                adam.advance()
            elif scheduler is not None:
                scheduler.step()

        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(max(1, int(warmup))):      # plans, workspaces and optimiser state exist before the capture
                w = body()                            # (real optimiser steps: `record`, a list, receives their outputs)
                if record is not None:
                    record.append(tuple(t.clone() for t in w))
                host_advance()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            out = body()

        def step():
            graph.replay()
            host_advance()
            return out

        step.graph = graph
        step.adam = adam
        step.last_lr = (adam.last_lr if adam is not None else
                        (lambda: float(scheduler.get_last_lr()[0])) if scheduler is not None else None)
        step.found_inf = found        # the captured kernels write and read this tensor on every replay: it lives as long
        return step                   # as step() does (freed, its block is handed to the caller's next small tensor)

    def training_step(self, params, optimizer_born, scheduler, gradient_clip_norm):
        """One epoch body (reference :111-161) without the logging: device step, NaN/Inf guard, clip,
        optimiser and scheduler step.  Returns (loss_value, grad_norm or None if skipped, q)."""
        optimizer_born.zero_grad()
        loss_t, grad64, q = self.ksd_and_grad()
        loss_value = float(loss_t.item())        # the epoch's one host sync (reference: loss.item(), :163)
        if np.isnan(loss_value) or np.isinf(loss_value):
            return loss_value, None, q
        theta = self.born_machine.theta
        if theta.dtype == torch.float32 and len(params) == 1:
            # cast + clip_grad_norm_ (reference :153) in one device launch; same formula as torch's
            g32, grad_norm = backend.clip_cast_grad(grad64, gradient_clip_norm)
            theta.grad = g32.to(theta.device)
        else:
            theta.grad = grad64.to(device=theta.device, dtype=theta.dtype)
            grad_norm = nn_utils.clip_grad_norm_(params, gradient_clip_norm)
        optimizer_born.step()
        if scheduler is not None:
            scheduler.step()
        return loss_value, grad_norm, q

    def train(self, x_observation_dict, num_epochs, lr_born_machine,
              verbose=True, true_posterior_for_tvd=None,
              use_lr_scheduler=True, gradient_clip_norm=10.0,
              optimizer_type="adam", adam_betas=(0.9, 0.999), *, host_sync=True):
        """Same signature, history keys and messages as the reference (ksd_vi_quantum.py:76-190).
        host_sync (keyword-only extra): True = the reference's epoch, which reads `loss.item()` every epoch.  False =
        the same epochs without host synchronisation (theta on the GPU): `training_step_async` per epoch -- for
        n <= 13 with Adam replayed from ONE HIP graph (`make_graphed_step`) -- with the NaN/Inf guard on the device; losses,
        norms and TVDs stay on the device and are read where the reference prints and once at the end, so a "Skipping
        update" warning appears at the next such point.  Same history."""
        if not host_sync:
            return self._train_deferred(x_observation_dict, num_epochs, lr_born_machine, verbose, true_posterior_for_tvd,
                                        use_lr_scheduler, gradient_clip_norm, optimizer_type, adam_betas)

        if self.num_observed_vars > 0 and set(x_observation_dict.keys()) != set(self.observed_vars_names):
            raise ValueError("Keys in x_observation_dict must match self.observed_vars_names.")

        qbm_x_condition_input = None
        if self.num_observed_vars > 0 and self.born_machine.conditioning_dim > 0:
            x_obs_list_for_qbm = [x_observation_dict[name] for name in self.observed_vars_names]
            qbm_x_condition_input = torch.tensor(x_obs_list_for_qbm, dtype=torch.float32, device=self.pytorch_device)

        self._precompute_all_s_p(x_observation_dict)

        params, optimizer_born, scheduler = self.make_optimizer(lr_born_machine, num_epochs, use_lr_scheduler,
                                                                optimizer_type, adam_betas)

        history = {'loss_ksd': [], 'tvd': [], 'grad_norm': []}
        best_tvd = float('inf')
        best_params = None
        grad_norm = None
        log_every = (num_epochs // 10 if num_epochs >= 10 else 1)

        for epoch in range(num_epochs):
            if self.born_machine.conditioning_dim > 0 and qbm_x_condition_input is not None:
                print("Warning: Conditioning with x_condition not fully implemented in PQC ansatz yet.")
            loss_value, step_norm, q = self.training_step(params, optimizer_born, scheduler, gradient_clip_norm)

            if verbose and epoch % log_every == 0:
                print(f"  Epoch {epoch+1} Q Probs (first 4): {q[:4].detach().cpu().numpy()}")

            if q.shape[0] != self.num_possible_latent_states:
                raise ValueError(f"Probabilities from Born machine have unexpected shape")

            if step_norm is None:
                print(f"Warning: NaN or Inf KSD loss: {loss_value}. Skipping update.")
            else:
                grad_norm = step_norm
                if verbose and epoch % log_every == 0:
                    print(f"  Epoch {epoch+1} Grad Norm (after clipping): {grad_norm:.4f}")

            history['loss_ksd'].append(loss_value)
            history['grad_norm'].append(grad_norm if grad_norm is not None else 0.0)

            if true_posterior_for_tvd is not None:
                if torch.is_tensor(true_posterior_for_tvd):
                    # array form (stein_utils.true_posterior_table): no dict of 2^n tuples; like the reference the
                    # distribution AFTER this epoch's update is compared (one more circuit, :168)
                    q_now = self.born_machine.get_probabilities(x_condition=qbm_x_condition_input).detach().squeeze()
                    tvd = float(tvd_table(true_posterior_for_tvd.to(q_now.device), q_now))
                else:
                    current_q_dist_dict = self.born_machine.get_prob_dict(x_condition=qbm_x_condition_input)
                    tvd = calculate_tvd(true_posterior_for_tvd, current_q_dist_dict)
                history['tvd'].append(tvd)
                if tvd < best_tvd:
                    best_tvd = tvd
                    best_params = self.born_machine.state_dict()     # aliases the live tensors (quirk Q3)
            else:
                history['tvd'].append(np.nan)

            if verbose and (epoch % max(1, num_epochs // 20) == 0 or epoch == num_epochs - 1):
                log_msg = f"Epoch {epoch+1}/{num_epochs} | KSD: {loss_value:.6f}"
                if scheduler is not None:
                    log_msg += f" | LR: {scheduler.get_last_lr()[0]:.6f}"
                if true_posterior_for_tvd is not None and len(true_posterior_for_tvd) and not np.isnan(history['tvd'][-1]):
                    log_msg += f" | TVD: {history['tvd'][-1]:.6f}"
                print(log_msg)

        if best_params is not None and verbose:
            print(f"\nRestoring best parameters (TVD: {best_tvd:.6f})")
            self.born_machine.load_state_dict(best_params)

        return history

    def _train_deferred(self, x_observation_dict, num_epochs, lr_born_machine, verbose, true_posterior_for_tvd,
                        use_lr_scheduler, gradient_clip_norm, optimizer_type, adam_betas):
        """train(host_sync=False): see there.  The epochs are `training_step_async` (or its HIP-graph replay)."""
        if self.num_observed_vars > 0 and set(x_observation_dict.keys()) != set(self.observed_vars_names):
            raise ValueError("Keys in x_observation_dict must match self.observed_vars_names.")
        theta = self.born_machine.theta
        if not (theta.is_cuda and theta.dtype == torch.float32):
            raise backend.BornviError("train(host_sync=False) needs a float32 theta on the GPU (pytorch_device='cuda:N')")
        self._precompute_all_s_p(x_observation_dict)
        n = self.num_latent_vars
        rank, ws = shard.world(self.process_group)
        use_graph = (optimizer_type == "adam" and n <= 13 and ws == 1 and self.timers is None and num_epochs > 4
                     and not self.overlap_streams and true_posterior_for_tvd is None)   # (a TVD per epoch needs theta
                                                                                          # after exactly that epoch)
        params, optimizer_born, scheduler = self.make_optimizer(lr_born_machine, num_epochs, use_lr_scheduler,
                                                                optimizer_type, adam_betas, capturable=use_graph)
        dev = theta.device
        losses, norms, tvds, first4 = [], [], [], {}
        log_every = (num_epochs // 10 if num_epochs >= 10 else 1)
        tvd_table_dev = None
        if true_posterior_for_tvd is not None:
            tvd_table_dev = (true_posterior_for_tvd if torch.is_tensor(true_posterior_for_tvd)
                             else torch.tensor([true_posterior_for_tvd.get(o, 0.0) for o in self.all_latent_states_tuples],
                                               dtype=torch.float64)).to(self._S.device)
        step = None
        seen = 0                              # epochs whose warnings / values have been reported

        adam = None                           # the graphed step's DeviceAdam: the kernel keeps the history, no clones

        def report(upto):
            nonlocal seen
            if upto <= seen:
                return None
            if adam is not None:
                vals = adam.history(seen, upto)[0].cpu().tolist()
            else:
                vals = torch.stack([l.reshape(()) for l in losses[seen:upto]]).cpu().tolist()
            for v in vals:
                if np.isnan(v) or np.isinf(v):
                    print(f"Warning: NaN or Inf KSD loss: {v}. Skipping update.")
            seen = upto
            return vals[-1]

        for epoch in range(num_epochs):
            if use_graph and epoch == 0:
                rec = []
                step = self.make_graphed_step(params, optimizer_born, scheduler, gradient_clip_norm, warmup=2, record=rec)
                pending = rec                 # epochs 0 and 1 are the graph's two eager warm-up steps
                adam = step.adam
            if use_graph and epoch < 2:
                loss_t, gn, q = pending[epoch]
            elif use_graph and adam is not None:
                loss_t, gn, q = step()        # graph-owned: read below, before the next replay, or not at all
            elif use_graph:
                loss_t, gn, q = (t.clone() for t in step())
            else:
                loss_t, gn, q = self.training_step_async(params, optimizer_born, scheduler, gradient_clip_norm)
            if q.shape[0] != self.num_possible_latent_states:
                raise ValueError(f"Probabilities from Born machine have unexpected shape")
            if adam is None:
                losses.append(loss_t)
                norms.append(gn)
            if tvd_table_dev is not None:     # like the reference: the distribution AFTER this epoch's update (:168)
                q_now = self.born_machine.get_probabilities().detach().squeeze()
                tvds.append(tvd_table(tvd_table_dev.to(q_now.device), q_now))
            if verbose and epoch % log_every == 0:
                print(f"  Epoch {epoch+1} Q Probs (first 4): {q[:4].detach().cpu().numpy()}")
                report(epoch + 1)
                print(f"  Epoch {epoch+1} Grad Norm (after clipping): {float(gn):.4f}")
            if verbose and (epoch % max(1, num_epochs // 20) == 0 or epoch == num_epochs - 1):
                last = report(epoch + 1)
                last = float(loss_t) if last is None else last
                log_msg = f"Epoch {epoch+1}/{num_epochs} | KSD: {last:.6f}"
                if scheduler is not None:
                    lr_now = (step.adam.lr_at(epoch + 1) if step is not None and step.adam is not None
                              else float(scheduler.get_last_lr()[0]))
                    log_msg += f" | LR: {lr_now:.6f}"
                if tvds:
                    log_msg += f" | TVD: {float(tvds[-1]):.6f}"
                print(log_msg)
        report(num_epochs)
        if adam is not None:
            hl, hn = adam.history(0, num_epochs)
            loss_h, norm_h = hl.cpu().tolist(), hn.to(torch.float64).cpu().tolist()
        else:
            loss_h = torch.stack([l.reshape(()) for l in losses]).cpu().tolist() if losses else []
            norm_h = torch.stack([g.reshape(()).to(torch.float64) for g in norms]).cpu().tolist() if norms else []
        # the reference keeps the last good norm on a skipped epoch (0.0 before the first good one)
        grad_h, last_good = [], None
        for lv, gv in zip(loss_h, norm_h):
            if not (np.isnan(lv) or np.isinf(lv)):
                last_good = gv
            grad_h.append(last_good if last_good is not None else 0.0)
        tvd_h = (torch.stack([t.reshape(()) for t in tvds]).cpu().tolist() if tvds else [np.nan] * num_epochs)
        history = {'loss_ksd': loss_h, 'tvd': tvd_h, 'grad_norm': grad_h}
        if tvds and verbose:
            print(f"\nRestoring best parameters (TVD: {min(tvd_h):.6f})")     # (a no-op in the reference too: quirk Q3)
        return history
