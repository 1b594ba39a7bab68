"""QuantumBornMachine on the MI355X statevector backend.

Drop-in for the reference class of the same name (quantum_born_machine.py:7-201): same constructor
arguments, attributes (`theta`, `num_ansatz_params`, `num_latent_vars`, `conditioning_dim`,
`ansatz_type`, `ansatz_layers`, `all_outcomes_tuples`, `pqc`, `dev`) and methods
(`get_probabilities`, `get_prob_dict`, `sample`, `get_log_q_z_x`), with the PennyLane
`default.qubit` QNode replaced by the hand-written HIP circuit engine behind the C ABI
(`bornvi_circuit_probs`, `bornvi_paramshift_probs`).

Where the circuit runs: always on an MI355X (the tensor's own cuda device if `theta` lives on one,
otherwise the process's current cuda device) -- just as the reference always simulated on the host
with PennyLane whatever `pytorch_device` was.  Outputs are returned on `theta.device`.
There is no CPU fallback: without a GPU the calls raise.

Numerics: theta is float32 (reference :42-47) and is upcast exactly to float64; the simulation is
complex128; probabilities are returned as float64 (quirk Q9 in SURVEY.md).  The gradient is the
two-term parameter-shift rule with shift pi/2 (what `diff_method="parameter-shift"` does at
:58/:90/:114), evaluated as 2P extra circuits, optionally sharded over the ranks of a process group.
"""
import numpy as np
import torch
import torch.nn as nn

from . import backend
from . import paramshift_shard as shard
from .utils import generate_all_binary_outcomes


class _BornviDevice:
    """Stand-in for the `qml.device(...)` object kept in `self.dev` (reference :28)."""

    def __init__(self, name, wires):
        self.name = name
        self.short_name = name
        self.wires = tuple(range(wires))
        self.num_wires = wires
        self.shots = None
        self.backend = "bornvi-hip-gfx950"

    def __repr__(self):
        return f"<bornvi statevector device standing in for '{self.name}', wires={self.num_wires}>"


class _CircuitProbs(torch.autograd.Function):
    """q_theta with a parameter-shift backward (2P circuit evaluations on the GPU)."""

    @staticmethod
    def forward(ctx, weights, machine):
        dev = backend.compute_device(weights.device)
        th64 = weights.detach().to(device=dev, dtype=torch.float64).contiguous()
        probs = backend.circuit_probs(machine.ansatz_type, machine.num_latent_vars, machine.ansatz_layers,
                                      th64.view(1, -1))[0]
        ctx.machine = machine
        ctx.save_for_backward(th64)
        ctx.out_device = weights.device
        ctx.in_dtype = weights.dtype
        return probs.to(weights.device)

    @staticmethod
    def backward(ctx, grad_out):
        (th64,) = ctx.saved_tensors
        m = ctx.machine
        dev = th64.device
        dLdq = grad_out.detach().to(device=dev, dtype=torch.float64).contiguous()
        P = th64.numel()
        rank, ws = shard.world(m.process_group)
        lo, hi, step = shard.shard_params(P, rank, ws)
        local = backend.paramshift_grad(m.ansatz_type, m.num_latent_vars, m.ansatz_layers, th64, dLdq, lo, hi, step)
        full = shard.all_gather_grad(local, P, m.process_group)
        return full.to(device=ctx.out_device, dtype=ctx.in_dtype), None


class QuantumBornMachine(nn.Module):
    def __init__(self, num_latent_vars, ansatz_layers=1, conditioning_dim=0,
                 device_name="default.qubit", ansatz_type="hardware_efficient",
                 init_method="small_random"):
        """
        Args (reference quantum_born_machine.py:8-21):
            num_latent_vars (int): number of qubits
            ansatz_layers (int): layers of the parameterised circuit
            conditioning_dim (int): dimension of the conditioning variable x (ignored by the circuit, as in the reference)
            device_name (str): PennyLane device name; accepted for compatibility, the circuit always runs
                on the HIP statevector engine
            ansatz_type (str): "hardware_efficient", "all_to_all", anything else = "basic"
            init_method (str): "zero", "small_random", anything else = uniform [0, 2 pi)
        """
        super().__init__()
        self.num_latent_vars = num_latent_vars
        self.conditioning_dim = conditioning_dim
        self.ansatz_type = ansatz_type
        self.ansatz_layers = ansatz_layers
        self.process_group = None      # set by the trainer to shard the parameter-shift circuits

        self.dev = _BornviDevice(device_name, num_latent_vars)

        if ansatz_type in ("hardware_efficient", "all_to_all"):
            self.num_ansatz_params = ansatz_layers * 3 * num_latent_vars       # reference :31-36
        else:
            self.num_ansatz_params = ansatz_layers * 2 * num_latent_vars       # reference :37-38

        if init_method == "zero":
            init = torch.zeros(self.num_ansatz_params, dtype=torch.float32)
        elif init_method == "small_random":
            init = 0.1 * torch.randn(self.num_ansatz_params, dtype=torch.float32)
        else:
            init = torch.rand(self.num_ansatz_params, dtype=torch.float32) * 2 * torch.pi
        self.theta = nn.Parameter(init)

        self._outcomes = None
        if num_latent_vars == 0:
            self._outcomes = [()]

        def pqc(weights, x_inputs=None):
            return _CircuitProbs.apply(weights, self)
        self.pqc = pqc

    @property
    def all_outcomes_tuples(self):
        """generate_all_binary_outcomes(n) (reference :49-54), built on first use: at n = 20 the list is
        2^20 Python tuples, which the training hot path never needs."""
        if self._outcomes is None:
            self._outcomes = generate_all_binary_outcomes(self.num_latent_vars)
            if not self._outcomes and self.num_latent_vars > 0:
                raise ValueError("Failed to generate outcome tuples.")
        return self._outcomes

    def get_probabilities(self, x_condition=None):
        """q_theta(z|x) over all 2^n states (float64), differentiable w.r.t. theta (reference :132-137)."""
        if self.conditioning_dim > 0 and x_condition is not None:
            print("Warning: Conditioning with x_condition not fully implemented in PQC ansatz yet.")
        return self.pqc(weights=self.theta)

    def get_prob_dict(self, x_condition=None):
        """{outcome tuple: probability} (reference :139-151)."""
        probs_tensor = self.get_probabilities(x_condition=x_condition)
        if probs_tensor.shape[0] != len(self.all_outcomes_tuples):
            raise ValueError(f"Mismatch between probability tensor shape and number of outcomes")
        return dict(zip(self.all_outcomes_tuples, probs_tensor.detach().cpu().tolist()))

    def sample(self, num_samples_to_draw, x_condition=None):
        """float32 [num, n] bit rows drawn from q_theta (reference :153-178)."""
        if self.num_latent_vars == 0:
            return torch.empty(num_samples_to_draw, 0, dtype=torch.float32, device=self.theta.device)
        probs = self.get_probabilities(x_condition=x_condition).detach()
        if probs.ndim > 1:
            if probs.shape[0] == 1 and probs.ndim == 2:
                probs = probs.squeeze(0)
            else:
                raise ValueError(f"Probabilities for sampling should be 1D, but got shape {probs.shape}")
        probs = probs / torch.sum(probs)
        if num_samples_to_draw == 0:
            return torch.empty(0, self.num_latent_vars, dtype=torch.float32, device=self.theta.device)
        idx = torch.multinomial(probs, num_samples_to_draw, replacement=True)
        n = self.num_latent_vars
        shifts = torch.arange(n - 1, -1, -1, device=idx.device)
        return ((idx.unsqueeze(1) >> shifts) & 1).to(torch.float32).to(self.theta.device)

    def get_log_q_z_x(self, z_samples_batch, x_condition=None):
        """log q_theta(z|x) for a batch of bit rows (reference :180-201); ValueError on a non-binary row."""
        if self.num_latent_vars == 0:
            if z_samples_batch.shape[0] == 0:
                return torch.empty(0, device=self.theta.device)
            return torch.zeros(z_samples_batch.shape[0], device=self.theta.device)
        probs_all_states = self.get_probabilities(x_condition=x_condition)
        log_probs_all_states = torch.log(probs_all_states.clamp(min=1e-9))
        if z_samples_batch.shape[0] == 0:
            return torch.empty(0, device=self.theta.device)
        z = z_samples_batch.detach().to(log_probs_all_states.device).long()      # `.long()` truncates like the reference
        bad = ((z != 0) & (z != 1)).any(dim=1)
        if z.shape[1] != self.num_latent_vars or bool(bad.any()):
            row = int(torch.nonzero(bad)[0]) if bool(bad.any()) else 0
            raise ValueError(f"Sample {tuple(z[row].tolist())} is not a valid outcome")
        n = self.num_latent_vars
        weights = (1 << torch.arange(n - 1, -1, -1, device=z.device))
        return log_probs_all_states[(z * weights).sum(dim=1)]
