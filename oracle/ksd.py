"""NumPy/torch-CPU restatement of the KSD objective and one training epoch (test infrastructure only).

Follows ksd_vi_quantum.py:110-161.  The reference rebuilds every k_p(z_i, z_j)
inside the epoch (ksd_vi_quantum.py:125-142) although K_p does not depend on
theta; the restatement takes the precomputed Gram matrix (same numbers).
"""
import math
import numpy as np
import torch

from . import circuit as oc


def ksd_squared(K, q):
    """sum_ij q_i q_j k_p(z_i, z_j)  (ksd_vi_quantum.py:123-142)."""
    return float(q @ (K @ q))


def ksd_loss(K, q):
    """sqrt(clamp(sum, min=1e-12))  (ksd_vi_quantum.py:144-145)."""
    return math.sqrt(max(ksd_squared(K, q), 1e-12))


def ksd_grad_q(K, q):
    """d loss / d q as torch autograd produces it through the reference's double loop:
    d/dq_m sum_ij q_i q_j K_ij = sum_j K_mj q_j + sum_i q_i K_im ; divided by 2*loss;
    zero where the clamp is active (sum < 1e-12)."""
    s = ksd_squared(K, q)
    if s < 1e-12:
        return np.zeros_like(q)
    return (K @ q + K.T @ q) / (2.0 * math.sqrt(s))


class EpochTrace:
    """Deterministic restatement of `KSDVariationalInference.train` (ksd_vi_quantum.py:77-191)
    on the oracle circuit: returns per-epoch loss, pre-clip grad norm and theta."""

    def __init__(self, ansatz_type, n, layers, K, theta0, lr, num_epochs, clip=10.0,
                 use_lr_scheduler=True, optimizer_type="adam", adam_betas=(0.9, 0.999)):
        self.ansatz_type, self.n, self.layers, self.K = ansatz_type, n, layers, K
        self.theta = torch.nn.Parameter(torch.as_tensor(theta0, dtype=torch.float32).clone())
        if optimizer_type == "adam":
            self.opt = torch.optim.Adam([self.theta], lr=lr, betas=adam_betas)      # :94
        elif optimizer_type == "sgd":
            self.opt = torch.optim.SGD([self.theta], lr=lr, momentum=0.9)           # :96
        else:
            self.opt = torch.optim.Adam([self.theta], lr=lr)                        # :98
        self.sched = (torch.optim.lr_scheduler.CosineAnnealingLR(
            self.opt, T_max=num_epochs, eta_min=lr / 10) if use_lr_scheduler else None)   # :103
        self.clip = clip
        self.history = {"loss_ksd": [], "grad_norm": [], "theta": []}

    def step(self):
        self.opt.zero_grad()
        th = self.theta.detach().to(torch.float64).numpy()     # float32 -> exact float64 upcast
        q = oc.probs(self.ansatz_type, self.n, self.layers, th)
        loss = ksd_loss(self.K, q)
        if math.isnan(loss) or math.isinf(loss):
            self.history["loss_ksd"].append(loss)
            self.history["grad_norm"].append(0.0)
            return loss
        dLdq = ksd_grad_q(self.K, q)
        g = oc.paramshift_vjp(self.ansatz_type, self.n, self.layers, th, dLdq)
        self.theta.grad = torch.as_tensor(g, dtype=torch.float32)
        gn = torch.nn.utils.clip_grad_norm_([self.theta], self.clip)                # :153
        self.opt.step()                                                              # :158
        if self.sched is not None:
            self.sched.step()                                                        # :160-161
        self.history["loss_ksd"].append(loss)
        self.history["grad_norm"].append(float(gn))
        self.history["theta"].append(self.theta.detach().clone().numpy())
        return loss
