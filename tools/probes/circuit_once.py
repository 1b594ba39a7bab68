"""One parameter-shift batch (n = N_QUBITS, default 16; L = LAYERS, default 6) run three times: the target of PMC passes
over the circuit engine (tools/gpu_session.sh pmc_circ)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tensornetworks_amd import backend

n, L = int(os.environ.get("N_QUBITS", "16")), int(os.environ.get("LAYERS", "6"))
dev = torch.device("cuda", 0)
P = backend.num_params("hardware_efficient", n, L)
theta = (0.1 * torch.randn(P, generator=torch.Generator().manual_seed(0), dtype=torch.float32)).double().to(dev)
out = torch.empty((2 * P + 1, 1 << n), dtype=torch.float64, device=dev)
for _ in range(3):
    backend.paramshift_probs("hardware_efficient", n, L, theta, 0, P, include_base=True, out=out)
torch.cuda.synchronize()
print("done", n, L, P)
