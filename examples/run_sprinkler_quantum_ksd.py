#!/usr/bin/env python3
"""The reference's headline experiment on this engine: KSD variational inference of P(C, S, R | W = 1) in the Sprinkler
network with a 3-qubit Born machine (run_sprinkler_quantum_ksd.py of the reference: hardware_efficient ansatz, 4 layers,
small_random initialisation, Adam lr = 0.005 with cosine annealing, clip 10, 1000 epochs) -- the only change a user of
the reference makes is the package the two classes are imported from.  Prints the learned
distribution beside the exact posterior and the TVD statistics the reference prints; no plotting.

    python examples/run_sprinkler_quantum_ksd.py [--epochs 1000] [--device cuda:0] [--no-host-sync]

--no-host-sync: the same epochs without the per-epoch loss.item() (train(host_sync=False): one HIP-graph replay per
epoch at this size is not used because a TVD per epoch is requested; the losses are read back at the log points)."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tensornetworks_amd.bayesian_network import get_sprinkler_network          # noqa: E402
from tensornetworks_amd.ksd_vi_quantum import KSDVariationalInference         # noqa: E402
from tensornetworks_amd.utils import calculate_tvd                            # noqa: E402


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--epochs", type=int, default=1000)
    ap.add_argument("--device", default="cuda:0")
    ap.add_argument("--layers", type=int, default=4)
    ap.add_argument("--ansatz", default="hardware_efficient", choices=["hardware_efficient", "all_to_all", "basic"])
    ap.add_argument("--lr", type=float, default=0.005)
    ap.add_argument("--seed", type=int, default=0, help="torch seed of the small_random initialisation")
    ap.add_argument("--no-host-sync", action="store_true")
    ap.add_argument("--quiet", action="store_true")
    args = ap.parse_args()

    import torch
    torch.manual_seed(args.seed)
    latent, observed, evidence = ["C", "S", "R"], ["W"], {"W": 1}
    network = get_sprinkler_network(random_cpts=False)
    posterior, p_evidence = network.get_true_posterior(latent, evidence)
    print(f"Sprinkler network, evidence {evidence}: P(evidence) = {p_evidence:.4f}")
    if p_evidence < 1e-9:
        raise SystemExit("the evidence has probability zero under the network")

    vi = KSDVariationalInference(bayesian_network=network, latent_vars_names=latent, observed_vars_names=observed,
                                 qbm_num_latent_vars=len(latent), qbm_ansatz_layers=args.layers, qbm_conditioning_dim=0,
                                 qbm_pennylane_device_name="default.qubit", qbm_ansatz_type=args.ansatz,
                                 qbm_init_method="small_random", base_kernel_length_scale=1.0, pytorch_device=args.device)
    n_params = sum(p.numel() for p in vi.born_machine.parameters() if p.requires_grad)
    print(f"Born machine: {len(latent)} qubits, {args.layers} layers of {args.ansatz}, {n_params} parameters; "
          f"Adam lr {args.lr} with cosine annealing, clip 10, {args.epochs} epochs on {args.device}")

    t0 = time.perf_counter()
    history = vi.train(x_observation_dict=evidence, num_epochs=args.epochs, lr_born_machine=args.lr, verbose=not args.quiet,
                       true_posterior_for_tvd=posterior, use_lr_scheduler=True, gradient_clip_norm=10.0, optimizer_type="adam",
                       adam_betas=(0.9, 0.999), **({"host_sync": False} if args.no_host_sync else {}))
    seconds = time.perf_counter() - t0

    learned = vi.born_machine.get_prob_dict(x_condition=None)
    print(f"\n{'outcome ' + str(tuple(latent)):<22} | {'true P(z|x)':<13} | {'learned Q(z|x)':<15} | difference")
    print("-" * 70)
    worst = 0.0
    for z in sorted(posterior):
        p, q = posterior.get(z, 0.0), learned.get(z, 0.0)
        worst = max(worst, abs(p - q))
        print(f"{str(z):<22} | {p:<13.6f} | {q:<15.6f} | {abs(p - q):.6f}")
    tvd = np.asarray(history["tvd"], dtype=np.float64)
    print(f"\nFinal TVD: {calculate_tvd(posterior, learned):.6f}   max pointwise difference: {worst:.6f}")
    print(f"Best TVD during training: {tvd.min():.6f}   mean {tvd.mean():.6f}   std {tvd.std():.6f}   "
          f"mean of the last 100 epochs {tvd[-100:].mean():.6f}")
    print(f"KSD loss: first {history['loss_ksd'][0]:.6f}, last {history['loss_ksd'][-1]:.6f}; "
          f"{args.epochs} epochs in {seconds:.2f} s ({args.epochs / seconds:.0f} epochs/s)")


if __name__ == "__main__":
    main()
