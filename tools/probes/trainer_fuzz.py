"""Probe: KSDVariationalInference.train over a grid of small configurations (n, ansatz, layers, gram mode, optimiser,
gradient engine, prefix sharing, theta on cpu / cuda, hidden nodes): every run must finish with a finite history."""
import itertools
import sys
import os
import io
import contextlib
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tensornetworks_amd import backend
from tensornetworks_amd.bayesian_network import synthetic_network, get_sprinkler_network
from tensornetworks_amd.ksd_vi_quantum import KSDVariationalInference

dev = torch.device("cuda", 0)
bad = 0
grid = itertools.product((1, 2, 5, 9), ("hardware_efficient", "all_to_all", "basic"), (0, 1, 3), ("auto", "dense", "kron"),
                         ("adam", "sgd"), ("paramshift", "adjoint"), (0, 1), ("cpu", "cuda:0"))
for i, (n, ansatz, L, gm, opt, eng, share, pdev) in enumerate(grid):
    if (i * 7919) % 9:            # a fixed ninth of the 1728 combinations
        continue
    tag = f"n{n} {ansatz} L{L} {gm} {opt} {eng} share{share} theta@{pdev}"
    try:
        bn, lat, obs, x = synthetic_network(n, seed=i)
        torch.manual_seed(i)
        vi = KSDVariationalInference(bn, lat, obs, qbm_num_latent_vars=n, qbm_ansatz_layers=L, qbm_ansatz_type=ansatz,
                                     pytorch_device=pdev, gram_mode=gm)
        vi.grad_engine = eng
        backend.set_option(dev, "prefix_share", share)
        with contextlib.redirect_stdout(io.StringIO()):
            h = vi.train(x, 3, 0.05, verbose=(i % 2 == 0), optimizer_type=opt, use_lr_scheduler=bool(i % 3))
        ok = len(h["loss_ksd"]) == 3 and all(np.isfinite(h["loss_ksd"]))
        print(("ok   " if ok else "BAD  ") + tag, [round(v, 6) for v in h["loss_ksd"]], flush=True)
        bad += not ok
    except Exception as e:
        bad += 1
        print("EXC  " + tag, type(e).__name__, str(e)[:160], flush=True)
backend.set_option(dev, "prefix_share", 0)
# hidden nodes (Sprinkler: latents C, R; S marginalised) and a conditional observation
bn = get_sprinkler_network(False)
vi = KSDVariationalInference(bn, ["C", "R"], ["W"], qbm_num_latent_vars=2, qbm_ansatz_layers=2, pytorch_device="cuda:0")
with contextlib.redirect_stdout(io.StringIO()):
    h = vi.train({"W": 1}, 5, 0.05)
print("sprinkler hidden S:", [round(v, 6) for v in h["loss_ksd"]])
print("done, bad =", bad)
