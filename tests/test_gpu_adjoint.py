"""OPT-IN adjoint differentiation engine against the reference's rule -- 2P
parameter-shift evaluations (diff_method="parameter-shift", quantum_born_machine.py:58, :90, :114) -- and the oracle."""
import math

import numpy as np
import pytest
import torch

from oracle import circuit as oc
from tensornetworks_amd.bayesian_network import synthetic_network

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda", 0)


@pytest.mark.parametrize("ansatz", oc.ANSATZ_TYPES)
@pytest.mark.parametrize("n,L", [(1, 2), (2, 2), (3, 4), (5, 3), (8, 4), (11, 2)])
def test_adjoint_state_and_gradient_match_oracle_and_parameter_shift(dev, ansatz, n, L):
    from tensornetworks_amd import backend
    rng = np.random.default_rng(31 * n + L)
    P = oc.num_params(ansatz, n, L)
    th = rng.uniform(-np.pi, np.pi, P)
    w = rng.normal(size=2 ** n)
    tht, wt = torch.as_tensor(th, device=dev), torch.as_tensor(w, device=dev)
    state, probs = backend.adjoint_state(ansatz, n, L, tht)
    q_o = oc.probs(ansatz, n, L, th)
    np.testing.assert_allclose(probs.cpu().numpy(), q_o, rtol=1e-10, atol=1e-14)
    np.testing.assert_allclose((state.abs() ** 2).cpu().numpy(), q_o, rtol=1e-10, atol=1e-14)
    g_adj = backend.adjoint_vjp(ansatz, n, L, tht, state, wt).cpu().numpy()
    g_ps = backend.paramshift_grad(ansatz, n, L, tht, wt, 0, P).cpu().numpy()
    np.testing.assert_allclose(g_adj, g_ps, rtol=1e-10, atol=1e-12 * max(1.0, np.abs(g_ps).max()))
    if n <= 8:
        np.testing.assert_allclose(g_adj, oc.paramshift_vjp(ansatz, n, L, th, w), rtol=1e-9, atol=1e-12 * max(1.0, np.abs(g_ps).max()))
    # deterministic, and the state is left untouched by the backward walk
    st2 = state.clone()
    assert torch.equal(backend.adjoint_vjp(ansatz, n, L, tht, state, wt).cpu(), torch.as_tensor(g_adj))
    assert torch.equal(state, st2)


def test_c_abi_switch_routes_paramshift_grad_through_the_adjoint_walk(dev):
    """bornvi_set_option("grad_engine", 1): bornvi_paramshift_grad answers with the adjoint walk -- same numbers to
    rounding for any parameter range; 0 restores the 2P-circuit rule bit for bit."""
    from tensornetworks_amd import backend
    ansatz, n, L = "hardware_efficient", 9, 3
    P = oc.num_params(ansatz, n, L)
    rng = np.random.default_rng(9)
    tht = torch.as_tensor(rng.uniform(-1, 1, P), device=dev)
    wt = torch.as_tensor(rng.normal(size=2 ** n), device=dev)
    g0 = backend.paramshift_grad(ansatz, n, L, tht, wt, 0, P)
    import ctypes as C
    from tensornetworks_amd import _ext
    h = _ext.handle_for(dev)
    aid = backend.ansatz_id(ansatz)
    try:
        backend.set_option(dev, "grad_engine", 1)
        for lo, hi in ((0, P), (5, 17), (P - 1, P)):
            need = h.size("bornvi_paramshift_grad_workspace_bytes", aid, n, L, lo, hi)
            ws = torch.empty(need, dtype=torch.uint8, device=dev)
            g = torch.empty(hi - lo, dtype=torch.float64, device=dev)
            h.call("bornvi_paramshift_grad", aid, n, L, C.c_void_p(tht.data_ptr()), C.c_void_p(wt.data_ptr()), lo, hi,
                   C.c_void_p(g.data_ptr()), C.c_void_p(ws.data_ptr()), ws.numel(), None)
            np.testing.assert_allclose(g.cpu().numpy(), g0[lo:hi].cpu().numpy(), rtol=1e-10, atol=1e-13)
        with pytest.raises(_ext.BornviError):
            backend.set_option(dev, "grad_engine", 2)
    finally:
        backend.set_option(dev, "grad_engine", 0)
    assert torch.equal(backend.paramshift_grad(ansatz, n, L, tht, wt, 0, P), g0)


@pytest.mark.parametrize("n,L,mode", [(6, 2, "dense"), (12, 3, "dense"), (14, 2, "kron")])
def test_trainer_with_adjoint_engine_walks_the_same_trajectory(dev, n, L, mode):
    """KSDVariationalInference(grad_engine = "adjoint"): loss, gradient and three optimiser steps equal the
    parameter-shift trainer's to rounding (same q, same contraction)."""
    from tensornetworks_amd.ksd_vi_quantum import KSDVariationalInference
    bn, lat, obs, x = synthetic_network(n, seed=2)
    runs = []
    for engine in ("paramshift", "adjoint"):
        torch.manual_seed(4)
        vi = KSDVariationalInference(bn, lat, obs, qbm_num_latent_vars=n, qbm_ansatz_layers=L, pytorch_device="cuda:0",
                                     gram_mode=mode)
        vi.grad_engine = engine
        vi._prepare_stein(x)
        loss, grad, q = vi.ksd_and_grad()
        params, opt, sched = vi.make_optimizer(0.01, 3, True, "adam", (0.9, 0.999))
        losses = [vi.training_step(params, opt, sched, 10.0)[0] for _ in range(3)]
        runs.append((loss.item(), grad.cpu().numpy(), q.cpu().numpy(), losses, vi.born_machine.theta.detach().cpu().numpy()))
    (l0, g0, q0, ls0, t0), (l1, g1, q1, ls1, t1) = runs
    np.testing.assert_allclose(q1, q0, rtol=1e-10, atol=1e-15)
    assert math.isclose(l1, l0, rel_tol=1e-10)
    np.testing.assert_allclose(g1, g0, rtol=1e-8, atol=1e-10 * np.abs(g0).max())
    np.testing.assert_allclose(ls1, ls0, rtol=1e-7)
    np.testing.assert_allclose(t1, t0, rtol=0, atol=2e-6)
    with pytest.raises(ValueError):
        vi.grad_engine = "finite-difference"
        vi.ksd_and_grad()


def test_adjoint_gradient_at_full_size_n16(dev):
    """BASELINE config 3's circuit (n = 16, L = 6, 421 gates, P = 288): adjoint == 2P parameter-shift circuits."""
    from tensornetworks_amd import backend
    ansatz, n, L = "hardware_efficient", 16, 6
    P = oc.num_params(ansatz, n, L)
    g = torch.Generator().manual_seed(0)
    tht = (0.1 * torch.randn(P, generator=g, dtype=torch.float32)).double().to(dev)
    wt = torch.randn(2 ** n, generator=g, dtype=torch.float64).to(dev)
    state, probs = backend.adjoint_state(ansatz, n, L, tht)
    assert torch.allclose(probs, backend.circuit_probs(ansatz, n, L, tht.view(1, -1))[0], rtol=1e-10, atol=1e-16)
    g_adj = backend.adjoint_vjp(ansatz, n, L, tht, state, wt)
    g_ps = backend.paramshift_grad(ansatz, n, L, tht, wt, 0, P)
    assert float((g_adj - g_ps).abs().max()) <= 1e-10 * float(g_ps.abs().max())
