"""GPU parity of circuit_pass_r3_kernel (8 amplitudes per thread, four waves per SIMD, compact tables, matrices by scalar
loads, pivot-normalised 12-instruction gates) against the CPU oracle and against the 16-amplitude kernel, through the C ABI.
Reference circuit: quantum_born_machine.py:58-128."""
import numpy as np
import pytest
import torch

from oracle import circuit as oc

pytestmark = pytest.mark.gpu
RTOL, ATOL = 1e-10, 1e-14


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda", 0)


@pytest.fixture()
def be(dev):
    from tensornetworks_amd import backend
    defaults = {k: backend.get_option(dev, k) for k in ("reg_wires", "read_map")}
    yield backend
    backend.set_option(dev, "tile_bits", 13)
    backend.set_option(dev, "tile_bits_multi", 0)
    for k, v in defaults.items():
        backend.set_option(dev, k, v)


def probs(be, dev, ansatz, n, L, th):
    return be.circuit_probs(ansatz, n, L, torch.as_tensor(np.atleast_2d(th), dtype=torch.float64, device=dev)).cpu().numpy()


@pytest.mark.parametrize("read_map", [0, 1])
@pytest.mark.parametrize("ansatz", oc.ANSATZ_TYPES)
@pytest.mark.parametrize("n,L,kb", [(9, 2, 0), (10, 2, 0), (12, 3, 0), (13, 2, 0), (12, 2, 9), (13, 2, 11), (14, 3, 11), (14, 3, 13),
                                    (15, 2, 12), (16, 2, 13)])
def test_r3_probs_match_oracle(be, dev, ansatz, n, L, kb, read_map):
    be.set_option(dev, "reg_wires", 3)
    be.set_option(dev, "read_map", read_map)
    be.set_option(dev, "tile_bits", kb if kb else 13)
    rng = np.random.default_rng(7 * n + L + kb)
    th = rng.uniform(-np.pi, np.pi, (3, oc.num_params(ansatz, n, L)))
    q = probs(be, dev, ansatz, n, L, th)
    for b in range(3):
        np.testing.assert_allclose(q[b], oc.probs(ansatz, n, L, th[b]), rtol=RTOL, atol=ATOL)
        assert abs(q[b].sum() - 1.0) < 1e-13


def test_r3_pivot_exchange_and_scale(be, dev):
    """Angles that put the pivot of a fused gate in the OTHER row (|u10| > |u00|: RY(pi) and neighbours): the record's
    exchange flag folds into the write address / the post sign bits, the pivots' moduli into the scale of the
    probabilities.  theta = multiples of pi / 2 and random large angles, all ansaetze, against the oracle."""
    be.set_option(dev, "reg_wires", 3)
    for ansatz in oc.ANSATZ_TYPES:
        for n, L in ((10, 3), (14, 2)):
            P = oc.num_params(ansatz, n, L)
            rng = np.random.default_rng(n + L)
            ths = np.stack([np.full(P, np.pi), np.full(P, np.pi / 2), rng.integers(0, 4, P) * np.pi / 2,
                            np.pi + 1e-9 * rng.standard_normal(P), rng.uniform(-20, 20, P)])
            q = probs(be, dev, ansatz, n, L, ths)
            for b in range(len(ths)):
                np.testing.assert_allclose(q[b], oc.probs(ansatz, n, L, ths[b]), rtol=1e-9, atol=1e-13)


def test_r3_equals_16_amplitude_kernel_at_config3(be, dev):
    """BASELINE config 3's batch (n = 16, L = 6, 577 circuits) under both persistent kernels: rows agree to rounding
    (different stage cut, normalised gates: not bitwise), every row sums to 1; three rows against the oracle's C port."""
    from oracle import cpu_port as cp
    ansatz, n, L = "hardware_efficient", 16, 6
    P = oc.num_params(ansatz, n, L)
    g = torch.Generator().manual_seed(0)
    th = (0.1 * torch.randn(P, generator=g, dtype=torch.float32)).double().to(dev)
    outs = {}
    for r, rm in ((4, 0), (3, 0), (3, 1)):
        be.set_option(dev, "reg_wires", r)
        be.set_option(dev, "read_map", rm)
        outs[(r, rm)] = be.paramshift_probs(ansatz, n, L, th, 0, P, include_base=True).clone()
        assert float((outs[(r, rm)].sum(dim=1) - 1).abs().max()) < 1e-12
    for key in ((3, 0), (3, 1)):
        assert float((outs[key] - outs[(4, 0)]).abs().max()) < 1e-15
    if cp.available():
        thn = th.cpu().numpy()
        t1, t2 = thn.copy(), thn.copy()
        t1[7] += np.pi / 2
        t2[P - 1] -= np.pi / 2
        want = cp.circuit_probs(ansatz, n, L, np.stack([thn, t1, t2]))
        got = outs[(3, 1)][[0, 15, 2 * P]].cpu().numpy()
        np.testing.assert_allclose(got, want, rtol=1e-9, atol=1e-17)
    be.release_workspaces()


def test_r3_prefix_sharing_and_kron(be, dev):
    """The opt-in prefix sharing and the matrix-free Stein mat-vec (state in, state out, one shared real matrix) run on
    the 8-amplitude kernel too: shared batch bitwise equal to the plain one; K_p q against the dense product."""
    from tensornetworks_amd.bayesian_network import synthetic_network
    from tensornetworks_amd.stein_utils import pack_network
    be.set_option(dev, "reg_wires", 3)
    be.set_option(dev, "read_map", 1)
    ansatz, n, L = "hardware_efficient", 15, 4
    P = oc.num_params(ansatz, n, L)
    th = torch.as_tensor(np.random.default_rng(3).uniform(-np.pi, np.pi, P), device=dev)
    try:
        be.set_option(dev, "prefix_share", 0)
        ref = be.paramshift_probs(ansatz, n, L, th, 0, P, include_base=True).clone()
        be.set_option(dev, "prefix_share", 1)
        got = be.paramshift_probs(ansatz, n, L, th, 0, P, include_base=True)
        assert torch.equal(ref, got)
    finally:
        be.set_option(dev, "prefix_share", 0)
    np.testing.assert_allclose(ref[0].cpu().numpy(), oc.probs(ansatz, n, L, th.cpu().numpy()), rtol=RTOL, atol=ATOL)
    n = 13
    bn, lat, obs, x = synthetic_network(n, 2)
    S, _ = be.score_from_packed(pack_network(bn, lat, x), n, dev)
    K = be.stein_gram(S, n, 1.0)
    q = torch.rand(1 << n, dtype=torch.float64, generator=torch.Generator().manual_seed(1)).to(dev)
    q /= q.sum()
    ksd2, y = be.stein_matvec_kron(S, q, n, 1.0)
    yd = K @ q
    assert float((y - yd).abs().max()) <= 1e-11 * float((K.abs() @ q).max())
    assert abs(ksd2.item() - float(q @ yd)) <= 1e-11 * float(q @ (K.abs() @ q))


@pytest.mark.parametrize("ansatz,n,L,kb", [("hardware_efficient", 14, 3, 11), ("hardware_efficient", 15, 4, 13), ("all_to_all", 14, 2, 12),
                                           ("basic", 15, 3, 13), ("hardware_efficient", 16, 2, 13)])
def test_fused_dot_equals_stored_probabilities(be, dev, ansatz, n, L, kb):
    """bornvi_paramshift_dot_begin / _finish: the parameter-shift dot product accumulated inside the shifted circuits'
    last pass against the un-fused path (probabilities written, then bornvi_ksd_grad_finish): q bitwise, loss bitwise,
    gradient to 1e-12 of its largest entry; a strided share of the parameters (one rank's deal) and the plain 1/2 scale;
    deterministic (two runs bitwise equal).  The VJP is the one implied by ksd_vi_quantum.py:150."""
    be.set_option(dev, "reg_wires", 3)
    be.set_option(dev, "read_map", 1)
    be.set_option(dev, "tile_bits", kb)
    P = oc.num_params(ansatz, n, L)
    rng = np.random.default_rng(n + L)
    th = torch.as_tensor(rng.uniform(-np.pi, np.pi, P), device=dev)
    w = torch.as_tensor(rng.standard_normal(1 << n), device=dev)
    ksd2 = torch.tensor([3.7], dtype=torch.float64, device=dev)
    assert be.paramshift_dot_supported(ansatz, n, L, dev, P)
    for lo, hi, step in ((0, P, 1), (1, P, 3)):
        cnt = len(range(lo, hi, step))
        probs = be.paramshift_probs(ansatz, n, L, th, lo, hi, include_base=True, p_stride=step).clone()
        loss_u, grad_u, _ = be.ksd_grad_finish(n, probs[1:], cnt, w, ksd2)
        q, tok = be.paramshift_dot_begin(ansatz, n, L, th, lo, hi, p_stride=step)
        loss_f, grad_f = be.paramshift_dot_finish(tok, w, ksd2)
        assert torch.equal(q, probs[0]) and torch.equal(loss_f, loss_u)
        tol = 1e-12 * float(grad_u.abs().max())
        assert float((grad_f - grad_u).abs().max()) <= tol
        q2, tok2 = be.paramshift_dot_begin(ansatz, n, L, th, lo, hi, p_stride=step)
        loss_2, grad_2 = be.paramshift_dot_finish(tok2, w, ksd2)
        assert torch.equal(grad_2, grad_f) and torch.equal(q2, q)
        none, grad_h = be.paramshift_dot_finish(tok2, w, None)             # plain 1/2 (bornvi_paramshift_grad's scale)
        ref_h = 0.5 * ((probs[1::2] - probs[2::2]) @ w)
        assert none is None and float((grad_h - ref_h).abs().max()) <= 1e-12 * float(ref_h.abs().max())
    # below the clamp of the loss the gradient is zero (torch's clamp backward, ksd_vi_quantum.py:145)
    tiny = torch.tensor([1e-13], dtype=torch.float64, device=dev)
    q, tok = be.paramshift_dot_begin(ansatz, n, L, th, 0, P)
    loss_c, grad_c = be.paramshift_dot_finish(tok, w, tiny)
    assert float(grad_c.abs().max()) == 0.0 and abs(loss_c.item() - 1e-6) < 1e-18
    be.release_workspaces()


def test_fused_dot_in_the_trainer_step(be, dev):
    """KSDVariationalInference.ksd_and_grad with the fused dot (default where available) against the same step with
    fused_dot = False, n = 14 dense: loss bitwise, gradient to 1e-12."""
    from tensornetworks_amd.bayesian_network import synthetic_network
    from tensornetworks_amd.ksd_vi_quantum import KSDVariationalInference
    be.set_option(dev, "reg_wires", 3)
    be.set_option(dev, "read_map", 1)
    n, L = 14, 3
    bn, lat, obs, x = synthetic_network(n, 0)
    torch.manual_seed(0)
    vi = KSDVariationalInference(bn, lat, obs, qbm_num_latent_vars=n, qbm_ansatz_layers=L, pytorch_device="cuda:0", gram_mode="dense")
    vi._prepare_stein(x)
    assert be.paramshift_dot_supported("hardware_efficient", n, L, dev, vi.born_machine.num_ansatz_params)
    loss_f, grad_f, q_f = vi.ksd_and_grad()
    vi.fused_dot = False
    loss_u, grad_u, q_u = vi.ksd_and_grad()
    assert torch.equal(q_f, q_u) and torch.equal(loss_f, loss_u)
    assert float((grad_f - grad_u).abs().max()) <= 1e-12 * float(grad_u.abs().max())
