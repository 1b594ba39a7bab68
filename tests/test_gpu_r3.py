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
